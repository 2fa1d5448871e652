import os, sys, tempfile
sys.path.insert(0, '.')
import numpy as np
from mcaller_amd import _lib
from mcaller_amd.device import Device
dev = Device(0)
d = tempfile.mkdtemp()
p = os.path.join(d, 'a.tsv')
row = 'c1\t{pos}\tAAAAAA\tread{r}\tt\t{idx}\t80.5\t1.0\t0.001\tAAAAAA\t81.25\t1.0\t0.1\n'
open(p, 'w').write(''.join(row.format(pos=i, r=i // 50, idx=i) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)))
text = _lib.TextBlock(p, 0, os.path.getsize(p))
print('begin', flush=True)
slot = dev.parse_begin(text, ['c1'], 100000)
print('slot', slot, flush=True)
dev.sync()
print('synced', flush=True)
t = dev.parse_end(slot, text)
print('end', None if t is None else (t.n_rows, t.n_seg), getattr(dev, 'parse_fallback_reason', None), flush=True)
