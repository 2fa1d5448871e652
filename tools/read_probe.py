"""mc_read_file_range: a 146 MB piece of a file out of the page cache into pinned / plain memory, by thread count."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd.device import Device
dev = Device(0)                                   # (pinned allocations need the runtime)
codes = synth.genome()
table, qual = synth.make_table(2500000, seed=5, codes=codes)
tsv = '/tmp/rp_syn.tsv'
synth.write_tsv_native(table, codes, tsv)
sz = os.path.getsize(tsv)
n = min(sz, 146 << 20)
L = _lib.lib()
pinned = _lib.PinnedArray((n,), np.uint8)
plain = np.empty(n, dtype=np.uint8)
print('pinned' if L.mc_host_is_pinned(pinned.ptr) else 'NOT pinned', n, 'bytes')
for name, ptr in (('pinned', pinned.ptr), ('plain', plain.ctypes.data)):
    for nt in (4, 8, 16, 32, 64, 0):
        best = 1e9
        for rep in range(4):
            t = time.perf_counter()
            _lib.check(L.mc_read_file_range(tsv.encode(), 0, n, ptr, nt))
            best = min(best, time.perf_counter() - t)
        print('%-6s %3d threads: %.2f ms = %.1f GB/s' % (name, nt, best * 1e3, n / best / 1e9), flush=True)
