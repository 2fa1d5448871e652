import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from mcaller_amd import synth, _lib
codes = synth.genome()
t, q = synth.make_table(10000000, seed=5000, codes=codes)
d = tempfile.mkdtemp(prefix='mc_pp_')
paths = synth.write_inputs(t, q, codes, d)
size = os.path.getsize(paths['tsv'])
for n in (1, 8):
    cuts = _lib.eventalign_read_cuts(paths['tsv'], n, 0, size)
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            tb = _lib.parse_eventalign(paths['tsv'], cuts[i], cuts[i+1], ['ecoli_syn'], 0, exact_range=True)
        print(n, 'shards: %.4f s' % (time.perf_counter() - t0), flush=True)
os.environ['MCALLER_TRACE_HOST'] = '1'
cuts = _lib.eventalign_read_cuts(paths['tsv'], 8, 0, size)
tb = _lib.parse_eventalign(paths['tsv'], cuts[2], cuts[3], ['ecoli_syn'], 0, exact_range=True)
for nt in (16, 32, 64, 128):
    t0 = time.perf_counter(); tb = _lib.parse_eventalign(paths['tsv'], cuts[2], cuts[3], ['ecoli_syn'], nt, exact_range=True); print('threads', nt, '%.4f' % (time.perf_counter()-t0))
import shutil; shutil.rmtree(d)
