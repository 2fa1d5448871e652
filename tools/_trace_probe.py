import sys, time, os, ctypes as C
sys.path.insert(0, '.')
from mcaller_amd import synth, _lib
codes = synth.genome()
table, qual = synth.make_table(10000000, seed=5, codes=codes)
tsv = '/tmp/pt_syn.tsv'
synth.write_tsv_native(table, codes, tsv)
sz = os.path.getsize(tsv)
L = _lib.lib()
arr = (C.c_char_p * 1)(b'ecoli_syn')
os.environ['MCALLER_TRACE_HOST'] = '1'
for pool in (0, 1):
    L.mc_host_pool_config(pool, -1)
    for nt in (64, 128, 256):
        for rep in range(3):
            print('pool', pool, 'threads', nt, 'rep', rep, file=sys.stderr, flush=True)
            h = C.c_void_p()
            t = time.perf_counter(); L.mc_parse_eventalign_range(tsv.encode(), 0, sz, arr, 1, nt, C.byref(h)); dt = time.perf_counter() - t
            L.mc_parsed_free(h)
            print('   total %.1f ms' % (dt * 1e3), file=sys.stderr, flush=True)
