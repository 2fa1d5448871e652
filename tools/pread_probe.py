"""How fast can the host get the text of an eventalign file out of the page cache at all?  pread of 256 KB blocks into private
buffers from N threads (what the parser's threads do before they parse), and the whole-file parse beside it."""
import sys, time, os, threading, ctypes as C
sys.path.insert(0, '.')
from mcaller_amd import synth, _lib
codes = synth.genome()
table, qual = synth.make_table(10000000, seed=5, codes=codes)
tsv = '/tmp/pt_syn.tsv'
synth.write_tsv_native(table, codes, tsv) if hasattr(synth, 'write_tsv_native') else synth.write_tsv(table, codes, tsv)
sz = os.path.getsize(tsv)
fd = os.open(tsv, os.O_RDONLY)


def reader(lo, hi, block):
    off = lo
    while off < hi:
        b = os.pread(fd, min(block, hi - off), off)
        off += len(b)


for block in (256 << 10, 4 << 20):
    for nt in (8, 32, 64, 128, 256):
        best = 1e9
        for rep in range(3):
            th = [threading.Thread(target=reader, args=(sz * i // nt, sz * (i + 1) // nt, block)) for i in range(nt)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            best = min(best, time.perf_counter() - t)
        print('pread %4d KB blocks, %3d threads: %.4f s = %.1f GB/s' % (block >> 10, nt, best, sz / best / 1e9), flush=True)
L = _lib.lib()
arr = (C.c_char_p * 1)(b'ecoli_syn')
for nt in (32, 64, 128, 256):
    best = 1e9
    for rep in range(3):
        h = C.c_void_p()
        t = time.perf_counter(); L.mc_parse_eventalign_range(tsv.encode(), 0, sz, arr, 1, nt, C.byref(h)); dt = time.perf_counter() - t
        L.mc_parsed_free(h)
        best = min(best, dt)
    print('parse whole file, %3d threads: %.4f s = %.1f GB/s' % (nt, best, sz / best / 1e9), flush=True)
print('cores', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
