"""Phase timeline of k2_mlp from the MC_K2_TRACE variant build (tools/variants.sh k2_trace "MC_K2_TRACE=1"):
MCALLER_LIB=mcaller_amd/variants/k2_trace.so python tools/k2_trace.py [rows] [motif]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from mcaller_amd.model_io import load_model_file, shipped_model
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
motif = sys.argv[2] if len(sys.argv) > 2 else 'GATC'
codes = synth.genome()
ref = synth.SynthRef(codes, motif=motif)
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(load_model_file(shipped_model()), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)
for _ in range(4):
    dev.run(6, 0, 0.0)
W = 16
buf = np.zeros(1024 * W * 16, dtype=np.uint64)
L = _lib.lib()
L.mc_debug_k2_trace.argtypes = [C.c_void_p, C.c_int64]
assert L.mc_debug_k2_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, W, 16)
used = (t[:, :, 0] > 0) & (t[:, :, 6] > 0)
t0 = t[:, :, 0][used].min()
names = ['entry', 'simd setup done', 'A: classified', 'lists built', 'B done', 'after barrier', 'C done']
print('blocks with stamps:', int(used.any(axis=1).sum()), ' waves:', int(used.sum()))
for i, nm in enumerate(names):
    v = (t[:, :, i][used].astype(np.int64) - int(t0)) * 10      # ns
    print('%-18s min %7d  mean %7d  p90 %7d  max %7d ns' % (nm, v.min(), v.mean(), np.percentile(v, 90), v.max()))
info = t[:, :, 7][used]
simd = info & 15; quarter = (info >> 4) & 15; gfirst = (info >> 8) & 255; gstep = (info >> 16) & 255; ng = info >> 24
tick = dcy.sum() / dns.sum() if False else None
a_cy = t[:, :, 10][used].astype(np.int64) - t[:, :, 15][used].astype(np.int64)        # stretch begins -> A: classified (clock64 ticks)
b_cy = t[:, :, 12][used].astype(np.int64) - t[:, :, 11][used].astype(np.int64)
l_cy = t[:, :, 11][used].astype(np.int64) - t[:, :, 10][used].astype(np.int64)
c_cy = t[:, :, 14][used].astype(np.int64) - t[:, :, 12][used].astype(np.int64)
print('last stretch, clock64 ticks (~2.07 per ns): loads + classify mean %.0f  lists %.0f  B %.0f  barrier + C %.0f' % (a_cy.mean(), l_cy.mean(), b_cy.mean(), c_cy.mean()))
print('groups per block: min %d mean %.2f max %d' % (ng.min(), ng.mean(), ng.max()))
print('g_step histogram:', np.bincount(gstep.astype(int)))
print('quarter == simd for %.3f of the waves' % float((simd == quarter).mean()))
b = t[0]
dns = (t[:, :, 4][used].astype(np.int64) - t[:, :, 3][used].astype(np.int64)) * 10
dcy = t[:, :, 12][used].astype(np.int64) - t[:, :, 11][used].astype(np.int64)
print('clock64 ticks per ns during B: %.3f' % (dcy.sum() / dns.sum()))
print('block 0: wave simd quarter g_first  B-time(ns)')
for w in range(W):
    i = int(b[w, 7])
    print('   %2d  %d  %d  %d   %6d' % (w, i & 15, (i >> 4) & 15, (i >> 8) & 255, (int(b[w, 4]) - int(b[w, 3])) * 10))

# every stretch of wave 0 of the first 256 workgroups: begin, A done, behind the first barrier, B done (wave 0's groups), behind the
# second barrier, C done
tl = np.zeros(256 * 24 * 8, dtype=np.uint64)
L.mc_debug_k2_timeline.argtypes = [C.c_void_p, C.c_int64]
if L.mc_debug_k2_timeline(tl.ctypes.data, tl.size) == 0:
    tl = tl.reshape(256, 24, 8).astype(np.int64) * 10
    ok = (tl[:, :, 0] > 0) & (tl[:, :, 5] > 0)
    names = ['A (loads, lists)', 'wait at barrier 1', 'B (own groups)', 'wait at barrier 2', 'C']
    full = ok.copy(); full[:, 1:] &= ok[:, 1:]          # (every stretch that has stamps)
    print('stretches with stamps: %d; mean ns per phase of a stretch:' % int(ok.sum()))
    for i, nm in enumerate(names):
        d = (tl[:, :, i + 1] - tl[:, :, i])[ok]
        print('   %-20s mean %7.0f  p50 %7.0f  p90 %7.0f' % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
    d = (tl[:, :, 5] - tl[:, :, 0])[ok]
    print('   %-20s mean %7.0f' % ('stretch', d.mean()))
    gap = (tl[:, 1:, 0] - tl[:, :-1, 5])[ok[:, 1:] & ok[:, :-1]]
    print('   %-20s mean %7.0f' % ('between stretches', gap.mean()))
    print('workgroup 0, stretch by stretch (ns since its first): begin, A, b1, B, b2, C')
    for j in range(24):
        if ok[0, j]:
            print('   %2d  ' % j + ' '.join('%7d' % (tl[0, j, i] - tl[0, 0, 0]) for i in range(6)))
