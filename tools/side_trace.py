"""Phase timeline of the side stream's kernel (k2_mlp<.., PACK>) from the MC_K2_TRACE variant build (tools/variants.sh k2_trace "MC_K2_TRACE=1"):
MCALLER_LIB=mcaller_amd/variants/k2_trace.so python tools/side_trace.py [rows] [motif]   -- pipelined passes, one in flight."""
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
dev.set_reference(ref.device_arrays()); dev.set_mlp(weights, soc)
slot = dev.upload_table_async(table, qual)
for _ in range(4):
    dev.run_async(6, 0, 0.0, score=True)
    rec = dev.wait()
print('records', rec.n, 'pass info', dev.last_pass_info())
W = 16
buf = np.zeros(1024 * W * 16, dtype=np.uint64)
L = _lib.lib()
L.mc_debug_side_trace.argtypes = [C.c_void_p, C.c_int64]
assert L.mc_debug_side_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, W, 16).astype(np.int64)
used = (t[:, :, 0] > 0) & (t[:, :, 14] > 0)
t0 = t[:, :, 0][used].min()
names = ['entry', 'simd setup done', 'counts read', 'counts reduced + layout', 'walks done (prologue)', 'A: loaded, classified, listed', 'behind barrier 1',
         'packed', 'B done', 'behind barrier 2', 'C done', '', '', 'fp64 evaluations done', 'through']
print('blocks with stamps:', int(used.any(axis=1).sum()), ' waves:', int(used.sum()))
for i, nm in enumerate(names):
    col = t[:, :, i][used]
    ok = col > 0
    if not nm or not ok.any():
        continue
    v = (col[ok] - t0) * 10
    print('%-32s min %7d  mean %7d  p90 %7d  max %7d ns' % (nm, v.min(), v.mean(), np.percentile(v, 90), v.max()))

# every stretch of wave 0 of the first 256 workgroups (K2_TL): begin, A done, behind the first barrier, B done (wave 0's groups), behind
# the second barrier, C + packing done
tl = np.zeros(256 * 24 * 8, dtype=np.uint64)
L.mc_debug_k2_timeline.argtypes = [C.c_void_p, C.c_int64]
if L.mc_debug_k2_timeline(tl.ctypes.data, tl.size) == 0:
    tl = tl.reshape(256, 24, 8).astype(np.int64) * 10
    ok = (tl[:, :, 0] > 0) & (tl[:, :, 5] > 0)
    names = ['A (loads, lists)', 'wait at barrier 1', 'B (own groups)', 'wait at barrier 2', 'C + packing']
    print('stretches with stamps: %d; mean ns per phase of a stretch:' % int(ok.sum()))
    for i, nm in enumerate(names):
        d = (tl[:, :, i + 1] - tl[:, :, i])[ok]
        print('   %-20s mean %7.0f  p50 %7.0f  p90 %7.0f' % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
    d = (tl[:, :, 5] - tl[:, :, 0])[ok]
    print('   %-20s mean %7.0f' % ('stretch', d.mean()))
    gap = (tl[:, 1:, 0] - tl[:, :-1, 5])[ok[:, 1:] & ok[:, :-1]]
    if gap.size:
        print('   %-20s mean %7.0f' % ('between stretches', gap.mean()))
