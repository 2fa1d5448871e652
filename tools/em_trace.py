"""Phase times of k1_emit (the sparse emit) from the MC_EM_TRACE variant build (tools/variants.sh em_trace "MC_EM_TRACE=1"):
MCALLER_LIB=mcaller_amd/variants/em_trace.so python tools/em_trace.py [rows]
The first wave of the first 1024 workgroups stamps its first two rounds (100 MHz clock)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from mcaller_amd.model_io import load_model_file, shipped_model
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
codes = synth.genome()
ref = synth.SynthRef(codes, motif='GATC')
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(load_model_file(shipped_model()), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)
for _ in range(3):
    dev.run(6, 0, 0.0)
print(dev.times_ms())
buf = np.zeros(1024 * 2 * 8, dtype=np.uint64)
L = _lib.lib()
L.mc_debug_em_trace.argtypes = [C.c_void_p, C.c_int64]
assert L.mc_debug_em_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 2, 8).astype(np.int64)
names = ['descriptor + rows (one trip)', 'rows sorted into slots', 'slot means (event loads)', 'mask word + base waited for', 'info, packing counts']
for rnd in (0, 1):
    ok = t[:, rnd, 5] > 0
    print('round %d (%d waves)' % (rnd, ok.sum()))
    if not ok.any():
        continue
    for i, nm in enumerate(names):
        d = (t[ok, rnd, i + 1] - t[ok, rnd, i]) * 10
        print('  %-30s mean %7.0f ns  p90 %7.0f  max %7.0f' % (nm, d.mean(), np.percentile(d, 90), d.max()))
    d = (t[ok, rnd, 5] - t[ok, rnd, 0]) * 10
    print('  %-30s mean %7.0f ns  p90 %7.0f  max %7.0f' % ('whole round', d.mean(), np.percentile(d, 90), d.max()))
ok = (t[:, 0, 5] > 0) & (t[:, 0, 7] > 0)
d = (t[ok, 0, 0] - t[ok, 0, 7]) * 10
print('kernel start -> first round     mean %7.0f ns  p90 %7.0f' % (d.mean(), np.percentile(d, 90)))
st = (t[ok, 0, 7] - t[ok, 0, 7].min()) / 100.0
print('workgroup start after the first one (us): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f' % tuple(np.percentile(st, [10, 50, 90, 100])))
en = (np.maximum(t[ok, 0, 5], t[ok, 1, 5]) - t[ok, 0, 7].min()) / 100.0
print('end of the second round after the first start (us): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f' % tuple(np.percentile(en, [10, 50, 90, 100])))
start = t[ok, 0, 7].min()
last = np.maximum(t[:, 0, 5], t[:, 1, 5]).max()
print('first stamp -> last stamp of the traced waves: %.1f us' % ((last - start) / 100.0))
