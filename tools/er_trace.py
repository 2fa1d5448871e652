"""Phase times of k1_emit_runs from the MC_ER_TRACE variant build (tools/variants.sh er_trace "MC_ER_TRACE=1"):
MCALLER_LIB=mcaller_amd/variants/er_trace.so python tools/er_trace.py [rows]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from mcaller_amd.model_io import load_model_file, shipped_model
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
codes = synth.genome()
ref = synth.SynthRef(codes, motif='A')
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(load_model_file(shipped_model()), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)
for _ in range(3):
    dev.run(6, 0, 0.0)
print(dev.times_ms())
buf = np.zeros(1024 * 8, dtype=np.uint64)
L = _lib.lib()
L.mc_debug_er_trace.argtypes = [C.c_void_p, C.c_int64]
assert L.mc_debug_er_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 8).astype(np.int64)
ok = t[:, 5] > 0
names = ['rows + table (barrier)', 'heads found', 'runs numbered', 'means', 'windows']
for i, nm in enumerate(names):
    d = (t[ok, i + 1] - t[ok, i]) * 10
    print('%-24s mean %7.0f ns  p90 %7.0f  max %7.0f' % (nm, d.mean(), np.percentile(d, 90), d.max()))
d = (t[ok, 5] - t[ok, 0]) * 10
print('%-24s mean %7.0f ns  p90 %7.0f  max %7.0f   (%d workgroups)' % ('whole workgroup', d.mean(), np.percentile(d, 90), d.max(), ok.sum()))
