"""Phase times of k1_fused from the MC_FD_TRACE variant build (tools/variants.sh fd_trace "MC_FD_TRACE=1"):
MCALLER_LIB=mcaller_amd/variants/fd_trace.so python tools/fd_trace.py [rows]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd.device import Device
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
codes = synth.genome()
ref = synth.SynthRef(codes, motif='A')
table, qual = synth.make_table(n, seed=1000, codes=codes)
dev = Device(0)
dev.set_reference(ref.device_arrays())
slot = dev.upload_table_async(table, qual)
for i in range(4):
    if i == 3:
        dev.select_table(slot, as_new=(len(sys.argv) > 2 and sys.argv[2] == 'validate'))
    dev.run_async(6, 0, 0.0, score=False)
    dev.wait()
print(dev.last_pass_info())
buf = np.zeros(1024 * 10, dtype=np.uint64)
L = _lib.lib()
L.mc_debug_fd_trace.argtypes = [C.c_void_p, C.c_int64]
assert L.mc_debug_fd_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 10).astype(np.int64)
ok = t[:, 8] > 0
names = ['rows + blocks (barrier)', 'heads found', 'runs numbered', 'means + sites', 'barrier', 'closers listed', 'windows', 'holes']
for i, nm in enumerate(names):
    d = (t[ok, i + 1] - t[ok, i]) * 10
    print('%-24s mean %7.0f ns  p90 %7.0f  max %7.0f' % (nm, d.mean(), np.percentile(d, 90), d.max()))
d = (t[ok, 8] - t[ok, 0]) * 10
print('%-24s mean %7.0f ns  p90 %7.0f  max %7.0f   (%d workgroups)' % ('whole workgroup', d.mean(), np.percentile(d, 90), d.max(), ok.sum()))
