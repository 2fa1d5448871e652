"""What the side stream's kernels cost the pipelined sparse step: bench.py's loop (two tables in turn, each declared new, three passes
in flight) with and without the classifier on the side stream.   python tools/side_probe.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from mcaller_amd.model_io import load_model_file, shipped_model

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
codes = synth.genome()
ref = synth.SynthRef(codes)
dev = Device(0)
dev.set_reference(ref.device_arrays())
_, weights, _, soc = submodel_setup(load_model_file(shipped_model()), 'A')
dev.set_mlp(weights, soc)
slots = []
for seed in (1000, 1001):
    table, qual = synth.make_table(n, seed=seed, codes=codes)
    s = dev.upload_table_async(table, qual)
    dev.wait_upload(s)
    slots.append(s)
dev.set_pass_timing(0)
for score in (True, False, True, False):
    steps, depth = 200, 3
    def enqueue(i):
        dev.select_table(slots[i & 1], as_new=True)
        dev.run_async(6, 0, 0.0, score=score)
    for i in range(depth):
        enqueue(i)
    for i in range(10):
        dev.wait(); enqueue(depth + i)
    t = time.perf_counter()
    for i in range(steps):
        dev.wait_begin()
        enqueue(depth + 10 + i)
        dev.wait()
    dt = time.perf_counter() - t
    for i in range(depth):
        dev.wait()
    print('score=%s: %.4f ms per step' % (score, dt / steps * 1e3))
