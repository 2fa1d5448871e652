"""Wall-clock per pass of the pipelined interface (run_async / wait), for experiments with the enqueue order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from tests import helpers as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
codes = synth.genome()
ref = synth.SynthRef(codes)
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)
for rep in range(3):
    steps = 30
    t = time.perf_counter()
    depth = int(os.environ.get('DEPTH', '3'))
    for _ in range(depth):
        dev.run_async(6, 0, 0.0)
    for _ in range(steps - depth):
        if os.environ.get('SPLIT'):          # copy-out of the oldest pass started, the next pass enqueued, then the wait
            dev.wait_begin()
            dev.run_async(6, 0, 0.0)
            r = dev.wait()
        else:
            r = dev.wait()
            dev.run_async(6, 0, 0.0)
    for _ in range(depth):
        r = dev.wait()
    dt = time.perf_counter() - t
    print('pipelined: %.3f ms per pass (%d records) env NOEVENTS=%s' % (dt / steps * 1e3, r.n, os.environ.get('MCALLER_ASYNC_NOEVENTS')))

# where the host's time goes in the split loop (bench.py's): seconds inside each call, per pass
steps, depth = 200, 3
acc = dict(wait_begin=0.0, run_async=0.0, wait=0.0, times=0.0)
t_all = time.perf_counter()
for _ in range(depth):
    dev.run_async(6, 0, 0.0)
for _ in range(steps - depth):
    t0 = time.perf_counter(); dev.wait_begin()
    t1 = time.perf_counter(); dev.run_async(6, 0, 0.0)
    t2 = time.perf_counter(); r = dev.wait()
    t3 = time.perf_counter(); dev.times_ms()
    t4 = time.perf_counter()
    acc['wait_begin'] += t1 - t0; acc['run_async'] += t2 - t1; acc['wait'] += t3 - t2; acc['times'] += t4 - t3
for _ in range(depth):
    r = dev.wait()
dt = time.perf_counter() - t_all
print('split loop: %.3f ms per pass; host inside calls, ms per pass: %s' % (
    dt / steps * 1e3, {k: round(v / (steps - depth) * 1e3, 4) for k, v in acc.items()}))
