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
