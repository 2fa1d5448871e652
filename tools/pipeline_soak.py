"""Soak of the pipelined interface: N passes over the headline table, every pass's records compared bit for bit with the first
pass's (a missed cache write-back between the streams would show up as a difference)."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from tests import helpers as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
codes = synth.genome()
ref = synth.SynthRef(codes)
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); slot = dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)


def digest(r):
    h = hashlib.blake2b(digest_size=16)
    for a in (r.feats[:r.n_calls * r.k], r.prob[:r.n_calls], r.site_pos[:r.n], r.site_seg[:r.n], r.close_row[:r.n], r.info[:r.n],
              r.call_row[:r.n]):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


depth, first, bad, done, enq = 3, None, 0, 0, [0]


def enqueue():
    # every third pass is a FULL pass (the table declared new: validating scan), the others re-scan the validated table
    dev.select_table(slot, as_new=(enq[0] % 3 == 0))
    enq[0] += 1
    dev.run_async(6, 0, 0.0)


for _ in range(depth):
    enqueue()
dev.wait_begin()
for i in range(passes):
    dev.wait_begin()
    if i + depth < passes:
        enqueue()
    d = digest(dev.wait())
    first = first or d
    bad += d != first
    done += 1
    if done + depth > passes and done >= passes:
        break
print('%d pipelined passes over %d rows (every third a full, validating pass): %d differ from the first' % (done, n, bad))
