"""Classifier kernel times on the headline workload: k2_mlp vs k3_forest (50 trees, depth <= 10) on the same records."""
import os, sys
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
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual)
for tag, ms in (('mlp', H.load_modelset('r95')), ('forest', H.load_rf_modelset())):
    _, weights, _, soc = submodel_setup(ms, 'A')
    dev.set_classifier(weights, soc)
    ts = []
    for _ in range(6):
        dev.run(6, 0, 0.0)
        ts.append(dev.times_ms()['classifier'])
    rec = dev.fetch()
    print('%s: classifier stage %.3f ms (incl. the host sync before it), %d records, mean p %.4f'
          % (tag, float(np.median(ts[2:])), rec.n, float(np.nanmean(rec.prob[:rec.n]))))
