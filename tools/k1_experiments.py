"""Timing experiments for k1_scan on the GPU box: stage cuts (MCALLER_K1_DEBUG) for the built TILE."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth
from mcaller_amd.device import Device
from mcaller_amd.extract_contexts import submodel_setup
from tests import helpers as H

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
motif = sys.argv[2] if len(sys.argv) > 2 else 'GATC'
codes = synth.genome()
ref = synth.SynthRef(codes, motif=motif)
table, qual = synth.make_table(n, seed=1000, codes=codes)
_, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
dev = Device(0)
dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual); dev.set_mlp(weights, soc)
for dbg in [int(x) for x in os.environ.get('K1_DEBUGS', '0,0').split(',')]:
    os.environ['MCALLER_K1_DEBUG'] = str(dbg)
    ts = []
    for it in range(8):
        try:
            dev.run(6, 0, 0.0)
        except Exception as e:
            pass
        ts.append(dev.times_ms())
    k1 = np.median([t['window_scan'] for t in ts[2:]])
    print('TILE=%s debug=%d k1=%.4f ms  (%.0f GB/s alg) all=%s' % (os.environ.get('MC_TILE', 'default'), dbg, k1, 17.0 * n / k1 / 1e6,
          {k: round(float(np.median([t[k] for t in ts[2:]])), 4) for k in ts[0]}))
