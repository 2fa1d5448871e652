"""Time of the --train fit on the GPU (mc_mlp_fit: 5 GroupKFold fits + the final fit in one launch) vs the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcaller_amd.device import Device
from oracle import mlp_fit_oracle as mo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
rng = np.random.default_rng(1)
y = (np.arange(n) % 2).astype(np.uint8)
X = np.round(rng.normal(-0.17, 2.44, size=(n, 6)), 4)
X[y == 1, 2] += 1.5
X = np.hstack([X, rng.uniform(6, 12, size=(n, 1))])
fold = np.arange(n) % 5
rows = np.arange(n)
jobs = [(rows[fold != f], rows[fold == f]) for f in range(5)] + [(rows, np.zeros(0, np.int64))]
dev = Device(0)
for rep in range(2):
    t = time.perf_counter()
    fits = dev.mlp_fit(X, y, jobs, hidden=100, seed=3)
    dt = time.perf_counter() - t
print('GPU: %d rows, 6 fits, epochs %s: %.3f s; CV accuracy %s' % (n, [f['n_iter'] for f in fits], dt,
      [round(f['val_correct'] / max(f['n_val'], 1), 3) for f in fits[:5]]))
t = time.perf_counter()
w = mo.fit(X, y, hidden=100, max_iter=fits[5]['n_iter'], seed=3 + 5, tol=-1.0)
dt = time.perf_counter() - t
print('CPU oracle (numpy, one fit of %d epochs): %.3f s' % (w['n_iter'], dt))
