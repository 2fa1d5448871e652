#!/usr/bin/env python3
"""Golden vectors for the `--train` fit, made by RUNNING scikit-learn's MLPClassifier (the third-party code the
reference's train_model.py:47,:100 calls; 1.7.2 in the build container) -- never on the GPU box; no test imports this.

The reference fits with random_state=None, so its own runs are not reproducible; what can be pinned is the algorithm:
from GIVEN start weights and a FIXED row order (shuffle=False) scikit-learn's loss curve, epoch count and final weights
are deterministic.  Cases: the reference's hyper-parameters (train_model.py:47) on synthetic two-class feature rows
shaped like mCaller's (k slot means + read quality), sizes that exercise a short last batch (n % 200 != 0), n < 200, and
the stopping rule.  Also: GroupKFold fold assignments (train_model.py:62-65) for context-like groups.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import mlp_fit_oracle as mo      # noqa: E402  (start weights: our generator; the fit: scikit-learn)


def make_data(n, seed, sep):
    rng = np.random.default_rng(seed)
    y = (np.arange(n) % 2).astype(np.int64)
    rng.shuffle(y)
    X = np.round(rng.normal(-0.17, 2.44, size=(n, 6)), 4)
    X[y == 1, 2] += sep
    X[y == 1, 3] -= 0.6 * sep
    q = rng.uniform(6, 12, size=(n, 1))
    return np.hstack([X, q]), y


def sklearn_fit(X, y, init, max_iter):
    from sklearn.neural_network import MLPClassifier
    W1, b1, W2, b2 = init

    class Fixed(MLPClassifier):
        def _init_coef(self, fan_in, fan_out, dtype):
            if fan_in == W1.shape[0]:
                return W1.copy(), b1.copy()
            return W2.reshape(-1, 1).copy(), np.array([b2])

    m = Fixed(hidden_layer_sizes=(W1.shape[1]), alpha=0.001, learning_rate='adaptive', early_stopping=False,
              activation='tanh', shuffle=False, max_iter=max_iter)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.fit(X, np.where(y > 0, 'm6A', 'A'))
    assert list(m.classes_) == ['A', 'm6A']
    return m


def main():
    # --out DIR: write under DIR/tests/golden/train instead of into the repository (tests/test_pin_recipe.py)
    out_root = os.path.join(os.path.abspath(sys.argv[sys.argv.index('--out') + 1]), 'tests', 'golden') if '--out' in sys.argv else HERE
    out = os.path.join(out_root, 'train')
    os.makedirs(out, exist_ok=True)
    cases = [('n1000_h100', 1000, 100, 2.0, 200), ('n450_h100', 450, 100, 1.2, 60), ('n150_h100', 150, 100, 3.0, 200),
             ('n333_h16', 333, 16, 2.5, 40), ('n3000_h16_noise_stops', 3000, 16, 0.0, 200), ('n1000_h4_noise_stops', 1000, 4, 0.0, 200)]
    manifest = {}
    for tag, n, h, sep, max_iter in cases:
        X, y = make_data(n, 100 + n, sep)
        init = mo.init_weights(X.shape[1], h, seed=7 + n)
        m = sklearn_fit(X, y, init, max_iter)
        np.savez_compressed(os.path.join(out, tag + '.npz'), X=X, y=y.astype(np.uint8), W1_0=init[0], b1_0=init[1],
                            W2_0=init[2], b2_0=np.array([init[3]]), loss_curve=np.array(m.loss_curve_),
                            W1=m.coefs_[0], b1=m.intercepts_[0], W2=m.coefs_[1].reshape(-1), b2=m.intercepts_[1],
                            proba=m.predict_proba(X)[:, 1], train_accuracy=np.array([m.score(X, np.where(y > 0, 'm6A', 'A'))]))
        manifest[tag] = dict(n=n, hidden=h, max_iter=max_iter, n_iter=int(m.n_iter_), final_loss=float(m.loss_),
                             init_seed=7 + n)
        print(tag, 'epochs', m.n_iter_, 'loss', m.loss_)
    # GroupKFold assignments
    from sklearn.model_selection import GroupKFold
    rng = np.random.default_rng(5)
    folds = {}
    for tag, n_groups in [('g40', 40), ('g7', 7), ('g5', 5)]:
        sizes = rng.integers(1, 30, size=n_groups)
        groups = np.repeat(['CTX%03d' % i for i in range(n_groups)], sizes)
        rng.shuffle(groups)
        fold = np.full(len(groups), -1)
        for f, (_, test) in enumerate(GroupKFold(n_splits=5).split(np.zeros(len(groups)), groups=groups)):
            fold[test] = f
        folds[tag] = dict(groups=[str(g) for g in groups], fold=[int(f) for f in fold])
    # the reference's tsv2matrix (load_mCaller_data.py) on the committed --train capture plus rows it must leave out
    import tempfile
    import make_golden as mg
    mg.install_shims(tempfile.mkdtemp(prefix='mcaller_golden_train_'))
    from load_mCaller_data import tsv2matrix
    src = open(os.path.join(HERE, 'ref_outputs', 'train_positions_all.diffs.6.train')).read().strip().split('\n')
    extra = src[0].split('\t')
    extra[4] = ','.join(['0'] + extra[4].split(',')[1:])                      # a skipped slot: row is dropped (:15)
    tsv = os.path.join(out, 'training_rows.train')
    open(tsv, 'w').write('\n'.join(src + ['\t'.join(extra)]) + '\n')
    sig, ctx = tsv2matrix(tsv, 'A')
    json.dump(dict(signals=sig, contexts=ctx), open(os.path.join(out, 'training_rows.dicts.json'), 'w'))
    json.dump(dict(fits=manifest, group_kfold=folds, sklearn=__import__('sklearn').__version__),
              open(os.path.join(out, 'manifest.json'), 'w'), indent=0)


if __name__ == '__main__':
    main()
