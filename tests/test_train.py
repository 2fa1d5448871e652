"""The `--train` fit: the CPU restatement (oracle/mlp_fit_oracle.py) against scikit-learn's own runs (golden vectors made
by tests/golden/make_golden_train.py), GroupKFold assignments, and the host-side recipe of train_model.py:81-92."""
import json
import os

import numpy as np
import pytest

from oracle import mlp_fit_oracle as mo
from tests import helpers as H

TRAIN = os.path.join(H.GOLDEN, 'train')
MANIFEST = json.load(open(os.path.join(TRAIN, 'manifest.json')))


@pytest.mark.parametrize('tag', sorted(MANIFEST['fits']))
def test_oracle_fit_equals_sklearn(tag):
    """Same start weights, same row order (shuffle=False): loss curve, epoch count (stopping rule), weights."""
    meta = MANIFEST['fits'][tag]
    z = np.load(os.path.join(TRAIN, tag + '.npz'))
    init = (z['W1_0'], z['b1_0'], z['W2_0'], float(z['b2_0'][0]))
    got = mo.fit(z['X'], z['y'], hidden=meta['hidden'], max_iter=meta['max_iter'], shuffle=False, init=init)
    assert got['n_iter'] == meta['n_iter']
    np.testing.assert_allclose(got['loss_curve'], z['loss_curve'], rtol=1e-9, atol=0)
    for name in ('W1', 'b1', 'W2'):
        np.testing.assert_allclose(got[name], z[name], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got['b2'], z['b2'][0], rtol=1e-7, atol=1e-9)
    _, p = mo.forward(got['W1'], got['b1'], got['W2'], got['b2'], z['X'])
    np.testing.assert_allclose(p, z['proba'], rtol=1e-7, atol=1e-9)
    assert mo.accuracy(got, z['X'], z['y']) == float(z['train_accuracy'][0])


def test_init_weights_are_ours_and_glorot_bounded():
    W1, b1, W2, b2 = mo.init_weights(7, 100, seed=1007)
    z = np.load(os.path.join(TRAIN, 'n1000_h100.npz'))
    assert (W1 == z['W1_0']).all() and (W2 == z['W2_0']).all()
    assert np.abs(W1).max() <= np.sqrt(6 / 107) and np.abs(W2).max() <= np.sqrt(6 / 101)
    assert abs(W1.mean()) < 0.02 and W1.std() > 0.1


@pytest.mark.parametrize('tag', sorted(MANIFEST['group_kfold']))
def test_group_kfold_equals_sklearn(tag):
    case = MANIFEST['group_kfold'][tag]
    assert mo.group_kfold(case['groups'], 5).tolist() == case['fold']


def test_epoch_order_is_a_permutation():
    for n in (1, 2, 3, 5, 199, 200, 1000, 4097):
        for epoch in (0, 1, 17):
            o = mo.epoch_order(n, 12345, epoch)
            assert sorted(o.tolist()) == list(range(n))
    assert (mo.epoch_order(1000, 1, 0) != mo.epoch_order(1000, 1, 1)).any()


def test_product_group_kfold_equals_sklearn():
    from mcaller_amd.train_model import group_kfold
    for tag, case in MANIFEST['group_kfold'].items():
        assert group_kfold(case['groups'], 5).tolist() == case['fold'], tag
    with pytest.raises(ValueError):
        group_kfold(['a', 'b', 'a'], 5)


def test_tsv2matrix_equals_reference():
    from mcaller_amd.load_mCaller_data import tsv2matrix
    sig, ctx = tsv2matrix(os.path.join(TRAIN, 'training_rows.train'), 'A')
    gold = json.load(open(os.path.join(TRAIN, 'training_rows.dicts.json')))
    assert sig == gold['signals'] and ctx == gold['contexts']


def test_balanced_rows():
    from mcaller_amd.train_model import balanced_rows
    sig = {'m6A': [[1.0], [2.0], [3.0]], 'A': [[4.0], [5.0]]}
    grp = {'m6A': ['c1', 'c2', 'c3'], 'A': ['c4', 'c5']}
    assert balanced_rows(sig, grp) == (['m6A', 'm6A', 'A', 'A'], [[1.0], [2.0], [4.0], [5.0]], ['c1', 'c2', 'c4', 'c5'])


def test_model_file_round_trip(tmp_path):
    """Fitted arrays -> model file -> model_io: the estimators the reference would unpickle (when scikit-learn is
    installed) and the neutral .npz both load back to the same weights, with the dict (two-base) flag of the reference's
    pickle of {sub-model: estimator} (train_model.py:110-112 + extract_contexts.py:124-128)."""
    from mcaller_amd import train_model, model_io
    z = np.load(os.path.join(TRAIN, 'n333_h16.npz'))
    fit = dict(W1=z['W1'], b1=z['b1'], W2=z['W2'], b2=float(z['b2'][0]), loss_curve=z['loss_curve'], n_iter=len(z['loss_curve']))
    path = str(tmp_path / 'model.pkl')
    out = train_model.write_models({'general': fit}, {'general': ['A', 'm6A']}, {'general': 333}, path)
    ms = model_io.load_model_file(path)
    assert ms.twobase and ms.keys() == ['general']
    w = ms.models['general']
    assert (w.W1 == z['W1']).all() and (w.W2 == z['W2']).all() and w.b2[0] == z['b2'][0] and w.classes == ['A', 'm6A']
    try:
        import sklearn  # noqa: F401
    except ImportError:
        return
    est = out['general']
    np.testing.assert_allclose(est.predict_proba(z['X'])[:, 1], z['proba'], rtol=1e-12)
    assert list(est.predict(z['X'][:5])) == ['m6A' if p > 0.5 else 'A' for p in z['proba'][:5]]
    # and the neutral format
    import builtins
    real_import = builtins.__import__

    def no_sklearn(name, *a, **k):
        if name.split('.')[0] == 'sklearn':
            raise ImportError(name)
        return real_import(name, *a, **k)
    builtins.__import__ = no_sklearn
    try:
        train_model.write_models({'general': fit}, {'general': ['A', 'm6A']}, {'general': 333}, path)
    finally:
        builtins.__import__ = real_import
    ms = model_io.load_model_file(path)
    assert ms.twobase and (ms.models['general'].W1 == z['W1']).all() and ms.models['general'].classes == ['A', 'm6A']
