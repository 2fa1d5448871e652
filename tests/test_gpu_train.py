"""The HIP fit kernel (mc_mlp_fit) against the CPU restatement of scikit-learn's optimiser (oracle/mlp_fit_oracle.py,
itself pinned against scikit-learn runs) and against the committed scikit-learn golden vectors.  fp64 on both sides, but
tanh/exp/log and the summation order differ in the last bits and Adam amplifies that slowly: tolerances are stated."""
import json
import os

import numpy as np
import pytest

from oracle import mlp_fit_oracle as mo
from tests import helpers as H

pytestmark = pytest.mark.gpu
TRAIN = os.path.join(H.GOLDEN, 'train')
MANIFEST = json.load(open(os.path.join(TRAIN, 'manifest.json')))


@pytest.fixture(scope='module')
def dev():
    from mcaller_amd.device import Device
    d = Device(0)
    yield d
    d.close()


@pytest.mark.parametrize('tag', sorted(MANIFEST['fits']))
def test_fit_from_given_start_equals_sklearn(dev, tag):
    """scikit-learn's own run (same start weights, shuffle=False): loss curve to 1e-6 relative, same epoch count
    (stopping rule), weights to 1e-4, probabilities to 1e-5 (the north star's tolerance on p)."""
    meta = MANIFEST['fits'][tag]
    z = np.load(os.path.join(TRAIN, tag + '.npz'))
    n = len(z['y'])
    init = [(z['W1_0'], z['b1_0'], z['W2_0'], float(z['b2_0'][0]))]
    got = dev.mlp_fit(z['X'], z['y'], [(np.arange(n), np.arange(n))], hidden=meta['hidden'], max_iter=meta['max_iter'],
                      shuffle=False, init=init)[0]
    assert got['n_iter'] == meta['n_iter']
    np.testing.assert_allclose(got['loss_curve'], z['loss_curve'], rtol=1e-6)
    np.testing.assert_allclose(got['W1'], z['W1'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got['W2'], z['W2'], rtol=1e-4, atol=1e-6)
    _, p = mo.forward(got['W1'], got['b1'], got['W2'], got['b2'], z['X'])
    np.testing.assert_allclose(p, z['proba'], atol=1e-5)
    assert got['val_correct'] == int(round(float(z['train_accuracy'][0]) * n))


def test_seeded_fits_equal_the_oracle(dev):
    """Our start weights and epoch shuffles (the parts scikit-learn leaves to random_state=None): several jobs at once,
    train/validation splits, a short last batch."""
    z = np.load(os.path.join(TRAIN, 'n1000_h100.npz'))
    X, y = z['X'], z['y']
    fold = np.arange(len(y)) % 5
    jobs = [(np.nonzero(fold != f)[0], np.nonzero(fold == f)[0]) for f in range(5)] + [(np.arange(len(y)), np.zeros(0, np.int64))]
    seeds = [11, 12, 13, 14, 15, 99]
    got = dev.mlp_fit(X, y, jobs, hidden=100, max_iter=25, seeds=seeds)
    for j, (tr, va) in enumerate(jobs):
        want = mo.fit(X[tr], y[tr], hidden=100, max_iter=25, seed=seeds[j])
        assert got[j]['n_iter'] == want['n_iter'] == 25
        np.testing.assert_allclose(got[j]['loss_curve'], want['loss_curve'], rtol=1e-7)
        np.testing.assert_allclose(got[j]['W1'], want['W1'], rtol=1e-5, atol=1e-7)
        if len(va):
            assert abs(got[j]['val_correct'] - round(mo.accuracy(want, X[va], y[va]) * len(va))) <= 1
            assert got[j]["val_correct"] > 0.6 * len(va)          # better than chance after 25 epochs


def test_small_shapes(dev):
    rng = np.random.default_rng(3)
    for n, d, h in [(1, 7, 100), (5, 3, 4), (201, 9, 128), (64, 2, 1)]:
        X = rng.normal(size=(n, d))
        y = (rng.random(n) < 0.5).astype(np.uint8)
        got = dev.mlp_fit(X, y, [(np.arange(n), np.arange(n))], hidden=h, max_iter=8, seed=5)[0]
        want = mo.fit(X, y, hidden=h, max_iter=8, seed=5)
        np.testing.assert_allclose(got['loss_curve'], want['loss_curve'], rtol=1e-8)
        np.testing.assert_allclose(got['W2'], want['W2'], rtol=1e-6, atol=1e-9)


def test_more_fits_than_xcds_and_repeatability(dev):
    """Eleven fits in one call (more than the eight a launch spreads over the XCDs one per XCD: the plain numbering of a fit's
    workgroups) against the oracle, and the same call again: the workgroups of a fit add their sums in a fixed order, so a repeated
    call gives the same bytes whichever workgroup arrived first."""
    z = np.load(os.path.join(TRAIN, 'n1000_h100.npz'))
    X, y = z['X'], z['y']
    n = len(y)
    jobs = [(np.nonzero(np.arange(n) % 11 != f)[0], np.nonzero(np.arange(n) % 11 == f)[0]) for f in range(11)]
    seeds = list(range(21, 32))
    got = dev.mlp_fit(X, y, jobs, hidden=100, max_iter=12, seeds=seeds)
    again = dev.mlp_fit(X, y, jobs, hidden=100, max_iter=12, seeds=seeds)
    for j in (0, 5, 10):
        want = mo.fit(X[jobs[j][0]], y[jobs[j][0]], hidden=100, max_iter=12, seed=seeds[j])
        np.testing.assert_allclose(got[j]['loss_curve'], want['loss_curve'], rtol=1e-7)
        np.testing.assert_allclose(got[j]['W1'], want['W1'], rtol=1e-5, atol=1e-7)
    for j in range(11):
        assert np.array_equal(got[j]['W1'], again[j]['W1']) and np.array_equal(got[j]['loss_curve'], again[j]['loss_curve'])
    # (the library reads MCALLER_FIT_WGS once per process: other group sizes are exercised by tools/config5.py runs under that
    # variable -- profiles/README.md -- and by the build macro MC_FIT_GROUPS)
