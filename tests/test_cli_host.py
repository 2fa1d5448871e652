"""Host-side pieces of the command line that need no GPU: the `sort -n -k2 | uniq` merge behind `-t N` (mCaller.py:106),
the model files the reference ships."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers as H

REF = '/root/reference'


def test_merge_orders_like_sort_n_k2(tmp_path):
    """Read names are UUIDs: many start with digits, and `sort -n -k2` orders those by their numeric prefix before it falls
    back to the whole line.  Checked against sort(1) itself in the C locale."""
    from mcaller_amd.mCaller import merge_like_sort_uniq, numeric_key_k2
    names = ['2289b392-aaaa', 'cc1d-ffff', '0041', '41zz', '-7-neg', '3.5e', '3.25', '10', '9', '  spaced', '2289b392-aaaa',
             '007', '7', '1e3', '.5', '-.5x', 'abc']
    rows = []
    rng = np.random.default_rng(3)
    for i, nm in enumerate(names * 3):
        rows.append('chr%d\t%s\t%d\tGATCM\t0.1,0.2\t+\tA\t0.%d\n' % (rng.integers(0, 3), nm, rng.integers(0, 50), i % 10))
    rows += rows[:7]                                          # duplicates: uniq drops them
    parts = [str(tmp_path / ('x.tmp%d' % i)) for i in range(3)]
    for i, p in enumerate(parts):
        open(p, 'w').write(''.join(rows[i::3]))
    whole = str(tmp_path / 'whole.txt')
    open(whole, 'w').write(''.join(rows))
    want = subprocess.run('sort -n -k2 %s | uniq' % whole, shell=True, env=dict(os.environ, LC_ALL='C'), capture_output=True,
                          check=True).stdout
    out = str(tmp_path / 'merged')
    merge_like_sort_uniq(parts, out)
    assert open(out, 'rb').read() == want
    assert not any(os.path.exists(p) for p in parts)
    assert numeric_key_k2(b'c\t2289b392-x\t1\n') == 2289 and numeric_key_k2(b'c\tcc1d\t1\n') == 0


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is only present in the build container')
@pytest.mark.parametrize('tag', sorted(H.MODEL_STEMS))
def test_the_shipped_pickles_load_without_scikit_learn(tag):
    """extract_contexts.py:123-130: the model file is a pickle of an estimator or of a dict of estimators.  The four files
    the reference ships go through model_io's restricted unpickler (no scikit-learn import) and give the arrays of the
    committed .npz exports."""
    from mcaller_amd.model_io import load_model_file
    stem = H.MODEL_STEMS[tag]
    got = load_model_file(os.path.join(REF, stem + '.pkl'))
    want = H.load_modelset(tag)
    assert got.keys() == want.keys() and got.twobase == want.twobase == H.model_meta()[stem]['is_dict']
    for key in want.keys():
        for name in ('W1', 'b1', 'W2', 'b2'):
            a, b = np.asarray(getattr(got.models[key], name)), np.asarray(getattr(want.models[key], name))
            assert a.shape == b.shape and np.array_equal(a, b), (tag, key, name)
    assert load_model_file(os.path.join(H.MODELS, stem + '.npz')).twobase == want.twobase
