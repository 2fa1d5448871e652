"""`-c LR` / `-c NBC` (train_model.py:55-60; mCaller.py:137): the model files are read without scikit-learn, and the forward --
the numpy oracle on the CPU, k3_simple on the GPU -- reproduces scikit-learn's predict_proba captured by
tests/golden/make_golden.py (simple_meta.json) to 1e-12; inside the hot path the records an LR / NBC model file scores equal the
oracle's."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import clf_oracle


def meta():
    return json.load(open(os.path.join(H.GOLDEN, 'models', 'simple_meta.json')))


def modelset(tag):
    from mcaller_amd.model_io import load_model_file
    return load_model_file(os.path.join(H.GOLDEN, 'models', 'simple_twobase_model_%s_6_m6A.pkl' % tag))


@pytest.mark.parametrize('tag,kind', [('LR', 'logistic'), ('NBC', 'gnb')])
def test_model_files_load_without_sklearn_and_the_oracle_matches_known_answers(tag, kind):
    ms = modelset(tag)
    assert ms.twobase and ms.keys() == ['MG', 'MH'] and all(w.kind == kind and w.n_in == 7 for w in ms.models.values())
    m = meta()[tag]
    X = np.array(m['probes'])
    models = [ms.models[k] for k in ms.keys()]
    for i, key in enumerate(ms.keys()):
        p = clf_oracle.forward(models, X, np.full(len(X), i, dtype=np.uint8))
        want = np.array(m['known_answers'][key])
        assert np.abs(p - want).max() <= 1e-12, (key, np.abs(p - want).max())
        assert 0.0 <= p.min() and p.max() <= 1.0 and (p > 0.5).any() and (p < 0.5).any()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['LR', 'NBC'])
def test_kernel_matches_known_answers_and_scores_records(tag, tmp_path):
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    ms = modelset(tag)
    m = meta()[tag]
    X = np.array(m['probes'])
    _, models, _, soc = submodel_setup(ms, 'A')
    dev = Device(0)
    dev.set_classifier(models, soc)
    for i, key in enumerate(ms.keys()):
        p = dev.classifier_forward(X, np.full(len(X), i, dtype=np.uint8))
        want = np.array(m['known_answers'][key])
        assert np.abs(p - want).max() <= 1e-12, (key, np.abs(p - want).max())
    # inside the hot path: one pass at a time and pipelined (mc_extract_features_async), records == oracle
    codes = synth.genome(length=300000, seed=4)
    ref = synth.SynthRef(codes)
    table, qual = synth.make_table(300000, seed=8, codes=codes)
    dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual)
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, models, soc, 6)
    rec = dev.extract(6, 0, 0.0)
    H.assert_records_equal(rec, orc, 6, prob_tol=1e-12)
    assert np.isfinite(rec.prob[:rec.n]).sum() > 100
    dev.run_async(6, 0, 0.0)
    H.assert_records_equal(dev.wait(), orc, 6, prob_tol=1e-12)
    dev.close()


@pytest.mark.gpu
def test_cli_with_an_lr_model_file(tmp_path):
    """`-c LR -d <LR model file>` end to end: the rows' labels and probabilities are the LR model's (oracle: the numpy
    restatement on the features the file prints), the features those of the NN run."""
    import contextlib
    import io
    from mcaller_amd import synth, mCaller
    codes = synth.genome(length=200000, seed=27)
    table, qual = synth.make_table(250000, seed=11, codes=codes, read_len=(1500, 6000))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    lr = os.path.join(H.GOLDEN, 'models', 'simple_twobase_model_LR_6_m6A.pkl')
    with contextlib.redirect_stdout(io.StringIO()):
        mCaller.main(['-m', 'GATC', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', lr, '-c', 'LR'])
    rows = [l.split('\t') for l in open(paths['tsv'][:-4] + '.diffs.6').read().splitlines()]
    assert len(rows) > 100
    ms = modelset('LR')
    n_checked = 0
    for r in rows:
        ctx, feats, label, prob = r[3], [float(v) for v in r[4].split(',')], r[6], float(r[7])
        key = 'MG' if ctx[5:7] == 'MG' else 'MH'
        w = ms.models[key]
        p = clf_oracle.logistic_proba(w.coef, w.intercept, np.array([feats]))[0]
        assert abs(float(np.round(p, 2)) - prob) < 1e-9 and label == ('m6A' if p >= 0.5 else 'A'), r
        n_checked += 1
    assert n_checked == len(rows)
