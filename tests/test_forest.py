"""Random-forest classifier path (BASELINE config 5): the model file is read without scikit-learn,
and the forest forward (C oracle on CPU, k3_forest on the GPU) reproduces scikit-learn's predict_proba captured by
tests/golden/make_golden.py (rf_meta.json) bit for bit."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H


def meta():
    return json.load(open(os.path.join(H.GOLDEN, 'models', 'rf_meta.json')))


def test_forest_file_loads_without_sklearn_and_oracle_matches_known_answers():
    import sys
    ms = H.load_rf_modelset()
    assert ms.twobase and ms.keys() == ['MG', 'MH']
    assert all(w.kind == 'forest' and w.n_trees == 50 and w.n_in == 7 for w in ms.models.values())
    m = meta()
    X = np.array(m['probes'])
    forests = [ms.models[k] for k in ms.keys()]
    for i, key in enumerate(ms.keys()):
        p = H.oracle_forest_forward(forests, X, np.full(len(X), i, dtype=np.uint8))
        assert np.array_equal(p, np.array(m['known_answers'][key])), key


@pytest.mark.gpu
def test_forest_kernel_matches_known_answers_and_scores_records():
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    ms = H.load_rf_modelset()
    m = meta()
    X = np.array(m['probes'])
    _, forests, _, soc = submodel_setup(ms, 'A')
    dev = Device(0)
    dev.set_classifier(forests, soc)
    for i, key in enumerate(ms.keys()):
        p = dev.classifier_forward(X, np.full(len(X), i, dtype=np.uint8))
        assert np.array_equal(p, np.array(m['known_answers'][key])), key
    # inside the hot path: records scored by the forest == oracle
    codes = synth.genome(length=300000, seed=4)
    ref = synth.SynthRef(codes)
    table, qual = synth.make_table(300000, seed=8, codes=codes)
    dev.set_reference(ref.device_arrays()); dev.upload_table(table); dev.set_read_quality(qual)
    rec = dev.extract(6, 0, 0.0)
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, forests, soc, 6)
    H.assert_records_equal(rec, orc, 6, prob_tol=0.0)
    assert np.isfinite(rec.prob[:rec.n]).sum() > 100
    dev.close()
