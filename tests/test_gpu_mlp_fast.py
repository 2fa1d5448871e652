"""The MLP's fast forward (k2_mlp<.., true>, mcaller_amd/csrc/mc_classify.hip): the hidden layer in fp32, and fp64 again for every
record whose fast probability lies within its error bound of a threshold the reference's row depends on (the label p >= 0.5 and
the ties of np.round(p, 2), extract_contexts.py:200-207).  What must hold: the printed label and probability of EVERY record equal
the fp64 oracle's; the raw probability is within 1e-6 (north_star allows 1e-5).  Runs on a real MI355X only."""
import ctypes as C

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def setup(motif, n_rows, seed, model='r95_twobase_model_NN_6_m6A', read_len=None):
    from mcaller_amd import synth
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif=motif)
    kw = dict(read_len=read_len) if read_len else {}
    table, qual = synth.make_table(n_rows, seed=seed, codes=codes, **kw)
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model(model)), 'A')
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    return table, qual, arrays, weights, soc, orc


def test_tanh32_error_bound_by_exhaustion():
    """|tanh32s(s) - tanh(s ln2 / 2)| over all 2^32 floats s: what K2_TANH32_MAX_ERR (mc_dev.h: 2.5e-7) has to cover."""
    from mcaller_amd import _lib
    from mcaller_amd.device import Device
    d = Device(0)
    try:
        L = _lib.lib()
        out = C.c_double(0.0)
        assert L.mc_debug_tanh32_max_err(C.byref(out)) == 0
        print('largest |tanh32s - tanh| over all floats: %.4g' % out.value)
        assert 0.0 < out.value <= 2.5e-7
    finally:
        d.close()


@pytest.mark.parametrize('motif,n_rows', [('A', 10000000), ('GATC', 20000000)])
def test_fast_forward_prints_what_fp64_prints(motif, n_rows):
    """Dense (-m A: ~9 10^5 calls of 10^7 rows) and sparse tables through the pipelined interface and the synchronous one:
    labels and printed probabilities of every record equal the oracle's, raw probabilities within 1e-6."""
    from mcaller_amd.device import Device
    table, qual, arrays, weights, soc, orc = setup(motif, n_rows, 77)
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(6, 0, 0.0, score=True)
        H.assert_records_equal(d.wait(), orc, 6, prob_tol=1e-6)
        print('%s: %d scored records, largest |dp| %.3g' % (motif, int(np.isfinite(orc.prob[:orc.n]).sum()), H.assert_records_equal.last_prob_err))
        rec = d.extract(6, 0, 0.0, score=True)
        H.assert_records_equal(rec, orc, 6, prob_tol=1e-6)
    finally:
        d.close()


def test_fp64_behind_a_knob(monkeypatch):
    """MCALLER_MLP_FP64=1: every record in fp64, as before -- within 1e-9 of the oracle."""
    from mcaller_amd.device import Device
    table, qual, arrays, weights, soc, orc = setup('A', 1000000, 78)
    monkeypatch.setenv('MCALLER_MLP_FP64', '1')
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(6, 0, 0.0, score=True)
        H.assert_records_equal(d.wait(), orc, 6, prob_tol=1e-9)
        assert H.assert_records_equal.last_prob_err <= 1e-9
    finally:
        d.close()


def test_fast_forward_on_inputs_far_from_the_training_range():
    """Slot means of tens of pA (events far from the model: what a mis-aligned read looks like) make the fp32 bound wide: more
    records go through fp64, none prints differently.  Synthetic rows with the event column scaled."""
    from mcaller_amd import synth, _lib
    from mcaller_amd.device import Device
    table, qual, arrays, weights, soc, _ = setup('A', 300000, 79)
    ev = table.event_e4.astype(np.int64)
    mu = table.model_e4.astype(np.int64)
    table.event_e4[:] = (mu + (ev - mu) * 25).astype(np.int32)          # differences 25 times as large
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(6, 0, 0.0, score=True)
        H.assert_records_equal(d.wait(), orc, 6, prob_tol=1e-6)
    finally:
        d.close()


@pytest.mark.parametrize('k,hidden,n_models', [(4, 37, 3), (8, 100, 4), (5, 1, 2), (7, 64, 1), (6, 51, 4)])
def test_fast_forward_other_shapes(k, hidden, n_models):
    """The general instance (k2_mlp<0, true>: the number of inputs at run time) and the shapes the pairs of units have to get right:
    an odd hidden layer (a padding unit of zeros), a layer of one unit, quarters of uneven size, one to four sub-models -- random
    weights of the size a trained model has, a dense table, labels and printed probabilities as the fp64 oracle's."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.model_io import MLPWeights
    rng = np.random.default_rng(1000 * k + hidden)
    codes = synth.genome(length=300000, seed=k)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(400000, seed=500 + k, codes=codes)
    arrays = ref.device_arrays()
    weights = [MLPWeights(rng.normal(0, 0.6, (k + 1, hidden)), rng.normal(0, 0.5, hidden), rng.normal(0, 0.8, hidden), rng.normal(0, 0.3, 1))
               for _ in range(n_models)]
    soc = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate('ACGTM'):
        soc[ord(c)] = i % n_models
    orc = H.oracle_records(table, arrays, qual, k, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, k)
    assert int(np.isfinite(orc.prob[:orc.n]).sum()) > 20000
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(k, 0, 0.0, score=True)
        H.assert_records_equal(d.wait(), orc, k, prob_tol=1e-6)
        assert d.last_pass_info()[0] > 0                         # (the fused dense pass: the stretches made of pieces)
        rec = d.extract(k, 0, 0.0, score=True)                   # (the synchronous path: the scan + emit pair, stretches of records)
        H.assert_records_equal(rec, orc, k, prob_tol=1e-6)
    finally:
        d.close()
