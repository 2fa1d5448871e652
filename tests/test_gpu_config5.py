"""BASELINE config 5 as ONE workload: `--train` on labelled positions + the RF classifier, 10^7 events.
Positions mode with labels on a 10^7-row synthetic file: the train dicts and the `.train` rows of the streamed HIP path equal
what the C oracle's records give; the forest's probabilities through the pipelined interface are bit-equal to the oracle's; and
the streamed predict-mode file (RF model) equals the oracle's rows.  Runs on a real MI355X only: `pytest -m gpu`."""
import contextlib
import io
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_config5_train_dicts_and_forest_probabilities_at_1e7_rows(tmp_path, monkeypatch):
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    n_rows = 10000000
    codes = synth.genome()
    table, qual = synth.make_table(n_rows, seed=55, codes=codes)
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    del table
    # labelled positions: the A of every GATC on '+' and on '-' (the T of the forward strand), a third of them 'm6A'
    seq = np.frombuffer(synth.codes_to_str(codes).encode('ascii'), dtype=np.uint8)
    hit = np.flatnonzero((seq[:-3] == ord('G')) & (seq[1:-2] == ord('A')) & (seq[2:-1] == ord('T')) & (seq[3:] == ord('C')))
    posfile = str(tmp_path / 'positions.txt')
    with open(posfile, 'w') as fh:
        for p in hit:
            fh.write('ecoli_syn\t%d\t+\t%s\n' % (p + 1, 'm6A' if (p + 1) % 3 == 0 else 'A'))
            fh.write('ecoli_syn\t%d\t-\t%s\n' % (p + 2, 'm6A' if (p + 2) % 3 == 0 else 'A'))
    pos_label = H.pos2label(posfile)
    r2q = extract_read_quality(paths['fastq'])
    size = os.path.getsize(paths['tsv'])
    assert size > 1000000000

    # ---- the oracle: host parser -> C oracle's records -> the reference's per-record transcription (Finisher) ----
    with contextlib.redirect_stdout(io.StringIO()):
        P = ec.prepare(paths['tsv'], paths['fasta'], r2q, 0, size, 'A', None, posfile)
    assert P.fatal is None and P.table.n_rows == n_rows
    arrays = P.ref.device_arrays()
    orc = H.oracle_records(P.table, arrays, P.qual, 6, 0, 0.0)
    want = ec.Finisher(P, 6, 'A', True, pos_label=pos_label)
    with contextlib.redirect_stdout(io.StringIO()):
        assert want.run(orc) is None
    n_obs = want.num_observations
    assert n_obs > 5000 and sum(len(v) for v in want.signals['general'].values()) == n_obs

    # ---- --train: the file streamed through the GPU in shards, features only ----
    streamed = []
    real = ec.stream_features

    def spy(*a, **kw):
        res = real(*a, **kw)
        streamed.append((kw.get('train'), res.n_rows))
        return res
    monkeypatch.setattr(ec, 'stream_features', spy)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sig, ctx = ec.extract_features(paths['tsv'], paths['fasta'], r2q, 6, 0, 0.0, None, 'NN', 0, endline=size, train=True,
                                       pos_label=pos_label, base='A', motif=None, positions_list=posfile)
    assert streamed == [(True, n_rows)]                          # the shards really went through stream_features
    tmp = paths['tsv'][:-4] + '.diffs.6.train.tmp0'
    assert open(tmp, 'rb').read() == want.text()
    os.remove(tmp)
    assert sig == want.signals and ctx == want.contexts          # (floats: exactly the same doubles)
    assert [l for l in buf.getvalue().splitlines() if 'observations' in l or 'positions' in l or 'regions' in l] == want.counters()[1:]

    # ---- the forest through the pipelined interface: probabilities bit-equal ----
    ms = H.load_rf_modelset()
    _, forests, _, soc = ec.submodel_setup(ms, 'A')
    H.oracle_score(orc, P.table, P.qual, forests, soc, 6)
    dev = ec.get_device()
    dev.set_reference(arrays)
    dev.set_classifier(forests, soc)
    slot = dev.upload_table_async(P.table, P.qual)
    dev.run_async(6, 0, 0.0)
    dev.run_async(6, 0, 0.0)
    H.assert_records_equal(dev.wait(), orc, 6, prob_tol=0.0)
    H.assert_records_equal(dev.wait(), orc, 6, prob_tol=0.0)
    assert np.isfinite(orc.prob[:orc.n]).sum() == n_obs

    # ---- predict mode with the RF model file: the streamed file equals the oracle's rows ----
    # (the expected TEXT is made from the oracle's records by the product's own row formatter, Finisher / mc_format_diffs: what is
    # compared with the oracle here are the records -- above, bit for bit -- and that the streamed CLI path writes the rows of exactly
    # those records; the formatter itself is pinned to the reference's bytes on the CPU side, tests/test_host_pipeline.py and
    # test_extract_features_dropin_text)
    fin = ec.Finisher(P, 6, 'A', False, modelset=ms, device=dev)
    with contextlib.redirect_stdout(io.StringIO()):
        assert fin.run(orc) is None
    del streamed[:]
    rf_file = os.path.join(H.GOLDEN, 'models', 'rf_twobase_model_RF_6_m6A.pkl')
    with contextlib.redirect_stdout(io.StringIO()):
        ec.extract_features(paths['tsv'], paths['fasta'], r2q, 6, 0, 0.0, rf_file, 'RF', 0, endline=size, train=False,
                            base='A', motif=None, positions_list=posfile)
    assert streamed == [(False, n_rows)]
    tmp = paths['tsv'][:-4] + '.diffs.6.tmp0'
    assert open(tmp, 'rb').read() == fin.text()
