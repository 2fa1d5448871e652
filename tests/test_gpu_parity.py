"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's golden outputs.
Runs on a real MI355X only: `pytest -m gpu`."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from mcaller_amd.device import Device
    d = Device(0)
    yield d
    d.close()


@pytest.fixture(scope='module')
def td(tmp_path_factory):
    return H.testdata_paths(str(tmp_path_factory.mktemp('testdata')))


def device_vs_oracle(dev, P, k, skip_thresh, qual_thresh, modelset, base, train, tail_contig=-1):
    from mcaller_amd import extract_contexts as ec
    rec = ec.compute(P, k, skip_thresh, qual_thresh, modelset, base, train, device=dev, tail_contig=tail_contig)
    orc = H.oracle_records(P.table, P.ref.device_arrays(), P.qual, k, skip_thresh, qual_thresh, tail_contig=tail_contig)
    if not train:
        _, weights, _, soc = ec.submodel_setup(modelset, base)
        H.oracle_score(orc, P.table, P.qual, weights, soc, k)
    H.assert_records_equal(rec, orc, k)
    return rec


@pytest.mark.parametrize('kw,model,skip', [
    (dict(positions='test_positions_m6A.txt'), 'r95', 0),
    (dict(motif='GATC'), 'r95', 0),
    (dict(motif='A'), 'r95', 0),
    (dict(positions='test_positions.txt'), 'r95', 0),
    (dict(motif='GATC'), 'r95', 1),
    (dict(motif='A'), 'r94', 2),
])
def test_testdata_records(dev, td, kw, model, skip):
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    r2q = extract_read_quality(td['fastq'])
    posf = td[kw['positions']] if 'positions' in kw else None
    P = ec.prepare(td['tsv'], td['fasta'], r2q, 0, os.path.getsize(td['tsv']), 'A', kw.get('motif'), posf)
    rec = device_vs_oracle(dev, P, 6, skip, 0.0, H.load_modelset(model), 'A', False)
    assert rec.n > 0


@pytest.mark.parametrize('tag,kw', [
    ('config1_positions_m6A', dict(positions='test_positions_m6A.txt')),
    ('motif_GATC', dict(motif='GATC')),
    ('motif_A', dict(motif='A')),
])
def test_extract_features_dropin_text(td, tag, kw, tmp_path):
    """The drop-in boundary, end to end: same call as mCaller.py:58, output bytes == the reference's."""
    import shutil
    from mcaller_amd.extract_contexts import extract_features
    from mcaller_amd.read_qual import extract_read_quality
    tsv = str(tmp_path / 'masonread1.eventalign.tsv')
    shutil.copy(td['tsv'], tsv)
    r2q = extract_read_quality(td['fastq'])
    posf = td[kw['positions']] if 'positions' in kw else None
    stem = os.path.join(H.GOLDEN, 'models')
    # the model file goes through the restricted unpickler in production; the fixtures carry arrays only, so
    # write a pickle-free stand-in: extract_features accepts the path of an .npz export as well
    model = os.path.join(stem, 'r95_twobase_model_NN_6_m6A.npz')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        extract_features(tsv, td['fasta'], r2q, 6, 0, 0, model, 'NN', 0, endline=os.path.getsize(tsv), train=False,
                         pos_label=None, base='A', motif=kw.get('motif'), positions_list=posf)
    out = open(str(tmp_path / 'masonread1.eventalign.diffs.6.tmp0')).read()
    assert out == open(os.path.join(H.GOLDEN, 'ref_outputs', tag + '.diffs.6')).read()
    ref_stdout = open(os.path.join(H.GOLDEN, 'ref_outputs', tag + '.stdout')).read().split('\n')
    for line in buf.getvalue().split('\n'):
        if line.strip():
            assert line in ref_stdout


def test_micro_cases_records(dev, tmp_path):
    """Every committed micro-case: HIP records == oracle records (regular AND quirk cases)."""
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    from mcaller_amd._lib import McError
    n_ok = n_irregular = 0
    bad = []
    for case in H.micro_cases():
        d = tmp_path / ('c%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        a = case['args']
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                r2q = extract_read_quality(paths['fastq'])
                P = ec.prepare(paths['tsv'], paths['fasta'], r2q, 0, os.path.getsize(paths['tsv']), a['base'],
                               a['motif'], paths['positions'])
        except BaseException:
            continue                      # input errors are host-side paths (covered by the CPU suite)
        modelset = None if a['train'] else H.load_modelset(a['model'])
        if modelset is not None and a['model'] in ('CAAY', 'CRAA') and a['base'] == 'A':
            pass
        try:
            device_vs_oracle(dev, P, a['k'], a['skip_thresh'], a['qual_thresh'], modelset, a['base'], a['train'])
            n_ok += 1
        except McError as e:
            bad.append((case['seed'], case['flavour'], str(e)))
        except AssertionError as e:
            bad.append((case['seed'], case['flavour'], str(e)[:200]))
    print('micro-cases: %d identical' % n_ok)
    assert not bad, bad[:10]
    assert n_ok > 250


@pytest.mark.parametrize('n_rows,seed,motif,skip,qthresh', [
    (1000000, 1, 'GATC', 0, 0.0),
    (1000000, 2, 'GATC', 1, 9.0),
    (300000, 3, 'A', 0, 0.0),
    (300000, 4, 'AT', 2, 0.0),
    (5000, 5, 'GATC', 0, 0.0),
    (4097, 6, 'A', 0, 0.0),
])
def test_synthetic_records(dev, n_rows, seed, motif, skip, qthresh):
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=1000000, seed=11)
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(n_rows, seed=seed, codes=codes)

    class P(object):
        pass
    P.table, P.ref, P.qual = table, ref, qual
    for tail in (-1, 0):
        rec = device_vs_oracle(dev, P, 6, skip, qthresh, H.load_modelset('r95'), 'A', False, tail_contig=tail)
    assert rec.n > 0


def test_mlp_known_answers(dev):
    """K2 alone against scikit-learn's predict_proba captured in the build container (models_meta.json)."""
    from mcaller_amd.extract_contexts import submodel_setup
    meta = H.model_meta()
    for tag, stem in H.MODEL_STEMS.items():
        ms = H.load_modelset(tag)
        keys = ms.keys()
        dev.set_mlp([ms.models[key] for key in keys], np.zeros(256, dtype=np.uint8))
        X = np.array(meta[stem]['probes'], dtype=np.float64)
        for i, key in enumerate(keys):
            p = dev.mlp_forward(X, np.full(len(X), i, dtype=np.uint8))
            want = np.array(meta[stem]['known_answers'][key])
            assert np.abs(p - want).max() < 1e-12, (tag, key, np.abs(p - want).max())


@pytest.mark.parametrize('flavour,n', [('quirk_pal', 150), ('plain', 60), ('dense', 60), ('skips', 60), ('heavy', 40),
                                       ('multi_contig', 40), ('qual', 40), ('quirk_names', 80), ('quirk_flip', 80),
                                       ('quirk_backwards', 80), ('quirk_pos0', 80), ('header', 30), ('n_context', 40)])
def test_fresh_random_cases(dev, tmp_path, flavour, n):
    """Random cases that are NOT in the committed fixtures (seeds >= 10^6): HIP records == C-oracle records."""
    from oracle import casegen
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    from mcaller_amd._lib import McError
    n_ok = n_lit = 0
    for i in range(n):
        case = casegen.gen_case(1000000 + 1000 * (sum(map(ord, flavour)) % 97) + i, flavour=flavour)
        d = tmp_path / ('f%d' % i)
        d.mkdir()
        paths = H.materialise(case, str(d))
        a = case['args']
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                r2q = extract_read_quality(paths['fastq'])
                P = ec.prepare(paths['tsv'], paths['fasta'], r2q, 0, os.path.getsize(paths['tsv']), a['base'],
                               a['motif'], paths['positions'])
        except BaseException:
            continue
        try:
            device_vs_oracle(dev, P, a['k'], a['skip_thresh'], a['qual_thresh'],
                             None if a['train'] else H.load_modelset(a['model']), a['base'], a['train'])
            n_ok += 1
        except McError as e:
            raise AssertionError('seed %d: %s' % (case['seed'], e))
        except AssertionError as e:
            raise AssertionError('seed %d (%s): %s' % (case['seed'], flavour, str(e)[:300]))
    print('%s: %d identical' % (flavour, n_ok))
    assert n_ok >= n // 2
