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
    stem = H.MODELS
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


@pytest.mark.parametrize('n_in,n_hidden,n_models', [(7, 100, 4), (7, 100, 1), (7, 10, 8), (7, 3, 2), (5, 37, 3), (9, 128, 2), (1, 1, 1)])
def test_mlp_shapes_and_sizes(dev, n_in, n_hidden, n_models):
    """K2 (one lane per record, a quarter of the hidden units per wave, lists per sub-model) against a plain numpy fp64
    forward: every record count around the kernel's group (64) and stretch (1024) sizes, mixed sub-models, rows whose
    sub-model is not in the model (they stay NaN, the host's KeyError path), hidden layers that do not divide by four."""
    from mcaller_amd.model_io import MLPWeights
    rng = np.random.default_rng(n_in * 1000 + n_hidden * 10 + n_models)
    ws = [MLPWeights(rng.normal(0, 1.5, (n_in, n_hidden)), rng.normal(0, 1, n_hidden), rng.normal(0, 1.5, (n_hidden, 1)),
                     rng.normal(0, 1, 1)) for _ in range(n_models)]
    dev.set_mlp(ws, np.zeros(256, dtype=np.uint8))
    for n in (1, 2, 63, 64, 65, 127, 129, 1023, 1024, 1025, 2049, 70001):
        X = rng.normal(0, 2, (n, n_in))
        X[rng.random(n) < 0.02] *= 200.0                       # saturated units, |activation| far beyond the clamp
        if n > 3:
            X[1] = 0.0
        sub = rng.integers(0, n_models, n).astype(np.uint8)
        if n > 10:
            sub[rng.random(n) < 0.3] = 255                     # not scored
            sub[5] = n_models                                  # a key the model does not hold
        p = dev.mlp_forward(X, sub)
        want = np.full(n, np.nan)
        for m in range(n_models):
            rows = sub == m
            h = np.tanh(X[rows] @ ws[m].W1 + ws[m].b1)
            z = h @ ws[m].W2.reshape(-1) + ws[m].b2[0]
            want[rows] = 1.0 / (1.0 + np.exp(-z))
        assert np.array_equal(np.isnan(p), np.isnan(want)), (n, int(np.isnan(p).sum()), int(np.isnan(want).sum()))
        ok = ~np.isnan(want)
        if ok.any():
            assert np.abs(p[ok] - want[ok]).max() < 1e-12, (n, np.abs(p[ok] - want[ok]).max())
    with pytest.raises(Exception):
        dev.set_mlp(ws * 9, np.zeros(256, dtype=np.uint8))     # more sub-models than the kernel lists


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


def test_pipelined_passes_equal_the_synchronous_path(dev, tmp_path):
    """mc_extract_features_async / mc_wait_records: two passes in flight, records exported by a kernel into pinned host
    memory -- identical to mc_extract_features, for different parameters per pass, and for passes the fast path cannot
    finish alone (irregular reads: the micro-cases), which fall back inside mc_wait_records."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=600000, seed=13)
    ref = synth.SynthRef(codes, motif='GATC')
    table, qual = synth.make_table(700000, seed=31, codes=codes)
    modelset = H.load_modelset('r95')
    _, weights, _, soc = ec.submodel_setup(modelset, 'A')
    dev.set_reference(ref.device_arrays())
    dev.upload_table(table)
    dev.set_read_quality(qual)
    dev.set_mlp(weights, soc)
    params = [(6, 0, 0.0), (6, 1, 0.0), (6, 0, 9.0), (5, 2, 0.0), (6, 0, 0.0)]
    sync = []
    for k, skip, q in params:
        if k != 6:
            sync.append(None)
            continue
        sync.append(dev.extract(k, skip, q))
    got = []
    dev.run_async(*params[0])
    for i in range(1, len(params)):
        if params[i][0] == 6:
            dev.run_async(*params[i])
        else:
            dev.run_async(params[i][0], params[i][1], params[i][2], score=False)      # features only: k+1 != model inputs
        r = dev.wait().by_record()
        got.append((r.n, r.feats[:r.n * r.k].copy(), r.site_pos[:r.n].copy(), r.info[:r.n].copy(), r.prob[:r.n].copy(),
                    r.close_row[:r.n].copy(), r.site_seg[:r.n].copy()))
    r = dev.wait().by_record()
    got.append((r.n, r.feats[:r.n * r.k].copy(), r.site_pos[:r.n].copy(), r.info[:r.n].copy(), r.prob[:r.n].copy(),
                r.close_row[:r.n].copy(), r.site_seg[:r.n].copy()))
    for (k, skip, q), s_rec, g in zip(params, sync, got):
        if s_rec is None:
            orc = H.oracle_records(table, ref.device_arrays(), qual, k, skip, q)
            assert g[0] == orc.n and (g[1] == orc.feats[:orc.n * k]).all() and (g[2] == orc.site_pos[:orc.n]).all()
            continue
        n = s_rec.n
        assert g[0] == n > 100
        assert (g[1] == s_rec.feats[:n * k]).all() and (g[2] == s_rec.site_pos[:n]).all() and (g[3] == s_rec.info[:n]).all()
        assert (g[5] == s_rec.close_row[:n]).all() and (g[6] == s_rec.site_seg[:n]).all()
        assert np.array_equal(g[4], s_rec.prob[:n], equal_nan=True)
    # irregular reads: the pass is redone by the synchronous path inside wait()
    n_checked = 0
    for case in H.micro_cases()[:60]:
        if case['expected']['outcome'] != 'ok' or case['args']['train']:
            continue
        d = tmp_path / ('p%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        from mcaller_amd.read_qual import extract_read_quality
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                P = ec.prepare(paths['tsv'], paths['fasta'], extract_read_quality(paths['fastq']), 0,
                               os.path.getsize(paths['tsv']), case['args']['base'], case['args']['motif'], paths['positions'])
        except BaseException:
            continue
        if P.fatal is not None or P.table.n_rows == 0:
            continue
        a = case['args']
        ms = H.load_modelset(a['model'])
        _, w, _, so = ec.submodel_setup(ms, a['base'])
        if w[0].n_in != a['k'] + 1:
            continue
        dev.set_reference(P.ref.device_arrays())
        dev.upload_table(P.table)
        dev.set_read_quality(P.qual)
        dev.set_mlp(w, so)
        want = dev.extract(a['k'], a['skip_thresh'], a['qual_thresh'])
        dev.run_async(a['k'], a['skip_thresh'], a['qual_thresh'])
        dev.run_async(a['k'], a['skip_thresh'], a['qual_thresh'])
        for _ in range(2):
            r = dev.wait().by_record()
            assert r.n == want.n
            assert (r.feats[:r.n * r.k] == want.feats[:want.n * want.k]).all() and (r.info[:r.n] == want.info[:want.n]).all()
            assert np.array_equal(r.prob[:r.n], want.prob[:want.n], equal_nan=True)
        n_checked += 1
    assert n_checked > 20


@pytest.mark.parametrize('flavour', ['largest', 'mixed', 'equal_events'])
def test_packed_slot_means_at_the_edges_of_the_32_bit_form(dev, flavour):
    """mc_wait_records sends a slot mean as the integer d where it is fl(d / 10^4) (mc_calls_view.feats_lo32) and as two
    32-bit halves where it is not: currents at the edge of what a table may hold (|value| < 10^9, differences up to 2*10^9),
    ordinary ones next to them, and slots whose events are all equal (a mean of several events that is fl(d / 10^4)) -- the
    unpacked means are the oracle's doubles bit for bit."""
    from mcaller_amd import synth, _lib
    codes = synth.genome(length=300000, seed=17)
    ref = synth.SynthRef(codes, motif='GATC')
    table, qual = synth.make_table(400000, seed=71, codes=codes)
    rng = np.random.default_rng(5)
    n = table.n_rows
    ev, mu = table.evmu[:, 0].astype(np.int64), table.evmu[:, 1].astype(np.int64)
    keep_n = mu == 0                                        # (NNNNNN rows keep their model current of 0)
    top = 10 ** 9 - 1
    if flavour == 'largest':                                # |event - model| up to 2e9 - 2
        ev = rng.choice([top, top - 1, top - 7, 999999000], size=n) * rng.choice([-1, 1], size=n)
        mu = np.where(keep_n, 0, -np.sign(ev) * rng.choice([top, top - 1, top - 3, 999990000], size=n))
    elif flavour == 'mixed':                                # one row in three far out
        far = rng.random(n) < 0.33
        ev = np.where(far, rng.integers(-top, top, size=n), ev)
    else:                                                   # every event of a position the same: means of equal numbers
        ev = np.where(keep_n, ev, mu + (table.pos.astype(np.int64) * 7919 % 40001 - 20000) * 100)
    table.evmu[:, 0] = ev
    table.evmu[:, 1] = mu
    dev.set_reference(ref.device_arrays())
    dev.upload_table(table)
    dev.set_read_quality(qual)
    for skip in (0, 2):
        want = H.oracle_records(table, ref.device_arrays(), qual, 6, skip, 0.0)
        dev.run_async(6, skip, 0.0, score=False)
        dev.run_async(6, skip, 0.0, score=False)
        for _ in range(2):
            raw = dev.wait()
            lo, hi, wide = raw._packed
            got = raw.by_record()
            assert got.n == want.n > 500
            assert (got.info[:got.n] == want.info[:want.n]).all() and (got.site_pos[:got.n] == want.site_pos[:want.n]).all()
            a, b = got.feats[:got.n * 6].view(np.uint64), want.feats[:want.n * 6].view(np.uint64)
            kept = np.repeat((want.info[:want.n] & _lib.I_TOO_MANY) == 0, 6)
            assert (a[kept] == b[kept]).all()
        n_slots = len(lo)
        print('%s, skip %d: %d of %d slot means travel wide' % (flavour, skip, len(hi), n_slots))
        if flavour == 'equal_events':                       # (x + x + x) / 3 is not always x: a few are left
            assert len(hi) < 0.05 * n_slots
        else:
            assert 0 < len(hi) < n_slots


@pytest.mark.parametrize('n_rows,motif,score', [
    (100000000, 'GATC', True),      # BASELINE.json configs[2]: 10^8 events, -m GATC, NN classifier
    (10000000, 'A', False),         # configs[4]-sized feature-matrix build (--train: features only), dense sites
])
def test_full_size_records_equal_the_oracle(n_rows, motif, score):
    """The headline workload at its full size: every flush record of the HIP path (through the pipelined interface,
    as bench.py runs it) equals the C oracle's -- integers and slot means bit for bit, probabilities to 1e-9."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(n_rows, seed=1000, codes=codes)
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    dev = Device(0)
    try:
        dev.set_reference(ref.device_arrays())
        dev.upload_table(table)
        dev.set_read_quality(qual)
        dev.set_mlp(weights, soc)
        dev.run_async(6, 0, 0.0, score=score)
        dev.run_async(6, 0, 0.0, score=score)
        first = dev.wait()                    # (compacted views: slot means / probabilities of the calls only)
        keep = (first.n, first.feats[:first.n_calls * 6].copy(), first.prob[:first.n_calls].copy())
        rec = dev.wait()
        assert rec.n == keep[0] and (rec.feats[:rec.n_calls * 6] == keep[1]).all()    # passes are reproducible
        assert np.array_equal(rec.prob[:rec.n_calls], keep[2], equal_nan=True)
        orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
        if score:
            H.oracle_score(orc, table, qual, weights, soc, 6)
        else:
            rec.prob[:rec.n] = np.nan
            orc.prob[:orc.n] = np.nan
        H.assert_records_equal(rec, orc, 6)
        assert rec.n > n_rows // 1000
    finally:
        dev.close()


@pytest.mark.parametrize('k', [1, 2, 3, 4, 5, 7, 8])
def test_other_window_lengths(dev, k):
    """-n/--num_variables other than 6 (features only: the shipped models take 7 inputs), with skips, both paths."""
    from mcaller_amd import synth
    codes = synth.genome(length=300000, seed=17)
    ref = synth.SynthRef(codes, motif='GATC' if k % 2 else 'A')
    table, qual = synth.make_table(200000, seed=40 + k, codes=codes, read_len=(800, 5000))
    dev.set_reference(ref.device_arrays())
    dev.upload_table(table)
    dev.set_read_quality(qual)
    for skip in (0, min(k - 1, 2)):
        orc = H.oracle_records(table, ref.device_arrays(), qual, k, skip, 0.0)
        rec = dev.extract(k, skip, 0.0, score=False)
        rec.prob[:rec.n] = np.nan
        H.assert_records_equal(rec, orc, k)
        dev.run_async(k, skip, 0.0, score=False)
        rec2 = dev.wait()
        rec2.prob[:rec2.n] = np.nan
        H.assert_records_equal(rec2, orc, k)
        assert rec.n > 50


@pytest.mark.parametrize('read_len', [(40, 300), (300, 1500)])
def test_one_base_motif_with_short_reads(dev, read_len):
    """A one-base motif over reads of a few hundred events: a 1024-row chunk of the scan holds three and more name blocks (their
    units have no mask words there: every row of theirs is looked at out of line), the emit's pieces more blocks than their
    table holds.  First pass (validating), second, third, and the validating pass again with two in flight: the oracle's
    records every time."""
    from mcaller_amd import synth
    codes = synth.genome(length=120000, seed=41)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(400000, seed=4100 + read_len[0], codes=codes, read_len=read_len)
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    assert orc.n > 30000
    dev.set_reference(arrays)
    dev.upload_table(table)
    dev.set_read_quality(qual)
    rec = dev.extract(6, 0, 0.0, score=False)
    rec.prob[:rec.n] = np.nan
    H.assert_records_equal(rec, orc, 6)
    slot = dev.current_slot()
    for again in range(4):
        if again == 2:
            dev.select_table(slot, as_new=True)
        dev.run_async(6, 0, 0.0, score=False)
        if again == 2:
            continue
        for _ in range(2 if again == 3 else 1):
            rec2 = dev.wait()
            rec2.prob[:rec2.n] = np.nan
            H.assert_records_equal(rec2, orc, 6)


def test_empty_and_tiny_tables(dev):
    """No rows, one row, fewer rows than a window: no records, no crash, both paths."""
    from mcaller_amd import synth, _lib
    codes = synth.genome(length=50000, seed=2)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(3000, seed=9, codes=codes, read_len=(500, 900))
    dev.set_reference(ref.device_arrays())
    dev.set_read_quality(qual)
    for n_seg, n_rows in ((0, 0), (1, 1), (1, 3), (1, 40)):
        sub = table.slice_segments(0, n_seg)
        if n_seg:
            r1 = min(n_rows, sub.n_rows)
            sub = _lib.Table(sub.pos[:r1], sub.event_e4[:r1], sub.model_e4[:r1], sub.event_idx[:r1], sub.flags[:r1],
                             np.array([0, r1], dtype=np.int64), sub.seg_read[:1], sub.seg_contig[:1], sub.n_reads,
                             read_names=sub.read_names)
        dev.upload_table(sub)
        orc = H.oracle_records(sub, ref.device_arrays(), qual, 6, 0, 0.0)
        rec = dev.extract(6, 0, 0.0, score=False)
        assert rec.n == orc.n
        dev.run_async(6, 0, 0.0, score=False)
        assert dev.wait().by_record().n == orc.n
        if orc.n:
            rec.prob[:rec.n] = np.nan
            H.assert_records_equal(rec, orc, 6)


def test_record_buffers_too_small_are_grown(monkeypatch):
    """A record capacity guess that is far too low (forced here): the synchronous path repeats the pass with what it
    needed, the pipelined path re-runs the pass inside wait(); records equal the oracle either way."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    monkeypatch.setenv('MCALLER_RECORD_CAPACITY', '1000')
    codes = synth.genome(length=200000, seed=23)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(300000, seed=77, codes=codes, read_len=(1000, 4000))
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    assert orc.n > 20000
    for pipelined in (False, True):
        dev = Device(0)
        try:
            dev.set_reference(ref.device_arrays())
            dev.upload_table(table)
            dev.set_read_quality(qual)
            if pipelined:
                dev.run_async(6, 0, 0.0, score=False)
                dev.run_async(6, 0, 0.0, score=False)
                recs = [dev.wait(), dev.wait()]
            else:
                recs = [dev.extract(6, 0, 0.0, score=False)]
            for rec in recs:
                rec.prob[:rec.n] = np.nan
                orc.prob[:orc.n] = np.nan
                H.assert_records_equal(rec, orc, 6)
        finally:
            dev.close()


@pytest.mark.gpu
def test_copy_outs_queued_back_to_back():
    """bench.py's loop: the copy-out of pass i+1 is started (mc_wait_records_begin) before pass i is waited for, with a new
    pass enqueued in between; every pass, with its own parameters, equals the oracle; begin() with nothing left is a no-op."""
    from mcaller_amd import synth, extract_contexts as ec
    from mcaller_amd.device import Device
    codes = synth.genome(length=400000, seed=3)
    ref = synth.SynthRef(codes, motif='GATC')
    table, qual = synth.make_table(600000, seed=77, codes=codes, read_len=(1000, 9000))
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    params = [(6, 0, 0.0), (6, 1, 0.0), (6, 2, 8.0), (6, 0, 9.5), (6, 1, 7.0), (6, 0, 0.0), (6, 3, 0.0)]
    want = []
    for k, skip, q in params:
        orc = H.oracle_records(table, ref.device_arrays(), qual, k, skip, q)
        H.oracle_score(orc, table, qual, weights, soc, k)
        want.append(orc)
    dev = Device(0)
    try:
        dev.set_reference(ref.device_arrays())
        dev.upload_table(table)
        dev.set_read_quality(qual)
        dev.set_mlp(weights, soc)
        depth, got = 3, []
        for i in range(depth):
            dev.run_async(*params[i])
        dev.wait_begin()
        for i in range(depth, len(params)):
            dev.wait_begin()                       # the pass behind the oldest one
            dev.wait_begin()                       # and the one behind that (bench.py stops at two; three must work too)
            dev.run_async(*params[i])
            got.append(dev.wait())
            H.assert_records_equal(got[-1], want[len(got) - 1], 6)
        for _ in range(depth):
            dev.wait_begin()
            dev.wait_begin()
            dev.wait_begin()
            dev.wait_begin()                       # nothing left to start: no-op
            got.append(dev.wait())
            H.assert_records_equal(got[-1], want[len(got) - 1], 6)
        assert len(got) == len(params) and want[0].n > 500 and want[0].n != want[2].n   # (the quality threshold drops reads)
        with pytest.raises(Exception):
            dev.wait()                             # no pass in flight
    finally:
        dev.close()


@pytest.mark.gpu
def test_bind_to_the_numa_node_of_the_gpu():
    """mc_bind_to_device_numa_node: -1 (topology unknown, nothing changed) or the node, with the thread bound to its cores."""
    from mcaller_amd.device import Device
    before = os.sched_getaffinity(0)
    try:
        node = Device.bind_host_to_numa_node(0)
        after = os.sched_getaffinity(0)
        assert node >= -1 and len(after) >= 1
        if node < 0:
            assert after == before
        else:
            cpus = set()
            for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
                lo, _, hi = part.partition('-')
                cpus.update(range(int(lo), int(hi or lo) + 1))
            assert after == cpus & set(range(os.cpu_count() or max(cpus) + 1)) or after <= cpus
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.gpu
def test_pass_timing_interval():
    """mc_ctx_set_pass_timing: the timing events go with every n-th pipelined pass; the records do not depend on it."""
    from mcaller_amd import synth, extract_contexts as ec
    from mcaller_amd.device import Device
    codes = synth.genome(length=300000, seed=4)
    ref = synth.SynthRef(codes, motif='GATC')
    table, qual = synth.make_table(400000, seed=78, codes=codes, read_len=(1000, 9000))
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    dev = Device(0)
    try:
        dev.set_reference(ref.device_arrays())
        dev.upload_table(table)
        dev.set_read_quality(qual)
        dev.set_mlp(weights, soc)
        for every, want in ((3, [True, False, False, True, False, False]), (0, [False] * 3), (1, [True] * 3)):
            dev.set_pass_timing(every)
            seen = []
            for _ in want:
                dev.run_async(6, 0, 0.0)
                rec = dev.wait()
                seen.append(dev.last_pass_timed())
                H.assert_records_equal(rec, orc, 6)
                tm = dev.times_ms()
                assert tm['total'] >= 0.0
            if every == 3:                      # the count runs over all passes of the context: any rotation of the pattern
                assert sum(seen) == 2 and seen[:3] == seen[3:]
            else:
                assert seen == want
        with pytest.raises(Exception):
            dev.set_pass_timing(-1)
    finally:
        dev.close()


def test_dense_cluster_in_a_sparse_reference(dev):
    """A GATC-like motif picks the scan instance with the small candidate list; a stretch of (GATC)n in the genome makes every
    unit of a tile a candidate there, the list overflows and the tile takes the exact row-by-row path.  Records == oracle."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=400000, seed=77)
    codes[150000:190000] = np.tile(np.array([2, 0, 3, 1], dtype=np.uint8), 10000)      # GATCGATC...
    ref = synth.SynthRef(codes, motif='GATC')
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    for seed, rl in ((21, (3000, 12000)), (22, (200, 900))):
        table, qual = synth.make_table(400000, seed=seed, codes=codes, read_len=rl)
        dev.upload_table(table)
        dev.set_read_quality(qual)
        for skip in (0, 1):
            orc = H.oracle_records(table, ref.device_arrays(), qual, 6, skip, 0.0)
            H.oracle_score(orc, table, qual, weights, soc, 6)
            rec = dev.extract(6, skip, 0.0, score=True)
            H.assert_records_equal(rec, orc, 6)
            dev.run_async(6, skip, 0.0, score=True)
            H.assert_records_equal(dev.wait(), orc, 6)
        assert rec.n > 3000


def test_reference_layouts_far_apart(dev):
    """The character after the 'M' (it picks the sub-model) is read from the sequence at 32 * (the contig's mask offset) + a
    32-bit difference the name-block descriptor carries (NbDesc.seq_delta); a caller's reference whose mask words and bases lie
    further apart than 32 bits hold takes the two dependent loads through seq_off.  Records and probabilities == oracle."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=300000, seed=52)
    ref = synth.SynthRef(codes, motif='GATC')
    arrays = dict(ref.device_arrays())
    shift = (1 << 26) + 5                       # words in front of the contig's masks: 32 * shift > 2^31
    for name in ('mbits_fwd', 'mbits_rev'):
        arrays[name] = np.ascontiguousarray(np.concatenate([np.zeros(shift, dtype=arrays[name].dtype), arrays[name]]))
    arrays['word_off'] = np.array([shift], dtype=np.int64)
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    table, qual = synth.make_table(300000, seed=93, codes=codes, read_len=(800, 5000))
    dev.set_reference(arrays)
    dev.set_mlp(weights, soc)
    dev.upload_table(table)
    dev.set_read_quality(qual)
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    rec = dev.extract(6, 0, 0.0, score=True)
    H.assert_records_equal(rec, orc, 6)
    dev.run_async(6, 0, 0.0, score=True)
    H.assert_records_equal(dev.wait(), orc, 6)
    assert rec.n > 300


def test_one_base_motif_with_jumps_in_the_positions(dev):
    """k1_emit_runs keeps the position of a run in sixteen bits and tells the slots of a window from differences between
    neighbouring runs; a read whose positions jump by 30000 or more between two rows (an alignment across a large deletion) takes
    the row-by-row path for the windows at the jump -- also when the jump is a multiple of 65536.  Records == oracle."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=1000000, seed=43)
    ref = synth.SynthRef(codes, motif='A')
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    table, qual = synth.make_table(300000, seed=91, codes=codes, read_len=(600, 2500))
    rng = np.random.default_rng(17)
    jumps = [29990, 29999, 30000, 30001, 40000, 65530, 65536, 65538, 70000, 131072, 196608 + 3]
    n_jumped = 0
    for s in range(table.n_seg):
        a, b = int(table.seg_row_begin[s]), int(table.seg_row_begin[s + 1])
        if b - a < 400:
            continue
        for _ in range(4):                       # up to four jumps per read, somewhere inside it
            at = int(rng.integers(a + 50, b - 50))
            d = jumps[int(rng.integers(0, len(jumps)))]
            if int(table.pos[b - 1]) + d < len(codes) - 8 and table.pos[at] != table.pos[at - 1]:
                table.pos[at:b] += d
                n_jumped += 1
    assert n_jumped > 50
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    dev.upload_table(table)
    dev.set_read_quality(qual)
    for skip in (0, 2):
        orc = H.oracle_records(table, ref.device_arrays(), qual, 6, skip, 0.0)
        H.oracle_score(orc, table, qual, weights, soc, 6)
        rec = dev.extract(6, skip, 0.0, score=True)
        H.assert_records_equal(rec, orc, 6)
        dev.run_async(6, skip, 0.0, score=True)
        H.assert_records_equal(dev.wait(), orc, 6)
    assert rec.n > 10000


@pytest.mark.parametrize('motif,broken', [('GATC', False), ('GATC', True), ('A', False), ('A', True)])
def test_first_second_and_later_passes_over_one_table(dev, motif, broken):
    """What a pass does depends on what the passes before it left: the FIRST pass over a table classifies the reads on their
    first rows and validates every row while it scans (k1_scan, SCAN_VALIDATE), the second classifies on the complete
    validation flags and streams the positions, the third finds unit summaries (sparse motifs).  Every one of them, through
    the synchronous and the pipelined interface, must hand out the oracle's records -- also when a read turns irregular far
    from its first rows (the first pass notices while it scans and is repeated), and again after mc_ctx_select_table(as_new)."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    from tests.test_gpu_stream import _irregular
    codes = synth.genome(length=500000, seed=41)
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(300000, seed=77, codes=codes)
    if broken:
        table = _irregular(table, 5)
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    for first_sync in (True, False):
        slot = dev.upload_table_async(table, qual)
        dev.wait_upload(slot)
        for round_ in range(2):
            for i in range(4):
                if (i % 2 == 0) == first_sync:
                    rec = dev.extract(6, 0, 0.0)
                else:
                    dev.run_async(6, 0, 0.0)
                    rec = dev.wait()
                H.assert_records_equal(rec, orc, 6)
            dev.run_async(6, 0, 0.0)                   # two in flight over the same table
            dev.run_async(6, 0, 0.0)
            H.assert_records_equal(dev.wait(), orc, 6)
            H.assert_records_equal(dev.wait(), orc, 6)
            dev.select_table(slot, as_new=True)        # ... and everything again, as if the table had just arrived
    dev.sync()


def with_stalls(table, n_stalls, reps, seed):
    """Rows repeated `reps` times in place (a pore that stalls: dozens of events at one position), event indices renumbered
    monotone in every read's own direction: windows longer than the emit looks back (64 rows) and, from 129 repeats on, slots of more
    than 128 events (NumPy's pairwise recursion proper)."""
    from mcaller_amd import _lib
    rng = np.random.default_rng(seed)
    n = table.n_rows
    count = np.ones(n, dtype=np.int64)
    ok = np.flatnonzero((table.flags & (_lib.F_MODEL_N | _lib.F_SEG_START)) == 0)
    picks = rng.choice(ok, size=n_stalls, replace=False)
    count[picks] = rng.choice(reps, size=n_stalls)
    sb = table.seg_row_begin
    new_begin = np.concatenate([[0], np.cumsum(count)])[sb]
    idx = np.repeat(table.event_idx, count).astype(np.int64)
    for s in range(table.n_seg):
        a, b = int(new_begin[s]), int(new_begin[s + 1])
        if b - a < 2:
            continue
        up = table.event_idx[sb[s] + 1] > table.event_idx[sb[s]]
        idx[a:b] = idx[a] + (np.arange(b - a) if up else -np.arange(b - a))
        if idx[a:b].min() < 0:
            idx[a:b] -= idx[a:b].min()
    fl = np.repeat(table.flags, count)
    first = np.concatenate([[True], np.diff(np.repeat(np.arange(n), count)) != 0])
    fl[~first] &= np.uint8(0xFF ^ (_lib.F_SEG_START | _lib.F_NAME_START))
    return _lib.Table(np.repeat(table.pos, count), np.repeat(table.event_e4, count), np.repeat(table.model_e4, count), idx.astype(np.int32), fl,
                      new_begin.astype(np.int64), table.seg_read, table.seg_contig, table.n_reads, read_names=table.read_names)


@pytest.mark.parametrize('motif,reps,skip,expect_rerun', [('GATC', (40, 70, 100, 110), 0, False), ('GATC', (70, 90), 2, False),
                                                          ('GATC', (70, 130, 300), 0, True), ('A', (40, 60), 0, False),
                                                          ('A', (70, 140), 1, True)])
def test_windows_left_to_the_row_by_row_walk_in_pipelined_passes(motif, reps, skip, expect_rerun):
    """Pipelined, scored passes over reads with stalls: windows of more than 64 rows are left to the row-by-row walk, which the side
    stream's one kernel does for its own records -- their rows in the packed block were counted by the emit from the walk's own rule
    (calls: a row; too many skips: none) -- and a slot of more than 128 events marks the pass, which is repeated by the synchronous
    path.  Records (slot means bit for bit, probabilities) as the oracle's; sparse motif (k1_emit) and dense (k1_fused)."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.device import Device
    dev = Device(0)             # (a context of its own: what an earlier test's pass ran out of -- room per piece, doubled since -- decides which kernels run)
    try:
        _windows_left_to_the_walk(dev, motif, reps, skip, expect_rerun)
    finally:
        dev.close()


def _windows_left_to_the_walk(dev, motif, reps, skip, expect_rerun):
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=200000, seed=23)
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(250000, seed=37, codes=codes, read_len=(1500, 6000))
    table = with_stalls(table, 3000 if motif == 'GATC' else 400, reps, seed=len(reps) + skip)     # (a sparse motif: a stall in seventy lies in a window)
    arrays = ref.device_arrays()
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, skip, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    dev.set_reference(arrays)
    dev.set_mlp(weights, soc)
    slot = dev.upload_table_async(table, qual)
    for as_new in (False, True):
        if as_new:
            dev.select_table(slot, as_new=True)
        dev.run_async(6, skip, 0.0, score=True)
        dev.run_async(6, skip, 0.0, score=True)
        for _ in range(2):
            rec = dev.wait()
            assert bool(dev.last_pass_info()[1]) == expect_rerun
            H.assert_records_equal(rec, orc, 6, prob_tol=1e-6)
    dev.run_async(6, skip, 0.0, score=False)                     # features only: the walk and the packing without a classifier
    rec = dev.wait()
    orc.prob[:orc.n] = np.nan
    rec.prob[:rec.n] = np.nan
    H.assert_records_equal(rec, orc, 6)
