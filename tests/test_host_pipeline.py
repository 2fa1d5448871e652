"""Host pipeline (native parser -> marking -> [records] -> formatter) against the golden fixtures, with the C
oracle standing in for the GPU as the producer of flush records.  This pins (a) the C oracle to the reference's
outputs and (b) every host-side piece the HIP path shares: parser, bitmasks, context slicing, number formatting,
counters, exit paths.  No GPU needed."""
import contextlib
import ctypes as C
import io
import os

import numpy as np
import pytest

from tests import helpers as H


def run_with_oracle(paths, args, out_dir):
    """What mcaller_amd.extract_contexts.extract_features does, with oracle records instead of device records."""
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    read2qual = extract_read_quality(paths['fastq'])
    k, train, base = args['k'], args['train'], args['base']
    modelset = None if train else H.load_modelset(args['model'])
    buf = io.StringIO()
    outcome, fin = 'ok', None
    with contextlib.redirect_stdout(buf):
        try:
            P = ec.prepare(paths['tsv'], paths['fasta'], read2qual, 0, os.path.getsize(paths['tsv']), base,
                           args['motif'], paths['positions'])
            rec = H.oracle_records(P.table, P.ref.device_arrays(), P.qual, k, args['skip_thresh'], args['qual_thresh'])
            if not train:
                _, weights, _, soc = ec.submodel_setup(modelset, base)
                H.oracle_score(rec, P.table, P.qual, weights, soc, k)

            class _NoDev(object):
                def mlp_forward(self, X, sub):
                    raise AssertionError('edge records need the device classifier')
            fin = ec.Finisher(P, k, base, train, modelset=modelset,
                              pos_label=H.pos2label(paths['positions']) if (train and paths['positions']) else None,
                              device=_NoDev())
            stop = fin.run(rec)
            if stop is None and P.fatal is not None:
                stop = P.fatal
            if stop is not None:
                outcome = 'exit' if isinstance(stop, SystemExit) else 'crash:' + type(stop).__name__
                rows = fin.rows[:(fin.num_observations // 5000) * 5000]
            else:
                rows = fin.rows
                for line in fin.counters():
                    print(line)
        except Exception as e:                         # noqa
            outcome = 'crash:' + type(e).__name__
            rows = []
    text = ''.join('\t'.join(r) + '\n' for r in rows)
    return outcome, text, [l for l in buf.getvalue().split('\n') if l.strip()], fin


@pytest.fixture(scope='module')
def td(tmp_path_factory):
    return H.testdata_paths(str(tmp_path_factory.mktemp('testdata')))


@pytest.mark.parametrize('tag,kw,model,skip', [
    ('config1_positions_m6A', dict(positions='test_positions_m6A.txt'), 'r95', 0),
    ('motif_GATC', dict(motif='GATC'), 'r95', 0),
    ('motif_A', dict(motif='A'), 'r95', 0),
    ('positions_all', dict(positions='test_positions.txt'), 'r95', 0),
    ('motif_GATC_s1', dict(motif='GATC'), 'r95', 1),
    ('motif_A_r94', dict(motif='A'), 'r94', 0),
])
def test_testdata_against_reference_outputs(td, tag, kw, model, skip, tmp_path):
    paths = dict(tsv=td['tsv'], fasta=td['fasta'], fastq=td['fastq'],
                 positions=td[kw['positions']] if 'positions' in kw else None)
    args = dict(k=6, skip_thresh=skip, qual_thresh=0, base='A', motif=kw.get('motif'), train=False, model=model)
    outcome, text, stdout, _ = run_with_oracle(paths, args, str(tmp_path))
    assert outcome == 'ok'
    ref = open(os.path.join(H.GOLDEN, 'ref_outputs', tag + '.diffs.6')).read()
    assert text == ref
    ref_stdout = open(os.path.join(H.GOLDEN, 'ref_outputs', tag + '.stdout')).read().split('\n')
    for line in stdout:
        assert line in ref_stdout


def test_reference_own_golden_columns(td, tmp_path):
    """The reference's own golden diffs.6 (README.md:132): columns 1-6 byte-for-byte (7-8 are stale there)."""
    paths = dict(tsv=td['tsv'], fasta=td['fasta'], fastq=td['fastq'], positions=td['test_positions_m6A.txt'])
    args = dict(k=6, skip_thresh=0, qual_thresh=0, base='A', motif=None, train=False, model='r95')
    _, text, _, _ = run_with_oracle(paths, args, str(tmp_path))
    gold = open(os.path.join(H.GOLDEN, 'testdata', 'masonread1.eventalign.diffs.6')).read()
    mine = ['\t'.join(l.split('\t')[:6]) for l in text.strip().split('\n')]
    theirs = ['\t'.join(l.split('\t')[:6]) for l in gold.strip().split('\n')]
    assert mine == theirs


def test_train_dicts(td, tmp_path):
    import json
    paths = dict(tsv=td['tsv'], fasta=td['fasta'], fastq=td['fastq'], positions=td['test_positions.txt'])
    args = dict(k=6, skip_thresh=0, qual_thresh=0, base='A', motif=None, train=True, model='r95')
    outcome, text, stdout, fin = run_with_oracle(paths, args, str(tmp_path))
    assert outcome == 'ok'
    assert text == open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.diffs.6.train')).read()
    gold = json.load(open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.dicts.json')))
    assert H.plain_signals(fin.signals) == gold['signals']
    assert fin.contexts == gold['contexts']
    ref_stdout = open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.stdout')).read().split('\n')
    for line in stdout:
        assert line in ref_stdout


def test_micro_cases(tmp_path):
    bad = []
    cases = H.micro_cases()
    for case in cases:
        d = tmp_path / ('c%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        outcome, text, stdout, fin = run_with_oracle(paths, case['args'], str(d))
        exp = case['expected']
        exp_bad, got_bad = exp['outcome'] != 'ok', outcome != 'ok'
        if exp_bad != got_bad:
            bad.append((case['seed'], case['flavour'], 'outcome', exp['outcome'], outcome))
            continue
        if exp_bad:
            if not outcome.startswith('crash') and (exp['text'] or '') != text:
                bad.append((case['seed'], case['flavour'], 'partial text'))
            continue
        if (exp['text'] or '') != text:
            bad.append((case['seed'], case['flavour'], 'text'))
        elif exp['stdout'] != stdout:
            bad.append((case['seed'], case['flavour'], 'stdout', exp['stdout'], stdout))
        elif exp['train'] is not None:
            if H.plain_signals(fin.signals) != exp['train']['signals'] or fin.contexts != exp['train']['contexts']:
                bad.append((case['seed'], case['flavour'], 'train dicts'))
    assert not bad, '%d of %d micro-cases differ: %s' % (len(bad), len(cases), bad[:10])


def test_native_formatter_equals_per_record_path(td, tmp_path):
    """mc_format_diffs (multi-threaded) writes the same bytes as the literal per-record transcription, on the dense
    `-m A` run (2713 rows, empty-slot zeros with -s 2) and with records handed back to the host in between."""
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    read2qual = extract_read_quality(td['fastq'])
    modelset = H.load_modelset('r95')
    for skip in (0, 2):
        P = ec.prepare(td['tsv'], td['fasta'], read2qual, 0, os.path.getsize(td['tsv']), 'A', 'A', None)
        rec = H.oracle_records(P.table, P.ref.device_arrays(), P.qual, 6, skip, 0)
        _, weights, _, soc = ec.submodel_setup(modelset, 'A')
        H.oracle_score(rec, P.table, P.qual, weights, soc, 6)
        native = ec.Finisher(P, 6, 'A', False, modelset=modelset)
        with contextlib.redirect_stdout(io.StringIO()):
            assert native.run(rec) is None
        literal = ec.Finisher(P, 6, 'A', False, modelset=modelset)
        literal._bind(rec)
        for j in range(rec.n):
            assert literal._one(j) is None
        assert native.text() == literal.text() and native.num_observations == literal.num_observations > 2000
        assert native.counters() == literal.counters()
        # hand every 97th scored record back to the host (NaN probability -> the host asks the device classifier)
        keep = rec.prob[:rec.n].copy()
        scored = np.nonzero((rec.info[:rec.n] & ec._I.I_TOO_MANY) == 0)[0][::97]
        handed_back = iter(scored)

        class _Dev(object):
            def mlp_forward(self, X, sub):           # records come back in order: answer with the value taken away
                return np.array([keep[next(handed_back)]])
        rec.prob[scored] = np.nan
        mixed = ec.Finisher(P, 6, 'A', False, modelset=modelset, device=_Dev())
        assert mixed.run(rec) is None
        rec.prob[:rec.n] = keep
        assert mixed.text() == literal.text() and mixed.counters() == literal.counters()


def _compacted(rec, k, as_sent=False):
    """The records as a pipelined pass hands them out (mc_wait_records): slot means / probabilities of the calls only.
    as_sent: without the call_row column and with 32-bit closing rows, the way they cross PCIe."""
    from mcaller_amd import _lib
    n = rec.n
    kept = (rec.info[:n] & _lib.I_TOO_MANY) == 0
    c = _lib.Records.__new__(_lib.Records)
    c.k, c.capacity, c.n = k, n, n
    for name in ('site_pos', 'site_seg', 'close_row', 'info'):
        setattr(c, name, getattr(rec, name)[:n].copy())
    dense = np.ascontiguousarray(rec.feats[:n * k].reshape(n, k)[kept]).reshape(-1)
    c._n_calls = int(kept.sum())
    if as_sent:
        c._close_row32, c._close_row = rec.close_row[:n].astype(np.int32), None
        c._compacted = True
        # the slot means packed like k_pack packs them: 32-bit d where the mean is fl(d / 1e4), both halves where it is not
        d = np.rint(dense * 1e4)
        ok = np.abs(d) < 2147483648.0
        back = np.where(ok, d, 0.0).astype(np.int32).astype(np.float64) / 1e4
        narrow = ok & (back.view(np.uint64) == dense.view(np.uint64))
        bits = dense.view(np.uint64)
        lo = np.where(narrow, np.where(ok, d, 0.0).astype(np.int32), (bits & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
        hi = (bits[~narrow] >> 32).astype(np.uint32)
        wide = (~narrow).reshape(-1, k)
        mask = (wide * (1 << np.arange(k))).sum(axis=1).astype(np.uint8)
        c._packed = (np.ascontiguousarray(lo, dtype=np.int32), np.ascontiguousarray(hi), np.ascontiguousarray(mask))
        assert 0 < len(hi) < len(dense)                      # (some of each kind)
    else:
        c.call_row = np.where(kept, np.cumsum(kept) - 1, -1).astype(np.int32)
        c.feats = dense
    c.prob = np.ascontiguousarray(rec.prob[:n][kept])
    return c


def test_rows_from_a_compacted_view(td, tmp_path):
    """Finisher (native formatter and the per-record path) on a compacted view == on one row per record."""
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    read2qual = extract_read_quality(td['fastq'])
    modelset = H.load_modelset('r95')
    P = ec.prepare(td['tsv'], td['fasta'], read2qual, 0, os.path.getsize(td['tsv']), 'A', 'A', None)
    rec = H.oracle_records(P.table, P.ref.device_arrays(), P.qual, 6, 1, 0)
    _, weights, _, soc = ec.submodel_setup(modelset, 'A')
    H.oracle_score(rec, P.table, P.qual, weights, soc, 6)
    comp = _compacted(rec, 6)
    sent = _compacted(rec, 6, as_sent=True)
    assert 0 < comp.n_calls < comp.n and sent.n_calls == comp.n_calls
    outs = []
    for r in (rec, comp, sent):
        native = ec.Finisher(P, 6, 'A', False, modelset=modelset)
        with contextlib.redirect_stdout(io.StringIO()):
            assert native.run(r) is None
        literal = ec.Finisher(P, 6, 'A', False, modelset=modelset)
        literal._bind(r)
        for j in range(r.n):
            assert literal._one(j) is None
        assert native.text() == literal.text()
        outs.append((native.text(), native.counters()))
    assert outs[0] == outs[1] == outs[2] and len(outs[0][0]) > 10000
    for c in (comp, sent):
        back = c.by_record()
        assert np.array_equal(back.feats[:rec.n * 6], rec.feats[:rec.n * 6])
        assert np.array_equal(back.prob[:rec.n], rec.prob[:rec.n], equal_nan=True)
        assert np.array_equal(back.close_row[:rec.n], rec.close_row[:rec.n])
    # the library's own expansion of the columns that are not sent (all cores)
    from mcaller_amd import _lib
    rows, close = np.empty(rec.n, dtype=np.int32), np.empty(rec.n, dtype=np.int64)
    dense = np.empty(sent.n_calls * 6, dtype=np.float64)
    fresh = _compacted(rec, 6, as_sent=True)                  # (nothing unpacked yet)
    v = fresh.view()
    assert not v.feats and v.feats_lo32 and v.n_wide > 0
    _lib.check(_lib.lib().mc_calls_expand(C.byref(v), rec.n, 6, rows.ctypes.data, close.ctypes.data, dense.ctypes.data))
    assert np.array_equal(rows, comp.call_row[:rec.n]) and np.array_equal(close, rec.close_row[:rec.n])
    assert np.array_equal(dense.view(np.uint64), comp.feats.view(np.uint64))       # bit for bit
    assert np.array_equal(fresh.feats.view(np.uint64), comp.feats.view(np.uint64))


def test_native_motif_marking_equals_str_replace(tmp_path):
    """MarkedReference.mark on a long contig goes through mc_mark_motifs (upper-casing and the two str.replace calls of
    extract_contexts.py:33-41 without the interpreter lock): the same strings and the same device arrays as the Python
    statement, for motifs that overlap themselves, one-base motifs, motifs without the base, lower-case stretches."""
    from mcaller_amd import refmark
    rng = np.random.default_rng(5)
    seq = ''.join(rng.choice(list('ACGT'), 150000))
    seq = seq[:500].lower() + seq[500:70000] + 'GATCGATCGATCGATC' + 'AAAAAAAAA' + seq[70000:100000].lower() + seq[100000:] + 'GATC'
    fa = str(tmp_path / 'r.fa')
    open(fa, 'w').write('>c1 some description\n' + '\n'.join(seq[i:i + 61] for i in range(0, len(seq), 61)) + '\n>short\nACGATCGA\n')
    for motif, base in (('GATC', 'A'), ('A', 'A'), ('C', 'C'), ('AAAA', 'A'), ('GATCGA', 'A'), ('TCGATC', 'C'), ('GATC', 'C'),
                        ('GG', 'A'), ('CCAGG', 'C')):
        native = refmark.MarkedReference(fa, base, motif, None)
        plain = refmark.MarkedReference(fa, base, motif, None)
        plain._mark_motif_native = lambda cid: None
        for cid in (0, 1):
            assert native.mark(cid) == plain.mark(cid), (motif, base, cid)
        assert 0 in native._upper_bytes and 1 not in native._upper_bytes      # (the short contig takes the Python statement)
        assert native.upper(0) == plain.upper(0) == seq.upper()
        a, b = native.device_arrays(), plain.device_arrays()
        assert all(np.array_equal(a[k], b[k]) for k in a), (motif, base)
        assert native.mark(0) == refmark.methylate_references(seq.upper(), base, motif=motif)


def test_count_records_equals_the_numpy_counters():
    """mc_count_records (what a streamed shard adds to the reference's counters, :184-185, :234-239, :247-248, in one native pass)
    against the numpy expressions it replaces: counts, the ascending test (repeated read names make it fail: the caller counts
    distinct pairs), the marks of the calls' positions with marks that are too short, empty input."""
    from mcaller_amd import _lib
    rng = np.random.default_rng(3)
    for n, repeat in ((0, False), (1, False), (5000, False), (5000, True)):
        rec = _lib.Records(max(n, 1), 6)
        rec.n = n
        n_seg = 40
        seg_read = (np.arange(n_seg, dtype=np.int32) if not repeat else rng.integers(0, 5, n_seg).astype(np.int32))
        rec.site_seg[:n] = np.sort(rng.integers(0, n_seg, n)).astype(np.int32)
        pos = np.zeros(n, dtype=np.int64)
        for s in range(n_seg):                                  # (ascending sites inside a segment, as records in file order have them)
            m = rec.site_seg[:n] == s
            pos[m] = np.cumsum(rng.integers(1, 9, int(m.sum())))
        rec.site_pos[:n] = pos.astype(np.int32)
        info = rng.integers(0, 64, n).astype(np.uint32) * (rng.random(n) < 0.3)
        info |= np.where(rng.random(n) < 0.3, _lib.I_TOO_MANY, 0).astype(np.uint32)
        info |= np.where(rng.random(n) < 0.2, _lib.I_MULTI, 0).astype(np.uint32)
        rec.info[:n] = info
        counts, ascending, lo, top = rec.count(n, seg_read=seg_read)
        too = (info & _lib.I_TOO_MANY) != 0
        key = (seg_read[rec.site_seg[:n]].astype(np.int64) << 32) | pos
        assert ascending == (n < 2 or bool((key[1:] > key[:-1]).all())) and (ascending or repeat)
        assert counts == (int(too.sum()), int((~too & ((info & _lib.I_EMPTY_MASK) != 0)).sum()), int(((info & _lib.I_MULTI) != 0).sum()))
        kept = pos[~too]
        assert (lo, top) == ((int(kept.min()), int(kept.max()) + 1) if len(kept) else (0, 0))
        marks = np.zeros(max(1, top // 2), dtype=bool)          # too short: the positions below its length only
        _, _, _, top2 = rec.count(n, pos_marks=marks)
        assert top2 == top and np.array_equal(np.flatnonzero(marks), np.unique(kept[kept < len(marks)]))
        marks = np.zeros(top + 3, dtype=bool)
        rec.count(n, pos_marks=marks)
        assert np.array_equal(np.flatnonzero(marks), np.unique(kept))
