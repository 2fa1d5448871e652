"""Streaming distinct tables through the table slots of a context (mc_ctx_upload_table_async): the reference's batch
loop (extract_contexts.py:140-148) as shards in flight.  Every shard's records must equal the C oracle's.
Runs on a real MI355X only: `pytest -m gpu`."""
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
def setup():
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=600000, seed=23)
    ref = synth.SynthRef(codes, motif='GATC')
    _, weights, _, soc = ec.submodel_setup(H.load_modelset('r95'), 'A')
    return codes, ref, weights, soc


def _oracle(table, ref, qual, weights, soc, tail):
    orc = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0, tail_contig=tail)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    return orc


def _irregular(table, seed):
    """Break a few reads of a regular table: positions going backwards, equal event indices, a read name that comes back."""
    from mcaller_amd import _lib
    rng = np.random.default_rng(seed)
    pos, idx = table.pos.copy(), table.event_idx.copy()
    seg_read = table.seg_read.copy()
    for seg in rng.choice(table.n_seg, size=max(1, table.n_seg // 6), replace=False):
        r0, r1 = int(table.seg_row_begin[seg]), int(table.seg_row_begin[seg + 1])
        if r1 - r0 < 8:
            continue
        r = int(rng.integers(r0 + 2, r1 - 2))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            pos[r] = pos[r - 1] - int(rng.integers(1, 4))
        elif kind == 1:
            idx[r] = idx[r - 1]
        elif seg > 1:
            seg_read[seg] = seg_read[seg - 2]
    return _lib.Table(pos, None, None, idx, table.flags, table.seg_row_begin, seg_read, table.seg_contig, table.n_reads,
                      read_names=table.read_names, evmu=table.evmu)


def test_stream_distinct_shards_two_in_flight(dev, setup):
    """Ten distinct shards (own rows, own read ids, own read qualities) streamed through the slots from pinned memory, two
    passes in flight while the next shard uploads; one shard holds irregular reads (re-run on ITS table when handed out)."""
    from mcaller_amd import synth
    codes, ref, weights, soc = setup
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    shards = []
    for i in range(10):
        n_rows = [60000, 250000, 3072, 120001, 400000, 77, 180000, 300000, 6144, 90000][i]
        t, q = synth.make_table(n_rows, seed=500 + i, codes=codes, read_len=(600, 6000) if i % 3 else (40, 300))
        if i == 4:
            t = _irregular(t, 9)
        tail = -1 if i == 9 else 0                      # every shard but the last is followed by rows of contig 0
        shards.append((t.pinned(), q, tail))
    dev.reserve_tables(max(t.n_rows for t, _, _ in shards), max(t.n_seg for t, _, _ in shards),
                       max(t.n_reads for t, _, _ in shards))
    in_flight, n_checked = [], 0

    def hand_out():
        nonlocal n_checked
        t, q, tail = in_flight.pop(0)
        rec = dev.wait()
        H.assert_records_equal(rec, _oracle(t, ref, q, weights, soc, tail), 6)
        n_checked += 1

    for t, q, tail in shards:
        dev.upload_table_async(t, q)
        dev.run_async(6, 0, 0.0, tail_contig=tail, score=True)
        in_flight.append((t, q, tail))
        if len(in_flight) > 2:
            hand_out()
    while in_flight:
        hand_out()
    assert n_checked == 10


def test_stream_three_rounds_reuse_the_slots(dev, setup):
    """The slots are reused round after round (no allocation: the sizes were reserved); smaller and bigger tables alternate,
    so stale rows of an earlier table sit behind the current one."""
    from mcaller_amd import synth
    codes, ref, weights, soc = setup
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    dev.reserve_tables(300000, 4000, 4000)
    prev = None
    for i in range(13):
        n_rows = 300000 if i % 2 == 0 else 20000 + 1000 * i
        t, q = synth.make_table(n_rows, seed=900 + i, codes=codes, read_len=(300, 3000))
        t = t.pinned()
        dev.upload_table_async(t, q)
        dev.run_async(6, 0, 0.0, tail_contig=0, score=True)
        if prev is not None:
            H.assert_records_equal(dev.wait(), _oracle(prev[0], ref, prev[1], weights, soc, 0), 6)
        prev = (t, q)
    H.assert_records_equal(dev.wait(), _oracle(prev[0], ref, prev[1], weights, soc, 0), 6)


def test_upload_while_passes_are_in_flight(dev, setup):
    """mc_ctx_upload_table (the one-table interface) while two passes over the previous table are in flight: they keep their
    table (another slot takes the new one) and hand out that table's records; with every slot scanned the asynchronous
    upload refuses with a clean error."""
    from mcaller_amd import synth, _lib
    codes, ref, weights, soc = setup
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    t1, q1 = synth.make_table(200000, seed=71, codes=codes, read_len=(500, 4000))
    t2, q2 = synth.make_table(150000, seed=72, codes=codes, read_len=(500, 4000))
    dev.upload_table(t1)
    dev.set_read_quality(q1)
    dev.run_async(6, 0, 0.0, score=True)
    dev.run_async(6, 0, 0.0, score=True)
    dev.upload_table(t2)                                  # pageable source, synchronous
    dev.set_read_quality(q2)
    dev.run_async(6, 0, 0.0, score=True)
    o1, o2 = _oracle(t1, ref, q1, weights, soc, -1), _oracle(t2, ref, q2, weights, soc, -1)
    H.assert_records_equal(dev.wait(), o1, 6)
    H.assert_records_equal(dev.wait(), o1, 6)
    H.assert_records_equal(dev.wait(), o2, 6)
    # every slot busy
    tables = []
    for i in range(4):
        t, q = synth.make_table(5000 + i, seed=80 + i, codes=codes, read_len=(300, 900))
        tables.append((t.pinned(), q))
    n_ok = 0
    with pytest.raises(_lib.McError, match='table slots|passes are in flight'):      # (twelve slots, six passes in flight)
        for t, q in tables + tables:
            dev.upload_table_async(t, q)
            dev.run_async(6, 0, 0.0, score=True)
            n_ok += 1
    assert 2 <= n_ok <= 6
    for i in range(n_ok):
        t, q = (tables + tables)[i]
        H.assert_records_equal(dev.wait(), _oracle(t, ref, q, weights, soc, -1), 6)


def test_validation_flags_many_small_and_broken_reads(dev, setup):
    """k_validate: tiles that hold more name blocks than its LDS table (reads of 20-60 rows), every kind of irregular block
    among them; the flags decide which path a block takes, so a wrong flag shows as a record mismatch."""
    from mcaller_amd import synth
    codes, _, weights, soc = setup
    ref = synth.SynthRef(codes, motif='A')
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    for seed, rl in ((1, (12, 40)), (2, (30, 120)), (3, (2000, 9000))):
        t, q = synth.make_table(40000, seed=seed, codes=codes, read_len=rl)
        t = _irregular(t, seed)
        dev.upload_table(t)
        dev.set_read_quality(q)
        rec = dev.extract(6, 1, 0.0, score=True)
        orc = H.oracle_records(t, ref.device_arrays(), q, 6, 1, 0.0)
        H.oracle_score(orc, t, q, weights, soc, 6)
        H.assert_records_equal(rec, orc, 6)


def test_parser_tables_come_from_the_pinned_pool(tmp_path):
    from mcaller_amd import synth, _lib
    codes = synth.genome(length=200000, seed=4)
    table, _ = synth.make_table(30000, seed=6, codes=codes, read_len=(400, 2000))
    tsv = str(tmp_path / 'p.eventalign.tsv')
    synth.write_tsv(table, codes, tsv)
    L = _lib.lib()
    L.mc_host_pool_config(1, -1)
    try:
        t = _lib.parse_eventalign(tsv, 0, 1 << 40, ['ecoli_syn'], exact_range=True)
        assert t.n_rows == table.n_rows and (t.evmu == table.evmu).all() and (t.pos == table.pos).all()
        assert L.mc_host_is_pinned(t.pos.ctypes.data) == 1 and L.mc_host_is_pinned(t.evmu.ctypes.data) == 1
        first = t.pos.ctypes.data
        del t
        t2 = _lib.parse_eventalign(tsv, 0, 1 << 40, ['ecoli_syn'], exact_range=True)
        blocks = {t2.pos.ctypes.data, t2.evmu.ctypes.data, t2.event_idx.ctypes.data, t2.flags.ctypes.data}
        assert first in blocks                           # the blocks of the freed table were handed out again
    finally:
        L.mc_host_pool_config(0, -1)


# ---------------------------------------------------------------------------------------------------------------------
# extract_features over a file, streamed in shards (mcaller_amd.extract_contexts.stream_features)
# ---------------------------------------------------------------------------------------------------------------------
def _run_extract(paths, args, monkeypatch, shards):
    """extract_features (the drop-in) on a case's files -> (outcome, text written, stdout lines)."""
    import contextlib
    import io
    import os
    from mcaller_amd.extract_contexts import extract_features
    from mcaller_amd.read_qual import extract_read_quality
    if shards:
        monkeypatch.setenv('MCALLER_STREAM_SHARDS', str(shards))
        monkeypatch.delenv('MCALLER_NO_STREAM', raising=False)
    else:
        monkeypatch.setenv('MCALLER_NO_STREAM', '1')
    out = '.'.join(paths['tsv'].split('.')[:-1]) + '.diffs.%d.tmp0' % args['k']
    if os.path.exists(out):
        os.remove(out)
    buf = io.StringIO()
    outcome = 'ok'
    with contextlib.redirect_stdout(buf):
        try:
            r2q = extract_read_quality(paths['fastq'])
            extract_features(paths['tsv'], paths['fasta'], r2q, args['k'], args['skip_thresh'], args['qual_thresh'],
                             os.path.join(H.MODELS, H.MODEL_STEMS[args['model']] + '.npz'), 'NN', 0,
                             endline=os.path.getsize(paths['tsv']), train=False, pos_label=None, base=args['base'],
                             motif=args['motif'], positions_list=paths['positions'])
        except SystemExit:
            outcome = 'exit'
        except Exception as e:                             # noqa
            outcome = 'crash:' + type(e).__name__
    text = open(out).read() if os.path.exists(out) else ''
    return outcome, text, [l for l in buf.getvalue().split('\n') if l.strip()]


def test_streamed_micro_cases_reproduce_the_reference(tmp_path, monkeypatch):
    """Every committed predict-mode micro-case (all quirk flavours, exit paths, several contigs) through extract_features
    with the file cut into shards: the bytes written and the printed lines are the reference's (captured in the build
    container).  Cases the shards cannot reproduce take the one-table path from scratch -- nothing may be printed twice."""
    bad, n_ok = [], 0
    for case in H.micro_cases():
        if case['args']['train']:
            continue
        d = tmp_path / ('s%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        outcome, text, stdout = _run_extract(paths, case['args'], monkeypatch, shards=3)
        exp = case['expected']
        if (exp['outcome'] != 'ok') != (outcome != 'ok'):
            bad.append((case['seed'], case['flavour'], 'outcome', exp['outcome'], outcome))
        elif exp['outcome'] != 'ok':
            if not outcome.startswith('crash') and (exp['text'] or '') != text:
                bad.append((case['seed'], case['flavour'], 'partial text'))
        elif (exp['text'] or '') != text:
            bad.append((case['seed'], case['flavour'], 'text'))
        elif exp['stdout'] != stdout:
            bad.append((case['seed'], case['flavour'], 'stdout', exp['stdout'], stdout))
        else:
            n_ok += 1
    assert not bad, '%d differ: %s' % (len(bad), bad[:8])
    assert n_ok > 150


@pytest.mark.parametrize('n_rows,motif,skip,qthresh,shards', [
    (300000, 'GATC', 0, 0.0, 7),
    (200000, 'A', 1, 0.0, 5),
    (250000, 'GATC', 0, 9.0, 16),          # reads filtered by quality: shards whose first reads are skipped whole
    (220000, 'GATC', 0, 11.5, 30),         # ... nearly all reads: more shards dropped than there are table slots
])
def test_streamed_file_equals_the_one_table_path(tmp_path, monkeypatch, n_rows, motif, skip, qthresh, shards):
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=500000, seed=31)
    table, qual = synth.make_table(n_rows, seed=n_rows % 97, codes=codes, read_len=(700, 5000))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    paths['positions'] = None
    args = dict(k=6, skip_thresh=skip, qual_thresh=qthresh, base='A', motif=motif, model='r95')
    streamed_calls = []
    real = ec.stream_features

    def spy(*a, **kw):
        out = real(*a, **kw)
        streamed_calls.append(out.n_bytes)
        return out
    monkeypatch.setattr(ec, 'stream_features', spy)
    got = _run_extract(paths, args, monkeypatch, shards=shards)
    assert streamed_calls and streamed_calls[0] > 0            # the shards really went through stream_features
    want = _run_extract(paths, args, monkeypatch, shards=0)
    assert got[0] == want[0] == 'ok'
    assert got[2] == want[2]                                   # the counter lines
    assert got[1] == want[1] and (len(got[1]) > 1000 or qthresh > 11)      # the rows, byte for byte


def test_streamed_rows_equal_the_python_oracle(tmp_path, monkeypatch):
    """... and against the literal Python restatement of the reference (pinned to the reference itself) on a small file."""
    from mcaller_amd import synth
    from oracle import py_oracle
    import numpy as np
    codes = synth.genome(length=200000, seed=8)
    table, qual = synth.make_table(50000, seed=12, codes=codes, read_len=(500, 2500))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    paths['positions'] = None
    args = dict(k=6, skip_thresh=0, qual_thresh=0.0, base='A', motif='GATC', model='r95')
    outcome, text, stdout = _run_extract(paths, args, monkeypatch, shards=4)
    assert outcome == 'ok'
    z = np.load(H.MODELS + '/r95_twobase_model_NN_6_m6A.npz')
    models = {k: (z[k + '.W1'], z[k + '.b1'], z[k + '.W2'], z[k + '.b2']) for k in ('MG', 'MH')}
    models['__twobase__'] = True
    import os
    res = py_oracle.extract_features_oracle(paths['tsv'], paths['fasta'], py_oracle.read_fastq_quality(paths['fastq']), 6, 0, 0.0,
                                            models, 0, os.path.getsize(paths['tsv']), base='A', motif='GATC')
    want = ''.join('\t'.join(r) + '\n' for r in res['rows'])
    assert text == want and len(res['rows']) > 30
    assert stdout == [l for l in res['stdout'] if l.strip()] or stdout[-6:] == [l for l in res['stdout'] if l.strip()][-6:]


@pytest.mark.parametrize('host_parser', [False, True])
def test_streamed_file_with_a_shard_the_device_parser_declines(tmp_path, monkeypatch, host_parser):
    """One row in the middle writes its event mean with an exponent (float() takes it, the device parser's plain-decimal form does
    not): that shard goes through the host parser, the others stay on the device; and the whole file through the host parser
    (MCALLER_HOST_PARSER).  Same bytes as the one-table path either way."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=400000, seed=5)
    table, qual = synth.make_table(240000, seed=3, codes=codes, read_len=(700, 4000))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    paths['positions'] = None
    lines = open(paths['tsv']).read().split('\n')
    t = lines[130000].split('\t')
    t[6] = '%.6e' % float(t[6])                                 # e.g. 8.123000e+01: the same value to float()
    assert float(t[6]) == float(lines[130000].split('\t')[6])
    lines[130000] = '\t'.join(t)
    open(paths['tsv'], 'w').write('\n'.join(lines))
    args = dict(k=6, skip_thresh=0, qual_thresh=0.0, base='A', motif='GATC', model='r95')
    if host_parser:
        monkeypatch.setenv('MCALLER_HOST_PARSER', '1')
    clocks = []
    real = ec.stream_features

    def spy(*a, **kw):
        out = real(*a, **kw)
        clocks.append(dict(ec.stream_features.last_clock))
        return out
    monkeypatch.setattr(ec, 'stream_features', spy)
    got = _run_extract(paths, args, monkeypatch, shards=6)
    assert clocks and clocks[0]['shards'] == 6
    assert clocks[0]['device_parsed'] == (0 if host_parser else 5)
    monkeypatch.delenv('MCALLER_HOST_PARSER', raising=False)
    want = _run_extract(paths, args, monkeypatch, shards=0)
    assert got[0] == want[0] == 'ok' and got[2] == want[2]
    assert got[1] == want[1] and len(got[1]) > 1000


def _rename(table, seg_to, seg_from):
    """The read of segment seg_to gets the name (and the quality) of the read of segment seg_from: a read name that comes back."""
    from mcaller_amd import _lib
    seg_read = table.seg_read.copy()
    seg_read[seg_to] = seg_read[seg_from]
    return _lib.Table(table.pos, None, None, table.event_idx, table.flags, table.seg_row_begin, seg_read, table.seg_contig, table.n_reads,
                      read_names=table.read_names, evmu=table.evmu)


def test_a_read_name_that_comes_back_far_away_does_not_stop_the_stream(tmp_path, monkeypatch):
    """A read name that occurs twice in a file, dozens of reads apart: `last_read` (extract_contexts.py:161-174) never equals it when
    its second read begins, so the cuts between the shards stand (extract_contexts.cut_names) -- the file is streamed, rows and
    counter lines equal the one-table path's.  (Up to round 4 any repeated name sent the whole file to the one-table path.)"""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=500000, seed=33)
    table, qual = synth.make_table(300000, seed=17, codes=codes, read_len=(700, 5000))
    assert table.n_seg > 40
    table = _rename(_rename(table, 30, 4), table.n_seg - 3, 11)
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    paths['positions'] = None
    args = dict(k=6, skip_thresh=0, qual_thresh=0.0, base='A', motif='GATC', model='r95')
    streamed = []
    real = ec.stream_features

    def spy(*a, **kw):
        out = real(*a, **kw)
        streamed.append(out.n_bytes)
        return out
    monkeypatch.setattr(ec, 'stream_features', spy)
    got = _run_extract(paths, args, monkeypatch, shards=9)
    assert streamed and streamed[0] > 0                         # streamed to the end: no fall-back
    want = _run_extract(paths, args, monkeypatch, shards=0)
    assert got == want and got[0] == 'ok' and len(got[1]) > 1000


def test_a_read_name_on_both_sides_of_a_cut_takes_the_one_table_path(tmp_path, monkeypatch):
    """... and the case that does matter: a read, a read the quality filter drops, and the first read's name again -- `last_read`
    still holds the name when it comes back.  With a cut between them the stream gives up (the rows appended so far are taken
    back) and the one-table path, whose literal machine handles the repeat, writes the file: the same bytes either way."""
    from mcaller_amd import synth
    from mcaller_amd import extract_contexts as ec
    codes = synth.genome(length=400000, seed=34)
    table, qual = synth.make_table(200000, seed=18, codes=codes, read_len=(900, 3000))
    qual = np.array(qual, dtype=np.float64)
    assert table.n_seg > 40
    for s in range(5, table.n_seg - 3, 6):                       # every sixth read: X, (filtered), X again
        table = _rename(table, s + 2, s)
        qual[table.seg_read[s + 1]] = 1.0
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    paths['positions'] = None
    args = dict(k=6, skip_thresh=0, qual_thresh=5.0, base='A', motif='GATC', model='r95')
    streamed = []
    real = ec.stream_features

    def spy(*a, **kw):
        out = real(*a, **kw)
        streamed.append(out.n_bytes)
        return out
    monkeypatch.setattr(ec, 'stream_features', spy)
    got = _run_extract(paths, args, monkeypatch, shards=table.n_seg)          # a cut at (nearly) every read start
    assert not streamed                                          # the stream gave up
    want = _run_extract(paths, args, monkeypatch, shards=0)
    assert got == want and got[0] == 'ok' and len(got[1]) > 500
