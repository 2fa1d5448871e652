"""The native eventalign parser against a line.split() parse (what the reference does, extract_contexts.py:140-152)."""
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import py_oracle as po


def split_parse(path, startline, endline, contigs):
    rows, unknown = [], []
    for line in po.consumed_lines(path, startline, endline):
        t = line.split()[:12]
        if len(t) < 12:
            continue
        if t[0] not in contigs:
            unknown.append(t[0])
            continue
        d = int(round((float(t[6]) - float(t[10])) * 10000))
        rows.append((contigs.index(t[0]), int(t[1]), t[3], int(t[5]), d, t[2] == t[9], t[9] == 'NNNNNN'))
    return rows, unknown


def check(path, startline, endline, contigs, n_threads=0):
    from mcaller_amd import _lib
    t = _lib.parse_eventalign(path, startline, endline, contigs, n_threads)
    rows, unknown = split_parse(path, startline, endline, contigs)
    assert t.n_rows == len(rows)
    assert t.unknown == unknown
    seg_of_row = np.repeat(np.arange(t.n_seg), np.diff(t.seg_row_begin))
    for i, (c, p, name, idx, d, eq, isn) in enumerate(rows):
        s = seg_of_row[i]
        assert t.seg_contig[s] == c and t.pos[i] == p and t.event_idx[i] == idx
        assert t.read_names[t.seg_read[s]] == name
        assert int(t.event_e4[i]) - int(t.model_e4[i]) == d
        assert bool(t.flags[i] & _lib.F_KMER_EQ) == eq and bool(t.flags[i] & _lib.F_MODEL_N) == isn
    # segment / name-block flags
    for i in range(t.n_rows):
        new_name = i == 0 or rows[i][2] != rows[i - 1][2]
        new_seg = new_name or rows[i][0] != rows[i - 1][0]
        assert bool(t.flags[i] & _lib.F_NAME_START) == new_name
        assert bool(t.flags[i] & _lib.F_SEG_START) == new_seg
    return t


def test_micro_cases(tmp_path):
    for case in H.micro_cases()[::3]:
        d = tmp_path / ('c%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        contigs = [r[0] for r in po.read_fasta(paths['fasta'])]
        check(paths['tsv'], 0, os.path.getsize(paths['tsv']), contigs)


def test_byte_ranges_follow_the_reference_window(tmp_path):
    """seek(max(start-500,0)); readlines(8000000) batches while linepos <= endline-500 (:141-146)."""
    from oracle import casegen
    case = casegen.gen_case(424242, flavour='plain')
    p = str(tmp_path / 'big.eventalign.tsv')
    body = case['tsv'] * 40
    open(p, 'w').write(body)
    size = os.path.getsize(p)
    contigs = ['ctg0', 'ecoli0', 'ctg1', 'ecoli1']
    for start, end in [(0, size), (0, size // 3), (size // 3, 2 * size // 3), (1234, 5678), (0, 400), (size - 100, size)]:
        check(p, start, end, contigs)
    # several threads: pieces cut at line starts and stitched (names, segments and flags across the cuts)
    for nt in (2, 3, 7, 16, 61):
        check(p, 0, size, contigs, n_threads=nt)
        check(p, size // 5, 4 * size // 5, contigs, n_threads=nt)


def test_testdata_columns(tmp_path):
    td = H.testdata_paths(str(tmp_path))
    t = check(td['tsv'], 0, os.path.getsize(td['tsv']), ['ecoli'])
    assert t.n_rows == 25344 and t.n_seg == 1 and int((t.flags & 2 != 0).sum()) == 1005


def test_malformed_numbers_are_rejected(tmp_path):
    from mcaller_amd import _lib
    p = str(tmp_path / 'bad.tsv')
    open(p, 'w').write(('c\t12\tACGTAC\tr\tt\t5\t80.1x\t1\t1\tACGTAC\t80.00\t1\t0\n') * 20)
    with pytest.raises(_lib.McError):
        _lib.parse_eventalign(p, 0, os.path.getsize(p), ['c'])


def test_read_cuts_and_exact_ranges(tmp_path):
    """mc_eventalign_read_cuts + mc_parse_eventalign_range: pieces cut at read starts, parsed one by one, concatenate to the
    table of the whole file; no read name occurs in two pieces."""
    from mcaller_amd import synth, _lib
    codes = synth.genome(length=200000, seed=3)
    table, qual = synth.make_table(60000, seed=8, codes=codes, read_len=(300, 1500))
    tsv = str(tmp_path / 'syn.eventalign.tsv')
    synth.write_tsv(table, codes, tsv)
    size = os.path.getsize(tsv)
    whole = _lib.parse_eventalign(tsv, 0, size, ['ecoli_syn'], exact_range=True)
    assert whole.n_rows == table.n_rows and (whole.pos == table.pos).all()
    for n_parts in (1, 2, 3, 7, 200):
        cuts = _lib.eventalign_read_cuts(tsv, n_parts)
        assert len(cuts) == n_parts + 1 and cuts[0] == 0 and cuts[-1] == size and cuts == sorted(cuts)
        parts = [_lib.parse_eventalign(tsv, cuts[i], cuts[i + 1], ['ecoli_syn'], n_threads=1 + i % 3, exact_range=True)
                 for i in range(n_parts)]
        assert sum(p.n_rows for p in parts) == whole.n_rows
        for col in ('pos', 'event_e4', 'model_e4', 'event_idx', 'flags'):
            assert (np.concatenate([getattr(p, col) for p in parts]) == getattr(whole, col)).all(), col
        names = [n for p in parts for n in p.read_names]
        assert names == whole.read_names                      # every read in exactly one piece, order kept
        if n_parts in (2, 3, 7):
            rows = [p.n_rows for p in parts]
            assert min(rows) > 0.5 * whole.n_rows / n_parts    # balanced
    # ... at offsets of the caller's choosing (a stream's small first shards, extract_contexts.shard_schedule): every cut is the first
    # read start at or behind its offset, the pieces concatenate to the whole
    from mcaller_amd import extract_contexts as ec
    text = open(tsv, 'rb').read()
    starts = [0]
    prev = None
    at = 0
    for line in text.splitlines(True):
        name = line.split(b'\t')[3]
        if prev is not None and name != prev:
            starts.append(at)
        prev = name
        at += len(line)
    starts = np.array(starts + [size])
    lo, hi = int(starts[3]), int(starts[-4])
    for want in ([lo + 1], [lo + 5000, lo + 5001, lo + 300000, hi - 10], ec.shard_schedule(lo, hi), [hi + 7], []):
        cuts = _lib.eventalign_read_cuts_at(tsv, want, lo, hi)
        assert len(cuts) == len(want) + 2 and cuts[0] == lo and cuts[-1] == hi and cuts == sorted(cuts)
        for w, c in zip(want, cuts[1:-1]):
            first_start = int(starts[np.searchsorted(starts, min(w, hi), side='left')])
            assert c == min(first_start, hi), (w, c)
        parts = [_lib.parse_eventalign(tsv, cuts[i], cuts[i + 1], ['ecoli_syn'], exact_range=True) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
        ref_rows = _lib.parse_eventalign(tsv, lo, hi, ['ecoli_syn'], exact_range=True)
        assert (np.concatenate([p.pos for p in parts]) == ref_rows.pos).all()
    sched = ec.shard_schedule(0, 3 << 30)
    sizes = np.diff([0] + sched + [3 << 30])
    assert len(sizes) >= ec.STREAM_MIN_SHARDS and sizes[0] * 7 < sizes[5] and sizes[-1] * 3 < sizes[5] and sizes.max() <= 1.1 * ec.STREAM_SHARD_BYTES


def test_thread_counts_follow_the_affinity_mask(tmp_path):
    """A worker bound to the cores next to its GPU (mc_bind_to_device_numa_node) must not start one parser thread per
    MACHINE core: with n_threads <= 0 the pieces are counted from sched_getaffinity."""
    import os
    from mcaller_amd import synth, _lib
    codes = synth.genome(length=300000, seed=2)
    table, _ = synth.make_table(400000, seed=3, codes=codes, read_len=(500, 3000))
    tsv = str(tmp_path / 'big.eventalign.tsv')
    synth.write_tsv_native(table, codes, tsv)
    assert os.path.getsize(tsv) > 40 << 20                    # enough text for > 8 pieces of 4 MB
    before = os.sched_getaffinity(0)
    try:
        for cores in (1, 3, 4):
            os.sched_setaffinity(0, set(sorted(before)[:cores]))
            if len(os.sched_getaffinity(0)) != cores:
                pytest.skip('cannot restrict the affinity mask here')
            assert _lib.lib().mc_host_cores() == cores
            t = _lib.parse_eventalign(tsv, 0, os.path.getsize(tsv), ['ecoli_syn'])
            assert 1 <= t.n_pieces <= cores and t.n_rows == table.n_rows
            assert (t.pos == table.pos).all() and (t.evmu == table.evmu).all()
    finally:
        os.sched_setaffinity(0, before)
    assert _lib.lib().mc_host_cores() == len(before)


def test_native_tsv_writer_equals_the_python_one(tmp_path):
    from mcaller_amd import synth
    codes = synth.genome(length=100000, seed=5)
    table, _ = synth.make_table(30000, seed=8, codes=codes, read_len=(200, 1500))
    a, b = str(tmp_path / 'a.tsv'), str(tmp_path / 'b.tsv')
    synth.write_tsv(table, codes, a)
    for nt in (1, 3, 0):
        assert synth.write_tsv_native(table, codes, b, n_threads=nt) == os.path.getsize(a)
        assert open(a, 'rb').read() == open(b, 'rb').read()


def test_read_file_range_is_the_files_bytes(tmp_path):
    """mc_read_file_range (what feeds the device parser): any byte range of a file, pieces read by several threads."""
    import ctypes as C
    from mcaller_amd import _lib
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 9_500_000, dtype=np.uint8).tobytes()
    p = str(tmp_path / 'blob.bin')
    open(p, 'wb').write(data)
    L = _lib.lib()
    for lo, hi, nt in ((0, len(data), 0), (1, len(data) - 1, 3), (4_194_303, 4_194_305, 2), (5_000_000, 5_000_000, 1),
                       (123, 8_388_731, 7)):
        buf = np.zeros(max(hi - lo, 1), dtype=np.uint8)
        _lib.check(L.mc_read_file_range(p.encode(), lo, hi, buf.ctypes.data, nt))
        assert buf[:hi - lo].tobytes() == data[lo:hi]
    buf = np.zeros(16, dtype=np.uint8)
    assert L.mc_read_file_range(p.encode(), len(data) - 8, len(data) + 8, buf.ctypes.data, 1) != 0      # beyond the end
    assert L.mc_read_file_range((p + '.nope').encode(), 0, 8, buf.ctypes.data, 1) != 0
    t = _lib.TextBlock(p, 10, 1000)
    assert t.n_bytes == 990 and bytes(t.array[:990]) == data[10:1000] and t.token(5, 3) == data[15:18].decode('utf-8', 'surrogateescape')
