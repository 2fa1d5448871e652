"""The fused pass of a dense reference (k1_fused, mcaller_amd/csrc/mc_fused.hip): scan, ordering and emit of a pipelined pass over
a one-base motif as ONE kernel with fixed room per 1024-row piece -- against the C oracle (the literal sequential machine,
extract_contexts.py:147-291), through the C ABI.  Every comparison also asks the library HOW the pass ran (mc_last_pass_info):
a pass that fell back to the scan + emit pair would pass these tests without testing anything.  Runs on a real MI355X only."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev():
    from mcaller_amd.device import Device
    d = Device(0)
    yield d
    d.close()


def wait_fused(dev, orc, k, expect_fused=True, prob_tol=None):
    rec = dev.wait()
    room, rerun = dev.last_pass_info()
    if expect_fused:
        assert room > 0 and not rerun, (room, rerun)
    if prob_tol is None:
        rec.prob[:rec.n] = np.nan
        orc.prob[:orc.n] = np.nan
        H.assert_records_equal(rec, orc, k)
    else:
        H.assert_records_equal(rec, orc, k, prob_tol=prob_tol)
    return rec


@pytest.mark.parametrize('read_len,n_rows,skip,k', [((3000, 20000), 300000, 0, 6), ((40, 300), 200000, 0, 6), ((300, 1500), 150000, 1, 6),
                                                    ((8, 60), 60000, 0, 6), ((1000, 5000), 120000, 2, 7), ((500, 4000), 100000, 0, 4),
                                                    ((2000, 9000), 131072 + 17, 0, 8), ((600, 2500), 4096, 1, 5), ((100, 900), 1025, 0, 1)])
def test_fused_pass_equals_the_oracle(dev, read_len, n_rows, skip, k):
    """Long reads (one name block per piece), short reads (three and more per piece, pieces with more blocks than the table holds:
    the row-per-lane walk), every window length, skips allowed; as a table's first pass (validating), as a later pass, and the
    validating pass again with two in flight."""
    from mcaller_amd import synth
    codes = synth.genome(length=150000, seed=61 + k)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(n_rows, seed=6100 + read_len[0] + skip, codes=codes, read_len=read_len)
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, k, skip, 0.0)
    assert orc.n > n_rows // 40
    dev.set_reference(arrays)
    slot = dev.upload_table_async(table, qual)
    dev.run_async(k, skip, 0.0, score=False)                     # the table's first pass: k1_fused<validating>
    wait_fused(dev, orc, k)
    dev.run_async(k, skip, 0.0, score=False)                     # a validated table: k1_fused<false>
    wait_fused(dev, orc, k)
    dev.select_table(slot, as_new=True)
    dev.run_async(k, skip, 0.0, score=False)
    dev.run_async(k, skip, 0.0, score=False)
    wait_fused(dev, orc, k)
    wait_fused(dev, orc, k)


def test_fused_pass_with_quality_filter_and_tail(dev):
    """Reads dropped by the quality threshold (their rows are nobody's closers), and the first row of the next shard closing
    the table's last window (tail_contig >= 0: R6, R8)."""
    from mcaller_amd import synth
    codes = synth.genome(length=100000, seed=77)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(90000, seed=771, codes=codes, read_len=(200, 2500))
    arrays = ref.device_arrays()
    dev.set_reference(arrays)
    dev.upload_table_async(table, qual)
    for qthr, tail in ((9.0, -1), (0.0, 0), (10.5, 0)):
        orc = H.oracle_records(table, arrays, qual, 6, 0, qthr, tail_contig=tail)
        dev.run_async(6, 0, qthr, tail_contig=tail, score=False)
        wait_fused(dev, orc, 6)


def test_fused_pass_scored_and_site_counts(dev):
    """The classifier and the per-site reduction behind a fused pass: holes are skipped like any record that is not a call, the
    host never sees one (record count, probabilities, per-site counts as from the oracle's records)."""
    from mcaller_amd import synth, make_bed
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome(length=80000, seed=5)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(120000, seed=51, codes=codes, read_len=(1500, 6000))
    arrays = ref.device_arrays()
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model('r95_twobase_model_NN_6_m6A')), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    dev.set_reference(arrays)
    dev.set_mlp(weights, soc)
    dev.upload_table_async(table, qual)
    dev.run_async(6, 0, 0.0, score=True)
    rec = wait_fused(dev, orc, 6, prob_tol=1e-6)
    assert rec.n == orc.n
    index = make_bed.SiteIndex(ref.meth, 1)
    want = make_bed.site_counts(orc, table, index)
    dev.site_counts()
    n_meth, n_total, first = dev.site_counts_fetch()
    assert np.array_equal(n_total, want[1]) and np.array_equal(n_meth, want[0])
    seen = n_total > 0
    assert np.array_equal(first[seen], want[2][seen])


def test_a_piece_out_of_room_repeats_the_pass(dev, monkeypatch):
    """Fixed room per piece that is too small (forced): the pass is marked, repeated by the scan + emit pair inside wait(), and
    the records are the oracle's; the next pass gets twice the room."""
    from mcaller_amd import synth
    codes = synth.genome(length=60000, seed=9)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(50000, seed=91, codes=codes, read_len=(1000, 4000))
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    dev.set_reference(arrays)
    dev.upload_table_async(table, qual)
    monkeypatch.setenv('MCALLER_FUSED_ROOM', '16')
    dev.run_async(6, 0, 0.0, score=False)
    rec = wait_fused(dev, orc, 6, expect_fused=False)
    assert dev.last_pass_info() == (16, True)
    monkeypatch.delenv('MCALLER_FUSED_ROOM')
    dev.run_async(6, 0, 0.0, score=False)
    wait_fused(dev, orc, 6)


def test_the_pair_behind_a_knob(dev, monkeypatch):
    """MCALLER_DENSE_FUSED=0: the scan + emit pair for pipelined passes too (what the synchronous interface always runs)."""
    from mcaller_amd import synth
    codes = synth.genome(length=60000, seed=10)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(40000, seed=92, codes=codes, read_len=(1000, 4000))
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    dev.set_reference(arrays)
    dev.upload_table_async(table, qual)
    monkeypatch.setenv('MCALLER_DENSE_FUSED', '0')
    dev.run_async(6, 0, 0.0, score=False)
    wait_fused(dev, orc, 6, expect_fused=False)
    assert dev.last_pass_info() == (0, False)


def test_fused_pass_at_full_size_dense():
    """10^7 rows, -m A, scored: the records of the fused pass (the product's streaming interface) equal the oracle's."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(10000000, seed=1234, codes=codes)
    arrays = ref.device_arrays()
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model('r95_twobase_model_NN_6_m6A')), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    assert orc.n > 800000
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(6, 0, 0.0, score=True)
        d.run_async(6, 0, 0.0, score=True)
        wait_fused(d, orc, 6, prob_tol=1e-6)
        wait_fused(d, orc, 6, prob_tol=1e-6)
    finally:
        d.close()


def with_model_gaps(table, n_gaps, seed, lengths=(64, 400)):
    """Stretches of 64 and more consecutive 'N' rows (model k-mer NNNNNN: a read across a gap of the model) inside reads, away
    from the reads' first rows."""
    from mcaller_amd import _lib
    rng = np.random.default_rng(seed)
    fl = table.flags
    sb = table.seg_row_begin
    made = 0
    for _ in range(n_gaps * 4):
        s = int(rng.integers(0, table.n_seg))
        lo, hi = int(sb[s]) + 120, int(sb[s + 1]) - 20
        L = int(rng.integers(lengths[0], lengths[1] + 1))
        if hi - lo <= L:
            continue
        a = int(rng.integers(lo, hi - L))
        fl[a:a + L] |= _lib.F_MODEL_N
        fl[a:a + L] &= np.uint8(0xFF ^ _lib.F_KMER_EQ)
        made += 1
        if made == n_gaps:
            break
    assert made >= n_gaps // 2
    return table


@pytest.mark.parametrize('read_len,n_rows,skip', [((3000, 20000), 400000, 0), ((600, 2500), 200000, 1), ((3000, 12000), 300000, 3)])
def test_fused_pass_across_gaps_of_filtered_rows(dev, read_len, n_rows, skip):
    """A read with 64 .. 400 consecutive 'N' rows: the first unfiltered row behind the gap closes the window of the last unfiltered
    row in front of it (:179 skips filtered rows), which lies in front of the rows a piece stages -- a special closer
    (mc_fused.hip: the first run of a block whose tested rows begin in front of the staged rows)."""
    from mcaller_amd import synth
    codes = synth.genome(length=150000, seed=67)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(n_rows, seed=6700 + skip, codes=codes, read_len=read_len)
    table = with_model_gaps(table, 150, seed=5 + skip)
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, skip, 0.0)
    dev.set_reference(arrays)
    slot = dev.upload_table_async(table, qual)
    dev.run_async(6, skip, 0.0, score=False)
    wait_fused(dev, orc, 6)
    dev.run_async(6, skip, 0.0, score=False)
    wait_fused(dev, orc, 6)
    dev.select_table(slot, as_new=True)
    dev.run_async(6, skip, 0.0, score=False)
    dev.run_async(6, skip, 0.0, score=False)
    wait_fused(dev, orc, 6)
    wait_fused(dev, orc, 6)


def test_out_of_room_on_a_first_pass_still_validates(dev, monkeypatch):
    """A table's FIRST pass runs out of room AND the table holds reads that contradict what they were classified on (positions
    going backwards in the middle of a read): the repeat inside wait() is planned as a later pass and classifies on the
    validation flags the first pass left -- which therefore have to be complete, whatever the pieces saw of the overflow."""
    from mcaller_amd import synth
    codes = synth.genome(length=60000, seed=9)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(80000, seed=93, codes=codes, read_len=(1000, 4000))
    rng = np.random.default_rng(17)
    sb = table.seg_row_begin
    for s in rng.choice(table.n_seg, size=min(12, table.n_seg), replace=False):
        lo, hi = int(sb[s]), int(sb[s + 1])
        if hi - lo < 400:
            continue
        a = int(rng.integers(lo + 200, hi - 100))
        table.pos[a:a + 40] -= 30                                 # (the read steps back: irregular, found only where it happens)
    arrays = ref.device_arrays()
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    dev.set_reference(arrays)
    monkeypatch.setenv('MCALLER_FUSED_ROOM', '16')
    for _ in range(2):
        slot = dev.upload_table_async(table, qual)
        dev.run_async(6, 0, 0.0, score=False)
        wait_fused(dev, orc, 6, expect_fused=False)
        assert dev.last_pass_info() == (16, True)
        dev.run_async(6, 0, 0.0, score=False)                     # a later pass over the same table (room still forced small)
        wait_fused(dev, orc, 6, expect_fused=False)


@pytest.mark.parametrize('motif', ['A', 'GATC'])
def test_a_packed_block_that_is_too_small_repeats_the_pass(motif, monkeypatch):
    """The block a pass's records are packed into for the copy-out is sized for the records a table is expected to leave (a fused
    pass: not for every slot of every piece).  Forced too small: the pass says how much it needed (Counters.pack_need), is repeated by
    the synchronous path, and the next pass's block holds it."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome(length=90000, seed=12)
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(200000, seed=94, codes=codes, read_len=(1000, 4000))
    arrays = ref.device_arrays()
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model('r95_twobase_model_NN_6_m6A')), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, 0, 0.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    assert orc.n > 300
    monkeypatch.setenv('MCALLER_PACK_RECORDS', '100')
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        d.run_async(6, 0, 0.0, score=True)
        rec = d.wait()
        assert d.last_pass_info()[1]                              # repeated
        H.assert_records_equal(rec, orc, 6, prob_tol=1e-6)
        d.run_async(6, 0, 0.0, score=True)
        d.run_async(6, 0, 0.0, score=True)
        for _ in range(2):
            rec = d.wait()
            assert not d.last_pass_info()[1]                      # the block has grown
            H.assert_records_equal(rec, orc, 6, prob_tol=1e-6)
    finally:
        d.close()


@pytest.mark.parametrize('grid', [3, 16])
def test_a_stretch_of_nothing_between_stretches_of_records(grid, monkeypatch):
    """Long reads under the quality threshold: sixteen and more consecutive pieces without a record are a stretch of nothing for the
    side stream's kernel, and the stretch behind it has to find its rows in the packed block where the stretch in FRONT of the empty
    one left off (few workgroups, forced: every one of them walks dozens of stretches)."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome(length=200000, seed=31)
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(700000, seed=311, codes=codes, read_len=(18000, 45000))
    qual = qual.copy()
    qual[1::2] = 7.0                                             # every second read is filtered (:167-168), the others are not
    qual[0::2] = 11.0
    arrays = ref.device_arrays()
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model('r95_twobase_model_NN_6_m6A')), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, 0, 9.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    assert orc.n > 20000
    monkeypatch.setenv('MCALLER_SIDE_GRID', str(grid))
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table, qual)
        for _ in range(2):
            d.run_async(6, 0, 9.0, score=True)
            wait_fused(d, orc, 6, prob_tol=1e-6)
    finally:
        d.close()


def test_fused_pass_at_the_headline_size_with_a_quality_filter():
    """10^8 rows, -m A, a quality threshold that drops a third of the reads (5-20 k rows each: whole stretches of pieces without a
    record in the side stream's kernel, 49 stretches a workgroup), scored, two passes in flight: equal to the oracle."""
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(100000000, seed=4242, codes=codes)
    arrays = ref.device_arrays()
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model('r95_twobase_model_NN_6_m6A')), 'A')
    orc = H.oracle_records(table, arrays, qual, 6, 0, 8.0)
    H.oracle_score(orc, table, qual, weights, soc, 6)
    assert 8000000 < orc.n < 9500000
    d = Device(0)
    try:
        d.set_reference(arrays)
        d.set_mlp(weights, soc)
        d.upload_table_async(table.pinned(), qual)
        d.run_async(6, 0, 8.0, score=True)
        d.run_async(6, 0, 8.0, score=True)
        wait_fused(d, orc, 6, prob_tol=1e-6)
        wait_fused(d, orc, 6, prob_tol=1e-6)
    finally:
        d.close()
