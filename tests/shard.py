"""Test / bench helper (not part of the product: the product shards FILES by byte range, mcaller_amd/multi_gpu.py).
Sharding a TABLE by read: a window never spans two name blocks (flush on read change,
extract_contexts.py:179,242), so a table can be cut at name-block boundaries and the shards scanned independently, one
GPU each.  Two things cross a cut: the first unfiltered row after a shard closes the shard's last window (R6) and gives
that record its `chrom` (R8) -> `tail_contig`; and `last_read`, which only matters when a read name occurs in more than
one name block -> such tables are not cut."""
import numpy as np

from mcaller_amd import _lib


def name_block_starts(table):
    """Segment indices that start a name block."""
    return np.nonzero(table.flags[table.seg_row_begin[:-1]] & _lib.F_NAME_START)[0]


def has_repeated_names(table):
    starts = name_block_starts(table)
    reads = table.seg_read[starts]
    return len(np.unique(reads)) != len(reads)


def shard_bounds(table, n_shards):
    """[(seg_lo, seg_hi)] per shard, balanced by rows, cut only at name-block starts.  One shard if a read name
    repeats (the machine's `last_read` would then depend on the previous shard)."""
    if n_shards <= 1 or table.n_seg == 0 or has_repeated_names(table):
        return [(0, table.n_seg)] + [(table.n_seg, table.n_seg)] * (max(n_shards, 1) - 1)
    starts = name_block_starts(table)
    rows_at = table.seg_row_begin[starts]
    cuts = [0]
    for s in range(1, n_shards):
        target = table.n_rows * s // n_shards
        j = int(np.searchsorted(rows_at, target, side='left'))
        j = min(max(j, 0), len(starts) - 1)
        cuts.append(max(int(starts[j]), cuts[-1]))
    cuts.append(table.n_seg)
    return [(cuts[i], cuts[i + 1]) for i in range(n_shards)]


def tail_close(table, qual, qual_thresh, seg_hi):
    """(global row, contig id) of the first unfiltered row at or after segment seg_hi: the row that closes the last
    window of a shard ending there.  (-1, -1): none, that window is lost (R6)."""
    for seg in range(seg_hi, table.n_seg):
        if qual[table.seg_read[seg]] < qual_thresh:
            continue
        r0, r1 = int(table.seg_row_begin[seg]), int(table.seg_row_begin[seg + 1])
        ok = (table.flags[r0:r1] & _lib.F_MODEL_N) == 0
        if ok.any():
            return r0 + int(np.argmax(ok)), int(table.seg_contig[seg])
    return -1, -1


def tail_contig(table, qual, qual_thresh, seg_hi):
    return tail_close(table, qual, qual_thresh, seg_hi)[1]


def concat_records(parts, k, row_offsets, seg_offsets, shard_rows, tail_rows):
    """Shard records (shard-local row/segment indices) -> one Records in file order with global indices.  A record
    closed by the next shard carries close_row == the shard's row count; it gets the true closing row (tail_rows)."""
    n = sum(p.n for p in parts)
    out = _lib.Records(n, k)
    o = 0
    for p, ro, so, nr, tr in zip(parts, row_offsets, seg_offsets, shard_rows, tail_rows):
        m = p.n
        out.feats[o * k:(o + m) * k] = p.feats[:m * k]
        out.site_pos[o:o + m] = p.site_pos[:m]
        out.site_seg[o:o + m] = p.site_seg[:m] + so
        cr = p.close_row[:m].copy()
        out.close_row[o:o + m] = np.where(cr >= nr, tr, cr + ro)
        out.info[o:o + m] = p.info[:m]
        out.prob[o:o + m] = p.prob[:m]
        o += m
    out.n = n
    return out
