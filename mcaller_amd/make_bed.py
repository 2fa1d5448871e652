#!/usr/bin/env python3
"""Per-position summary of mCaller calls: what the reference's make_bed.py:67-164 writes, from two sources --

  * a `.diffs.<n>` file (`summarise_diffs`, the command line below): rows -> SiteRows -> BED / GFF;
  * flush records on the GPUs (`site_counts`, `write_bed_from_counts`): per-site counts reduced on the device and summed
    over ranks with one all-reduce (mc_site_allreduce), the one exchange step of the multi-GPU path.

Options: -f, -d, -t, -p (per-position one-sample t-tests, make_bed.py:115-127; needs scipy, like the reference), --control,
--vo, --gff (with --vo: fracLow/fracUp/identificationQv, make_bed.py:146-149), --ref.  Plotting is out of scope; the
reference's two debugging prints (the dict of feature rows, the output name once per row: make_bed.py:101,:153) are not
reproduced.
"""
import os
import sys

import numpy as np

from .refmark import read_fasta, revcomp


class SiteRows(object):
    """The per-site table of a `.diffs` file -- the text-side twin of the device reduction (mc_site_counts): one entry per
    (chrom, pos, strand, context) in first-occurrence order (make_bed.py:86-96,:134), with what the chosen options need:
    methylated / total calls, the printed probabilities (--vo), the feature rows (-p)."""

    def __init__(self, keep_probs=False, keep_features=False):
        self.slot = {}                      # (chrom, pos, strand, context) -> entry number
        self.chrom, self.pos, self.strand, self.context = [], [], [], []
        self.n_meth, self.depth = [], []
        self.probs = [] if keep_probs else None
        self.features = [] if keep_features else None

    def add(self, chrom, pos, strand, context, is_meth, prob_txt='', features=None):
        key = (chrom, pos, strand, context)
        i = self.slot.get(key)
        if i is None:
            i = self.slot[key] = len(self.chrom)
            self.chrom.append(chrom); self.pos.append(pos); self.strand.append(strand); self.context.append(context)
            self.n_meth.append(0); self.depth.append(0)
            if self.probs is not None:
                self.probs.append([])
            if self.features is not None:
                self.features.append([])
        self.depth[i] += 1
        self.n_meth[i] += 1 if is_meth else 0
        if self.probs is not None:
            self.probs[i].append(prob_txt)
        if self.features is not None:
            self.features[i].append(features)
        return i

    def __len__(self):
        return len(self.chrom)

    def fraction(self, i):
        return np.float64(self.n_meth[i]) / np.float64(self.depth[i])      # == np.mean of the 0/1 list (make_bed.py:143)


def wanted_positions(path):
    """The -p file as a set of (chrom, start, end, strand) text tuples; lines of three characters or fewer are skipped
    (make_bed.py:13-19)."""
    with open(path, 'r') as fh:
        return set(tuple(line.strip().split('\t')[:4]) for line in fh if len(line) > 3)


def read_diffs(path, wanted=None, keep_probs=False):
    """`.diffs.<n>` rows -> SiteRows.  A row counts when the centre of its context is 'M' (and, with -p, when its position
    is wanted); 'm...' labels are methylated calls (make_bed.py:76-96)."""
    rows = SiteRows(keep_probs=keep_probs, keep_features=wanted is not None)
    with open(path, 'r') as fh:
        for line in fh:
            f = line.split('\t')
            if len(f) not in (7, 8):
                raise ValueError('not an mCaller row: %r' % line[:80])
            chrom, pos, context, values, strand, label = f[0], f[2], f[3], f[4], f[5], f[6]
            prob_txt = f[7].strip() if len(f) == 8 else ''
            if context[len(context) // 2] != 'M':
                continue
            if wanted is not None and (chrom, pos, str(int(pos) + 1), strand) not in wanted:
                continue
            feats = [float(v) for v in values.split(',')][:-1] if wanted is not None else None
            rows.add(chrom, pos, strand, context, label[0] == 'm', prob_txt, feats)
    return rows


def feature_statistics(feature_rows):
    """-p mode: one-sample t-tests of every feature column against 0 -> [largest t statistic, sum of -log10 p], rounded to
    three decimals (make_bed.py:115-127)."""
    from scipy import stats
    cols = np.asarray(feature_rows, dtype=np.float64)
    tests = [stats.ttest_1samp(cols[:, i], 0) for i in range(cols.shape[1])]
    summed = sum(-np.log10(t[1]) for t in tests)
    return [np.round(x, 3) for x in (max(t[0] for t in tests), summed)]


def reference_contexts(ref_path, rows):
    """--ref: 41 bases around every site, on the read's strand (make_bed.py:36-48)."""
    seqs = dict(read_fasta(ref_path))
    out = {}
    for i in range(len(rows)):
        if rows.chrom[i] in seqs:
            p = int(rows.pos[i])
            window = seqs[rows.chrom[i]][p - 20:p + 21].upper()
            out[i] = revcomp(window) if rows.strand[i] == '-' else window
    return out


def selected(rows, i, wanted, depth_thresh, mod_thresh, control):
    """Does entry i go into the summary?  -p: its position is listed; else depth and fraction thresholds (make_bed.py:21-28,
    :135-138; --control keeps the sites BELOW the fraction threshold)."""
    if wanted is not None:
        return (rows.chrom[i], rows.pos[i], str(int(rows.pos[i]) + 1), rows.strand[i]) in wanted
    if rows.depth[i] < depth_thresh:
        return False
    return (rows.fraction(i) >= mod_thresh) != bool(control)


def gff_attributes(rows, i, context, with_probs):
    frac = rows.fraction(i)
    text = 'coverage=%d;context=%s;IPDRatio=5;frac=%s' % (rows.depth[i], context, str(frac))
    if with_probs:                                                     # make_bed.py:146-149
        probs = np.array([float(x) for x in rows.probs[i]], dtype=np.float64)
        se_95 = 2 * (np.std(probs, ddof=1) / np.sqrt(len(probs)))      # 2 x the standard error of the mean
        text += ';fracLow=%s;fracUp=%s;identificationQv=%s' % (str(frac - se_95), str(frac + se_95),
                                                               str(int(100 * np.mean(probs))))
    return text


def summarise_diffs(diffs_path, out_path, depth_thresh, mod_thresh, positions=None, control=False, with_probs=False,
                    gff=False, ref=None, quiet=False):
    """make_bed.py:67-164 without the plotting: `.diffs.<n>` -> BED (or GFF) of the selected sites, in first-occurrence
    order.  Returns the number of sites written."""
    wanted = wanted_positions(positions) if positions else None
    rows = read_diffs(diffs_path, wanted, keep_probs=with_probs)
    ref_context = reference_contexts(ref, rows) if ref else None
    count = 0
    with open(out_path, 'w') as out:
        for i in range(len(rows)):
            if not selected(rows, i, wanted, depth_thresh, mod_thresh, control):
                continue
            count += 1
            context = ref_context[i] if ref else rows.context[i]     # (KeyError for an unknown contig, like the reference)
            end = str(int(rows.pos[i]) + 1)
            if gff:
                out.write('\t'.join([rows.chrom[i], 'kinModCall', 'm6A', end, end, '10', rows.strand[i], '.',
                                     gff_attributes(rows, i, context, with_probs)]) + '\n')
                continue
            # (the BED row keeps the row's own context: --ref only reaches the GFF attributes, make_bed.py:142,:155)
            cols = [rows.chrom[i], rows.pos[i], end, rows.context[i], str(rows.fraction(i)), rows.strand[i], str(rows.depth[i])]
            if wanted is not None:
                cols += [str(x) for x in feature_statistics(rows.features[i])]
            if with_probs:
                cols.append(','.join(rows.probs[i]))
            out.write('\t'.join(cols) + '\n')
    if wanted is None and not quiet:
        print(count, 'unmethylated' if control else 'methylated', 'loci found with min depth', depth_thresh, 'reads')
    return count


# ---- the same reduction from flush records, summed over ranks -----------------------------------------------------
class SiteIndex(object):
    """All marked sites ('M' of meth_fwd / meth_rev, extract_contexts.py:60-73) of the marked contigs, numbered in
    (contig, strand, position) order: the key space of the per-site reduction (~2 x 18k sites for E. coli GATC)."""

    def __init__(self, meth_strings, n_contigs):
        self.sites, self.base = {}, {}
        n = 0
        for c in range(n_contigs):
            for rev in (0, 1):
                if c in meth_strings:
                    arr = np.frombuffer(meth_strings[c][rev].encode('latin1'), dtype=np.uint8)
                    pos = np.flatnonzero(arr == ord('M')).astype(np.int64)
                else:
                    pos = np.zeros(0, dtype=np.int64)
                self.sites[(c, rev)] = pos
                self.base[(c, rev)] = n
                n += len(pos)
        self.n = n

    def keys(self, contig, rev, pos):
        """Site numbers of (contig[i], rev[i], pos[i]); every pos must be a marked site."""
        out = np.empty(len(pos), dtype=np.int64)
        for c in np.unique(contig):
            for r in (0, 1):
                sel = (contig == c) & (rev == r)
                if sel.any():
                    s = self.sites[(int(c), r)]
                    j = np.searchsorted(s, pos[sel])
                    if (j >= len(s)).any() or (s[np.minimum(j, len(s) - 1)] != pos[sel]).any():
                        raise ValueError('a record names a position that is not a marked site')
                    out[sel] = self.base[(int(c), r)] + j
        return out

    def locate(self, key):
        """site number -> (contig, rev, pos)."""
        for (c, r), b in self.base.items():
            s = self.sites[(c, r)]
            if b <= key < b + len(s):
                return c, r, int(s[key - b])
        raise KeyError(key)


def site_counts(rec, table, index, row_offset=0, prob=None, skip=None):
    """Per site of `index`: n_meth, n_total (int32) and the global row of the first occurrence (int64, max = none), from
    this rank's records (scored, not skipped).  prob: probability per record where the host scored some itself (default
    rec.prob); skip: records to leave out (cross_contig_records)."""
    from . import _lib
    n_meth = np.zeros(index.n, dtype=np.int32)
    n_total = np.zeros(index.n, dtype=np.int32)
    first = np.full(index.n, np.iinfo(np.int64).max, dtype=np.int64)
    rec = rec.by_record()
    n = rec.n
    info = rec.info[:n]
    ok = (info & _lib.I_TOO_MANY) == 0
    if skip is not None:
        ok &= ~np.asarray(skip, dtype=bool)[:n]
    p = rec.prob[:n] if prob is None else np.asarray(prob)[:n]
    if ok.any():
        contig = table.seg_contig[rec.site_seg[:n][ok]].astype(np.int64)
        rev = ((info[ok] & _lib.I_REV) != 0).astype(np.int64)
        key = index.keys(contig, rev, rec.site_pos[:n][ok].astype(np.int64))
        np.add.at(n_total, key, 1)
        np.add.at(n_meth, key, (p[ok] >= 0.5).astype(np.int32))
        np.minimum.at(first, key, rec.close_row[:n][ok] + row_offset)
    return n_meth, n_total, first


def cross_contig_records(rec, table, ref, k, host_scored=None, tail_chrom=None, row_offset=0):
    """Records whose row carries another contig than their site lies on: the chrom column is the contig of the row that
    CLOSED the window (R8, extract_contexts.py:216), make_bed keys on it, and (that contig, position) is no marked site --
    at most one per read that ends a contig.  -> dict(records: bool per record, rows: [(chrom, pos, strand, context,
    is_meth, first global row)]): the per-site reductions leave these records out and the BED writer adds the rows."""
    from . import _lib
    rec = rec.by_record()
    n = rec.n
    info = rec.info[:n]
    mask = np.zeros(n, dtype=bool)
    rows = []
    if n == 0:
        return dict(records=mask, rows=rows)
    site_contig = table.seg_contig[rec.site_seg[:n]].astype(np.int64)
    close_seg = np.searchsorted(table.seg_row_begin, rec.close_row[:n], side='right') - 1
    tail_id = ref.names.index(tail_chrom) if tail_chrom is not None else -1
    close_contig = np.where(close_seg >= table.n_seg, tail_id,
                            table.seg_contig[np.minimum(close_seg, max(table.n_seg - 1, 0))].astype(np.int64))
    # (closed by a row beyond the table with no contig behind it: the kernels emit no such record -- R6, the window is lost)
    if ((close_contig < 0) & ((info & _lib.I_TOO_MANY) == 0)).any():
        raise ValueError('a record is closed by a row beyond the table, but no contig follows the table')
    mask = ((info & _lib.I_TOO_MANY) == 0) & (close_contig != site_contig)
    for j in np.flatnonzero(mask):
        rev = bool(info[j] & _lib.I_REV)
        pos = int(rec.site_pos[j])
        context = revcomp(ref.meth[int(site_contig[j])][1 if rev else 0][pos - k + 1:pos + k], rev)
        p = (host_scored or {}).get(int(j), rec.prob[j])
        rows.append((ref.names[int(close_contig[j])], pos, '-' if rev else '+', context, bool(p >= 0.5),
                     int(rec.close_row[j]) + row_offset))
    return dict(records=mask, rows=rows)


def add_pending_site_counts(dev, rec, table, index, row_offset=0, prob=None, skip=None):
    """Records the device could not score (NaN there; the host scored them: `prob`, default rec.prob) -> added to the
    device-side counts of Device.site_counts()."""
    from . import _lib
    rec = rec.by_record()
    n = rec.n
    info = rec.info[:n]
    sel = ((info & _lib.I_TOO_MANY) == 0) & np.isnan(rec.prob[:n])
    if skip is not None:
        sel &= ~np.asarray(skip, dtype=bool)[:n]
    if prob is None or not sel.any():
        return 0
    p = np.asarray(prob)[:n]
    contig = table.seg_contig[rec.site_seg[:n][sel]].astype(np.int64)
    rev = ((info[sel] & _lib.I_REV) != 0).astype(np.int64)
    key = index.keys(contig, rev, rec.site_pos[:n][sel].astype(np.int64))
    dev.site_counts_add(key, (p[sel] >= 0.5).astype(np.uint8), rec.close_row[:n][sel] + row_offset)
    return int(sel.sum())


def write_bed_from_counts(aggfi, n_meth, n_total, first, index, contig_names, meth_strings, k, depth_thresh, mod_thresh,
                          control=False, extras=()):
    """BED rows in first-occurrence order (make_bed.py:134,154-159) from reduced counts; `extras`: the rows of
    cross_contig_records (all ranks), which are entries of their own."""
    entries = []                                        # (first row, chrom, pos, context, strand, meth, depth)
    # (the selection of :143-154 on the count arrays first: only the sites that will be written get a context string)
    n_meth, n_total = np.asarray(n_meth), np.asarray(n_total)
    seen = n_total > 0
    with np.errstate(divide='ignore', invalid='ignore'):
        frac_all = n_meth.astype(np.float64) / n_total.astype(np.float64)
    keep = seen & (n_total >= depth_thresh) & ((frac_all >= mod_thresh) != bool(control))
    for key in np.nonzero(keep)[0]:
        c, rev, pos = index.locate(int(key))
        context = revcomp(meth_strings[c][rev][pos - k + 1:pos + k], bool(rev))
        entries.append([int(first[key]), contig_names[c], pos, context, '-' if rev else '+', int(n_meth[key]), int(n_total[key])])
    merged = {}
    for chrom, pos, strand, context, is_meth, row in extras:
        e = merged.get((chrom, pos, strand, context))
        if e is None:
            e = merged[(chrom, pos, strand, context)] = [row, chrom, pos, context, strand, 0, 0]
            entries.append(e)
        e[0] = min(e[0], row)
        e[5] += 1 if is_meth else 0
        e[6] += 1
    entries.sort(key=lambda e: e[0])
    count = 0
    with open(aggfi, 'w') as outfi:
        for _, chrom, pos, context, strand, meth, depth in entries:
            frac = np.float64(meth) / np.float64(depth)
            if depth < depth_thresh or ((frac >= mod_thresh) == bool(control)):
                continue
            outfi.write('\t'.join([chrom, str(pos), str(pos + 1), context, str(frac), strand, str(depth)]) + '\n')
            count += 1
    return count


def main(argv=None):
    from argparse import ArgumentParser
    parser = ArgumentParser(description='Produce bed file of methylated positions based on mCaller output')
    parser.add_argument('-d', '--min_read_depth', type=int, required=False, default=15)
    parser.add_argument('-t', '--mod_threshold', type=float, required=False, default=0.5)
    parser.add_argument('-f', '--mCaller_file', type=str, required=True)
    parser.add_argument('-p', '--positions', type=str, required=False)
    parser.add_argument('--control', action='store_true', required=False)
    parser.add_argument('--gff', action='store_true', required=False)
    parser.add_argument('--ref', type=str, required=False)
    parser.add_argument('--plot', action='store_true', required=False)
    parser.add_argument('--plotsummary', action='store_true', required=False)
    parser.add_argument('--plotdir', type=str, required=False, default='mCaller_position_plots')
    parser.add_argument('--vo', action='store_true', required=False)
    parser.add_argument('-v', '--version', action='version', version='%(prog)s v1.0')
    args = parser.parse_args(argv)
    assert os.path.isfile(args.mCaller_file), 'file not found at ' + args.mCaller_file
    if args.positions:
        output_file = args.mCaller_file.split('.')[0] + '.methylation.positions.summary'
    elif not args.control:
        output_file = args.mCaller_file.split('.')[0] + '.methylation.summary'
    else:
        output_file = args.mCaller_file.split('.')[0] + '.methylation.control.summary'
    output_file = output_file + ('.gff' if args.gff else '.bed')
    print(args.mCaller_file)
    if args.plot or args.plotsummary:
        raise NotImplementedError('plotting is out of scope')
    summarise_diffs(args.mCaller_file, output_file, args.min_read_depth, args.mod_threshold, positions=args.positions,
                    control=args.control, with_probs=args.vo, gff=args.gff, ref=args.ref)


if __name__ == '__main__':
    main()
