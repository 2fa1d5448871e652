#!/usr/bin/env python3
"""Per-position summary of an mCaller `.diffs.<n>` file: the BED writer of the reference's make_bed.py:67-164, plus the
same reduction computed from flush records with an all-reduce over ranks (the one exchange step of the multi-GPU path).

Supported: -f, -d, -t, -p (per-position one-sample t-tests, make_bed.py:115-127; needs scipy, like the reference),
--control, --vo, --gff (with --vo: fracLow/fracUp/identificationQv, make_bed.py:146-149), --ref.  Plotting options are
out of scope.
"""
import os
import sys

import numpy as np

from .refmark import read_fasta, revcomp


def make_pos_set(pos_list):
    """make_bed.py:13-19: (chrom, start, end, strand) of every line longer than 3 characters."""
    pos_set = set()
    with open(pos_list, 'r') as fi:
        for line in fi:
            if len(line) > 3:
                pos_set.add(tuple(line.strip().split('\t')[:4]))
    return pos_set


def check_thresh(locus_list, mod_thresh, depth_thresh, control):
    """make_bed.py:21-28 (returns None below the depth threshold, like the reference)."""
    if len(locus_list) >= depth_thresh:
        if not control and np.mean(locus_list) >= mod_thresh:
            return True
        elif control and np.mean(locus_list) < mod_thresh:
            return True
        else:
            return False


def ref2context(ref, pos_dict):
    """make_bed.py:36-48: +-20 bp context around each locus."""
    ref_dict = {name: seq for name, seq in read_fasta(ref)}
    out = {}
    for pos in pos_dict:
        if pos[0] in ref_dict:
            cx = ref_dict[pos[0]][int(pos[1]) - 20:int(pos[1]) + 21].upper()
            if pos[4] == '-':
                cx = revcomp(cx)
            out[pos] = cx
    return out


def aggregate_by_pos(meth_fi, aggfi, depth_thresh, mod_thresh, pos_list, control, verbose_results, gff, ref,
                     plot=False, plotdir=None, plotsummary=False):
    """make_bed.py:67-164 for the non-plotting, non-positions modes."""
    if plot or plotsummary:
        raise NotImplementedError('plotting is out of scope')
    pos_dict, pos_dict_verbose, values_dict = {}, {}, {}
    pos_set = make_pos_set(pos_list) if pos_list else None
    for line in open(meth_fi, 'r'):
        try:
            csome, read, pos, context, values, strand, label, prob = tuple(line.split('\t'))
        except ValueError:
            csome, read, pos, context, values, strand, label = tuple(line.split('\t'))
            prob = ''
        nextpos = str(int(pos) + 1)
        if (pos_list and (csome, pos, nextpos, strand) not in pos_set) or context[int(len(context) / 2)] != 'M':
            continue
        key = (csome, pos, nextpos, context, strand)
        if key not in pos_dict:
            pos_dict[key] = []
            pos_dict_verbose[key] = []
            values_dict[key] = []
        if pos_list:
            values_dict[key].append([float(v) for v in values.split(',')][:-1])
        pos_dict[key].append(1 if label[0] == 'm' else 0)
        if verbose_results:
            pos_dict_verbose[key].append(prob.strip())
    print(values_dict)                                                 # make_bed.py:101
    if pos_list:                                                       # make_bed.py:115-127
        from scipy import stats
        for locus in values_dict:
            cols = np.asarray(values_dict[locus], dtype=np.float64)
            pvals = []
            for i in range(cols.shape[1]):
                ttest = stats.ttest_1samp(cols[:, i], 0)
                pvals.append((ttest[1], ttest[0]))
            pval = (sum([-np.log10(x[0]) for x in pvals]), max([x[1] for x in pvals]))
            values_dict[locus] = [np.round(x, 3) for x in [pval[1], pval[0]]]
    context_dict = ref2context(ref, pos_dict) if ref else None
    count = 0
    with open(aggfi, 'w') as outfi:
        for locus in pos_dict.keys():
            a = (not pos_list) and check_thresh(pos_dict[locus], mod_thresh, depth_thresh, control)
            b = pos_list and (locus[0], locus[1], locus[2], locus[4]) in pos_set
            if not (a or b):
                continue
            cx = context_dict[locus] if ref else locus[3]
            count += 1
            frac = np.mean(pos_dict[locus])
            if gff:
                deets = 'coverage=' + str(len(pos_dict[locus])) + ';context=' + cx + ';IPDRatio=5;frac=' + str(frac)
                if verbose_results:                                    # make_bed.py:146-149
                    from scipy import stats
                    probs = [float(x) for x in pos_dict_verbose[locus]]
                    se_95 = 2 * stats.sem(probs)
                    deets = (deets + ';fracLow=' + str(frac - se_95) + ';fracUp=' + str(frac + se_95) +
                             ';identificationQv=' + str(int(100 * np.mean(probs))))
                outfi.write('\t'.join([locus[0], 'kinModCall', 'm6A', locus[2], locus[2], '10', locus[4], '.', deets]) + '\n')
            else:
                print(aggfi)
                out_line = '\t'.join(list(locus)[:-1] + [str(np.mean(pos_dict[locus]))] + [locus[-1]] +
                                     [str(len(pos_dict[locus]))])
                if pos_list:
                    out_line = out_line + '\t' + '\t'.join([str(x) for x in values_dict[locus]])
                if verbose_results:
                    out_line = out_line + '\t' + ','.join(pos_dict_verbose[locus])
                outfi.write(out_line + '\n')
    if not pos_list:
        if not control:
            print(count, 'methylated loci found with min depth', depth_thresh, 'reads')
        else:
            print(count, 'unmethylated loci found with min depth', depth_thresh, 'reads')


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


def site_counts(rec, table, index, row_offset=0):
    """Per site of `index`: n_meth, n_total (int32) and the global row of the first occurrence (int64, max = none), from
    this rank's records (scored, not skipped)."""
    from . import _lib
    n_meth = np.zeros(index.n, dtype=np.int32)
    n_total = np.zeros(index.n, dtype=np.int32)
    first = np.full(index.n, np.iinfo(np.int64).max, dtype=np.int64)
    rec = rec.by_record()
    n = rec.n
    info = rec.info[:n]
    ok = (info & _lib.I_TOO_MANY) == 0
    if ok.any():
        contig = table.seg_contig[rec.site_seg[:n][ok]].astype(np.int64)
        rev = ((info[ok] & _lib.I_REV) != 0).astype(np.int64)
        key = index.keys(contig, rev, rec.site_pos[:n][ok].astype(np.int64))
        np.add.at(n_total, key, 1)
        np.add.at(n_meth, key, (rec.prob[:n][ok] >= 0.5).astype(np.int32))
        np.minimum.at(first, key, rec.close_row[:n][ok] + row_offset)
    return n_meth, n_total, first


def add_pending_site_counts(dev, rec, table, index, row_offset=0, prob=None):
    """Records the device could not score (NaN there; the host scored them: `prob`, default rec.prob) -> added to the
    device-side counts of Device.site_counts()."""
    from . import _lib
    rec = rec.by_record()
    n = rec.n
    info = rec.info[:n]
    sel = ((info & _lib.I_TOO_MANY) == 0) & np.isnan(rec.prob[:n])
    if prob is None or not sel.any():
        return 0
    p = np.asarray(prob)[:n]
    contig = table.seg_contig[rec.site_seg[:n][sel]].astype(np.int64)
    rev = ((info[sel] & _lib.I_REV) != 0).astype(np.int64)
    key = index.keys(contig, rev, rec.site_pos[:n][sel].astype(np.int64))
    dev.site_counts_add(key, (p[sel] >= 0.5).astype(np.uint8), rec.close_row[:n][sel] + row_offset)
    return int(sel.sum())


def allreduce_site_counts(n_meth, n_total, first, dist=None, backend_hint=None):
    """Sum / min over ranks through torch.distributed: nccl (= RCCL over xGMI, GPU tensors) or gloo (CPU tensors).
    Messages: 2 x 4 B + 8 B per site (~0.6 MB for E. coli GATC): latency-bound, no custom collective needed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return n_meth, n_total, first
    import torch
    dev = 'cuda' if (backend_hint or dist.get_backend()) == 'nccl' else 'cpu'
    packed = torch.from_numpy(np.stack([n_meth, n_total])).to(dev)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    fmin = torch.from_numpy(first.copy()).to(dev)
    dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
    packed = packed.cpu().numpy()
    return packed[0], packed[1], fmin.cpu().numpy()


def write_bed_from_counts(aggfi, n_meth, n_total, first, index, contig_names, meth_strings, k, depth_thresh, mod_thresh,
                          control=False):
    """BED rows in first-occurrence order (make_bed.py:134,154-159) from reduced counts."""
    keys = np.nonzero(n_total > 0)[0]
    keys = keys[np.argsort(first[keys], kind='stable')]
    count = 0
    with open(aggfi, 'w') as outfi:
        for key in keys:
            depth, meth = int(n_total[key]), int(n_meth[key])
            frac = np.float64(meth) / np.float64(depth)
            if depth < depth_thresh or ((frac >= mod_thresh) == bool(control)):
                continue
            c, rev, pos = index.locate(int(key))
            context = revcomp(meth_strings[c][rev][pos - k + 1:pos + k], bool(rev))
            outfi.write('\t'.join([contig_names[c], str(pos), str(pos + 1), context, str(frac), '-' if rev else '+',
                                   str(depth)]) + '\n')
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
    aggregate_by_pos(args.mCaller_file, output_file, args.min_read_depth, args.mod_threshold, args.positions, args.control,
                     args.vo, args.gff, args.ref, args.plot, args.plotdir, args.plotsummary)


if __name__ == '__main__':
    main()
