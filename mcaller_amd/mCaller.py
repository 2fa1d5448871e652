#!/usr/bin/env python3
"""mCaller command line on MI355X -- same flags, output naming and messages as the reference's mCaller.py:118-184.

`-t/--threads` is accepted for compatibility: the GPU path is one process per GPU and always produces the reference's
single-process (`-t 1`) row order; with `-t N>1` the rows are passed through the `sort -n -k2 | uniq` the reference applies
after its workers (mCaller.py:106): numeric prefix of the read name, then the whole line in the C locale.
"""
import glob
import os
import sys

assert sys.version_info >= (3, 0), 'please use python3'

from .extract_contexts import extract_features
from .read_qual import extract_read_quality
from .refmark import read_fasta


def pos2label(positions):
    """{(chrom, pos, strand): label} from a positions file (train_model.py:18-20)."""
    return {(pos.split()[0], int(pos.split()[1]), pos.split()[2]): pos.split()[3]
            for pos in open(positions, 'r').read().split('\n') if len(pos.split()) > 1}


def numeric_key_k2(line):
    """What `sort -n -k2` compares first (the reference merges its workers' files with `sort -n -k2 | uniq`, mCaller.py:106):
    the key runs from the end of field 1 to the end of the line; its leading number -- blanks, an optional '-', digits,
    an optional '.' and digits -- compares numerically, anything else counts as 0.  Read names are UUIDs that often start
    with digits: '2289b392-...' sorts as 2289, 'cc1d...' as 0."""
    from decimal import Decimal
    i, n = 0, len(line)
    while i < n and line[i:i + 1] not in (b' ', b'\t'):       # field 1
        i += 1
    while i < n and line[i:i + 1] in (b' ', b'\t'):
        i += 1
    j = i
    if j < n and line[j:j + 1] == b'-':
        j += 1
    d0 = j
    while j < n and line[j:j + 1].isdigit():
        j += 1
    if j < n and line[j:j + 1] == b'.':
        j2 = j + 1
        while j2 < n and line[j2:j2 + 1].isdigit():
            j2 += 1
        if j2 > j + 1:
            j = j2
    if j == d0:
        return Decimal(0)
    return Decimal(line[i:j].decode('ascii'))


def merge_like_sort_uniq(paths, out_path):
    """`sort -n -k2 | uniq` over the tmp files: numeric key of field 2, ties by the whole line (bytes, the C locale's
    order -- the reference's last-resort order depends on the user's locale), duplicates dropped."""
    lines = set()
    for path in paths:
        with open(path, 'rb') as fh:
            lines.update(fh.read().splitlines(True))
        os.remove(path)
    with open(out_path, 'wb') as out:
        out.writelines(sorted(lines, key=lambda l: (numeric_key_k2(l), l)))


class Run(object):
    """One invocation: what mCaller.py:25-115 does around extract_features, without its process fan-out (the GPU path is one
    process per GPU and writes the single-process row order; `-t` only selects the reference's merge step)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.stem = '.'.join(self.tsv.split('.')[:-1])
        self.output = self.stem + '.diffs.' + str(self.k) + ('.train' if self.train else '')      # mCaller.py:31-35

    def extract(self):
        """-> (signals, contexts) in train mode, None else; leaves `<stem>.diffs.<k>[.train].tmp0` behind."""
        if self.training_tsv:                               # mCaller.py:35-36: the matrix comes from an earlier run
            from .load_mCaller_data import tsv2matrix
            return tsv2matrix(self.training_tsv, self.base)
        labels = pos2label(self.positions) if self.train else None
        if (self.n_gpus > 1 and self.train) or (not self.train and (self.n_gpus > 1 or self.bed)):     # reads shard over the GPUs of the node (multi_gpu.py)
            from .multi_gpu import extract_features_sharded
            if self.bed:
                self.bed = dict(self.bed, path=self.output.split('.')[0] + '.methylation.summary.bed')   # make_bed.py:190
            self.sharded = extract_features_sharded(self.tsv, self.reference, self.read2qual, self.k, self.skip_thresh,
                                                    self.qual_thresh, self.modelfile, self.base, self.motif, self.positions,
                                                    self.n_gpus, bed=None if self.train else self.bed, fastq=getattr(self, 'fastq', None),
                                                    train=self.train, pos_label=labels)
            stats_path = os.environ.get('MCALLER_STATS_JSON')
            if stats_path:                                  # what the run measured (per worker: rows, seconds; the reduction)
                import json
                from . import multi_gpu
                with open(stats_path, 'w') as fh:
                    json.dump(multi_gpu.last_run, fh)
            if self.sharded:
                if self.train:
                    from . import multi_gpu
                    return multi_gpu.train_dicts
                return None
        return extract_features(self.tsv, self.reference, self.read2qual, self.k, self.skip_thresh, self.qual_thresh,
                                self.modelfile, self.classifier, 0, endline=os.path.getsize(self.tsv), train=self.train,
                                pos_label=labels, base=self.base, motif=self.motif, positions_list=self.positions)

    def collect(self):
        """The tmp file(s) become `<stem>.diffs.<k>[.train]` (mCaller.py:94-109)."""
        if self.training_tsv:
            return
        tmp = glob.glob(self.stem + '*.tmp[0-9]*')
        if self.threads > 1:
            print('Merging files...')
            merge_like_sort_uniq(tmp, self.output)
        else:
            os.rename(tmp[0], self.output)

    def summarise(self):
        """--bed without a sharded run (it declined: a read name in two pieces, an exit path ...) and --bed_vo: from the rows
        just written -- the concatenated file IS the gather of the workers' call records."""
        if not self.bed or self.train or self.training_tsv:
            return
        from . import make_bed
        from . import multi_gpu
        if not self.sharded or not multi_gpu.bed_written:       # (the sharded run declined, or its reduction did not finish)
            make_bed.summarise_diffs(self.output, self.bed['path'], self.bed['min_depth'], self.bed['mod_threshold'])
        if self.bed.get('vo'):
            # make_bed.py --vo's per-read probability lists (:114-115); the columns before them must be the reduced ones
            reduced = open(self.bed['path']).read().splitlines() if os.path.exists(self.bed['path']) else None
            make_bed.summarise_diffs(self.output, self.bed['path'], self.bed['min_depth'], self.bed['mod_threshold'],
                                     with_probs=True, quiet=True)
            if reduced is not None:
                verbose = [line.rsplit('\t', 1)[0] for line in open(self.bed['path']).read().splitlines()]
                if verbose != reduced:
                    raise RuntimeError('the per-site reduction and the rows written disagree on the BED file')

    def go(self):
        outdir = '/'.join(self.tsv.split('/')[:-1])
        print(outdir + '/' if len(outdir) > 1 else outdir)
        print('%d contigs' % self.num_refs)
        print('%d threads' % self.threads)
        self.sharded = False
        matrices = self.extract()
        print('Finished extracting signals')
        self.collect()
        self.summarise()
        if self.train:
            print('Training...')
            from .train_model import train_classifier
            train_classifier(matrices[0], matrices[1], self.modelfile, self.classifier, self.plot_training)
            print('Finished training')


# The reference's command line (mCaller.py:122-141), flag for flag and help text for help text, as data; then this build's
# own options.  (flags, keyword arguments of add_argument)
_SITE_CHOICE = (
    (('-p', '--positions'), dict(type=str, help='file with a list of positions at which to classify bases (must be formatted as '
                                                'space- or tab-separated file with chromosome, position, strand, and label if training)')),
    (('-m', '--motif'), dict(type=str, help='classify every base of type --base in the motif specified instead (can be single one-mer)')),
)
_REFERENCE_FLAGS = (
    (('-r', '--reference'), dict(type=str, required=True, help='fasta file with reference aligned to')),
    (('-e', '--tsv'), dict(type=str, required=True, help='tsv file with nanopolish event alignment')),
    (('-f', '--fastq'), dict(type=str, required=True, help='fastq file with nanopore reads')),
    (('-t', '--threads'), dict(type=int, default=1, help='specify number of processes (default = 1)')),
    (('-b', '--base'), dict(type=str, default='A', help='bases to classify as methylated or unmethylated (A or C, default A)')),
    (('-n', '--num_variables'), dict(type=int, default=6, help='change the length of the context used to classify (default of 6 '
                                                               'variables corresponds to 11-mer context (6*2-1))')),
    (('--train',), dict(action='store_true', default=False, help='train a new model (requires labels in positions file)')),
    (('--training_tsv',), dict(type=str, help='mCaller output file for training')),
    (('-d', '--modelfile'), dict(type=str, help='model file name')),
    (('-s', '--skip_thresh'), dict(type=int, default=0, help='number of skips to allow within an observation (default 0)')),
    (('-q', '--qual_thresh'), dict(type=float, default=0, help='quality threshold for reads (default none)')),
    (('-c', '--classifier'), dict(type=str, default='NN', help='use alternative classifier: options = NN (default), RF, LR, or NBC '
                                                               '(non-default may significantly increase runtime)')),
    (('--plot_training',), dict(action='store_true', default=False, help='plot probabilities distributions for training positions '
                                                                         '(requires labels in positions file and --train)')),
    (('-v', '--version'), dict(action='version', version='%(prog)s v1.0', help='print version')),
)
_OWN_FLAGS = (
    (('--gpus',), dict(type=int, default=None, help='(mcaller_amd) GPUs of this node to shard the reads over (default 1, or $MCALLER_GPUS)')),
    (('--bed',), dict(action='store_true', default=False,
                      help='(mcaller_amd) also write <stem>.methylation.summary.bed from the per-site reduction of the calls '
                           '(device-side counts, ncclAllReduce over the GPUs): what make_bed.py -f <diffs> writes')),
    (('--bed_vo',), dict(action='store_true', default=False,
                         help='(mcaller_amd) with --bed: append the per-read probabilities of every site, like make_bed.py --vo')),
    (('--bed_min_depth',), dict(type=int, default=15, help='(mcaller_amd) make_bed -d for --bed')),
    (('--bed_mod_threshold',), dict(type=float, default=0.5, help='(mcaller_amd) make_bed -t for --bed')),
)


def build_parser():
    from argparse import ArgumentParser
    parser = ArgumentParser(description='Classify bases as methylated or unmethylated', prog='mCaller')
    one_of = parser.add_mutually_exclusive_group(required=True)
    for flags, kw in _SITE_CHOICE:
        one_of.add_argument(*flags, **kw)
    for flags, kw in _REFERENCE_FLAGS + _OWN_FLAGS:
        parser.add_argument(*flags, **kw)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    n_gpus = args.gpus if args.gpus is not None else int(os.environ.get('MCALLER_GPUS', '1'))
    mods = {'A': 'm6A', 'C': 'm5C'}                                   # mCaller.py:144-150
    if args.base not in mods:
        print('classification only available for A or C bases so far')
        sys.exit(0)
    modelfile = args.modelfile or '%s/model_%s_%d_%s.pkl' % (os.path.dirname(os.path.realpath(sys.argv[0])), args.classifier,
                                                              args.num_variables, mods[args.base])
    assert args.train or os.path.isfile(modelfile), 'model file not found at ' + modelfile
    assert (args.skip_thresh < args.num_variables / 2), ('too many skips with only ' + str(args.num_variables) +
                                                         ' variables - try < half')
    assert os.path.isfile(args.fastq), 'fastq file not found at ' + args.fastq
    # read qualities (native FASTQ reader) and the contig count, side by side
    import threading
    box = {}

    def qualities():
        try:
            box['r2q'] = extract_read_quality(args.fastq)
        except BaseException as e:                                   # noqa
            box['err'] = e
    th = threading.Thread(target=qualities)
    th.start()
    try:
        num_refs = len(read_fasta(args.reference))
    except IOError:
        th.join()
        print('reference file missing')
        sys.exit(0)
    th.join()
    if 'err' in box:
        raise box['err']
    bed = dict(min_depth=args.bed_min_depth, mod_threshold=args.bed_mod_threshold, vo=args.bed_vo) if args.bed else None
    Run(tsv=args.tsv, reference=args.reference, read2qual=box['r2q'], num_refs=num_refs, positions=args.positions,
        motif=args.motif, base=args.motif if (args.motif and len(args.motif) == 1) else args.base, k=args.num_variables,
        threads=args.threads, train=args.train, training_tsv=args.training_tsv or None, modelfile=modelfile,
        skip_thresh=args.skip_thresh, qual_thresh=args.qual_thresh, classifier=args.classifier,
        plot_training=args.plot_training, n_gpus=max(1, n_gpus), bed=bed, fastq=args.fastq).go()


if __name__ == '__main__':
    main()
