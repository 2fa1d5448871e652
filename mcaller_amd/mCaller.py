#!/usr/bin/env python3
"""mCaller command line on MI355X -- same flags, output naming and messages as the reference's mCaller.py:118-184.

`-t/--threads` is accepted for compatibility: the GPU path is one process per GPU and always produces the reference's
single-process (`-t 1`) row order; with `-t N>1` the rows are passed through the same `sort | uniq` the reference applies
after its workers (mCaller.py:106), in the C locale.
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


def distribute_threads(positions_list, motif, tsvname, read2qual, refname, num_refs, base, mod, nprocs, nvariables, train,
                       modelfile, skip_thresh, qual_thresh, classifier, training_tsv, plot_training, n_gpus=1, bed=None):
    """mCaller.py:25-115 without the process fan-out: one GPU pass, then the same file naming."""
    outdir = '/'.join(tsvname.split('/')[:-1])
    if len(outdir) > 1:
        outdir = outdir + '/'
    print(outdir)
    if not train:
        tsv_output = '.'.join(tsvname.split('.')[:-1]) + '.diffs.' + str(nvariables)
        training_pos_dict = None
    else:
        tsv_output = '.'.join(tsvname.split('.')[:-1]) + '.diffs.' + str(nvariables) + '.train'
        if training_tsv:                                    # mCaller.py:35-36
            from .load_mCaller_data import tsv2matrix
            ret = tsv2matrix(training_tsv, base)
        else:
            training_pos_dict = pos2label(positions_list)

    print('%d contigs' % num_refs)
    print('%d threads' % nprocs)
    sharded = False
    if not training_tsv and not train and (n_gpus > 1 or bed):   # reads shard over the GPUs of the node (multi_gpu.py)
        from .multi_gpu import extract_features_sharded
        if bed:
            bed = dict(bed, path=tsv_output.split('.')[0] + '.methylation.summary.bed')       # make_bed.py:190
        sharded = extract_features_sharded(tsvname, refname, read2qual, nvariables, skip_thresh, qual_thresh, modelfile,
                                           base, motif, positions_list, n_gpus, bed=bed)
        ret = None
    if not training_tsv and not sharded:
        bytesize = os.path.getsize(tsvname)
        ret = extract_features(tsvname, refname, read2qual, nvariables, skip_thresh, qual_thresh, modelfile, classifier, 0,
                               endline=bytesize, train=train, pos_label=training_pos_dict, base=base, motif=motif,
                               positions_list=positions_list)
    print('Finished extracting signals')
    tmpfis = [] if training_tsv else glob.glob('.'.join(tsvname.split('.')[:-1]) + '*.tmp[0-9]*')
    if training_tsv:
        pass
    elif nprocs > 1:
        print('Merging files...')
        lines = set()
        for tmpfi in tmpfis:
            with open(tmpfi, 'rb') as fh:
                lines.update(fh.read().splitlines(True))
            os.remove(tmpfi)
        with open(tsv_output, 'wb') as out:                 # `sort -n -k2 | uniq`: field 2 is not numeric, so the
            out.writelines(sorted(lines))                   # order is the last-resort whole-line comparison
    else:
        os.rename(tmpfis[0], tsv_output)

    if bed and not sharded and not train and not training_tsv:
        # the sharded path declined (a read name in two pieces, an exit path ...): the BED comes from the rows just written
        from . import make_bed
        make_bed.aggregate_by_pos(tsv_output, bed['path'], bed['min_depth'], bed['mod_threshold'], None, False, False, False, None)

    if bed and bed.get('vo') and not train and not training_tsv:
        # --bed_vo: the per-read probability lists of make_bed.py --vo (:114-115) come from the rows just written -- the
        # concatenated file IS the gather of the workers' call records; the columns before them must be the reduced ones
        from . import make_bed
        reduced = open(bed['path']).read().splitlines() if os.path.exists(bed['path']) else None
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):      # (make_bed.py prints the file name once per row, :153)
            make_bed.aggregate_by_pos(tsv_output, bed['path'], bed['min_depth'], bed['mod_threshold'], None, False, True, False, None)
        if reduced is not None:
            verbose = [line.rsplit('\t', 1)[0] for line in open(bed['path']).read().splitlines()]
            if verbose != reduced:
                raise RuntimeError('the per-site reduction and the rows written disagree on the BED file')

    if train:
        print('Training...')
        from .train_model import train_classifier
        signal_mat, context_array = ret
        train_classifier(signal_mat, context_array, modelfile, classifier, plot_training)
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
    if args.gpus is None:
        args.gpus = int(os.environ.get('MCALLER_GPUS', '1'))
    mod = {'A': 'm6A', 'C': 'm5C'}.get(args.base)                     # mCaller.py:144-150
    if mod is None:
        print('classification only available for A or C bases so far')
        sys.exit(0)

    if not args.modelfile:
        modelfile = (os.path.dirname(os.path.realpath(sys.argv[0])) + '/model_' + args.classifier + '_' +
                     str(args.num_variables) + '_' + mod + '.pkl')
    else:
        modelfile = args.modelfile
    if not args.train:
        assert os.path.isfile(modelfile), 'model file not found at ' + modelfile

    if args.motif and len(args.motif) == 1:
        base = args.motif
    else:
        base = args.base

    assert (args.skip_thresh < args.num_variables / 2), ('too many skips with only ' + str(args.num_variables) +
                                                         ' variables - try < half')
    assert os.path.isfile(args.fastq), 'fastq file not found at ' + args.fastq
    read2qual = extract_read_quality(args.fastq)

    try:
        num_refs = len(read_fasta(args.reference))
    except IOError:
        print('reference file missing')
        sys.exit(0)

    distribute_threads(args.positions, args.motif, args.tsv, read2qual, args.reference, num_refs, base, mod, args.threads,
                       args.num_variables, args.train, modelfile, args.skip_thresh, args.qual_thresh, args.classifier,
                       args.training_tsv if args.training_tsv else None, args.plot_training, n_gpus=max(1, args.gpus),
                       bed=dict(min_depth=args.bed_min_depth, mod_threshold=args.bed_mod_threshold, vo=args.bed_vo) if args.bed else None)


if __name__ == '__main__':
    main()
