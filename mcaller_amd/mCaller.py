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


def main(argv=None):
    from argparse import ArgumentParser
    parser = ArgumentParser(description='Classify bases as methylated or unmethylated', prog='mCaller')
    all_or_some = parser.add_mutually_exclusive_group(required=True)
    all_or_some.add_argument('-p', '--positions', type=str, required=False, help='file with a list of positions at which to classify bases (must be formatted as space- or tab-separated file with chromosome, position, strand, and label if training)')
    all_or_some.add_argument('-m', '--motif', type=str, required=False, help='classify every base of type --base in the motif specified instead (can be single one-mer)')
    parser.add_argument('-r', '--reference', type=str, required=True, help='fasta file with reference aligned to')
    parser.add_argument('-e', '--tsv', type=str, required=True, help='tsv file with nanopolish event alignment')
    parser.add_argument('-f', '--fastq', type=str, required=True, help='fastq file with nanopore reads')
    parser.add_argument('-t', '--threads', type=int, required=False, help='specify number of processes (default = 1)', default=1)
    parser.add_argument('-b', '--base', type=str, required=False, help='bases to classify as methylated or unmethylated (A or C, default A)', default='A')
    parser.add_argument('-n', '--num_variables', type=int, required=False, help='change the length of the context used to classify (default of 6 variables corresponds to 11-mer context (6*2-1))', default=6)
    parser.add_argument('--train', action='store_true', required=False, help='train a new model (requires labels in positions file)', default=False)
    parser.add_argument('--training_tsv', type=str, required=False, help='mCaller output file for training')
    parser.add_argument('-d', '--modelfile', type=str, required=False, help='model file name')
    parser.add_argument('-s', '--skip_thresh', type=int, required=False, help='number of skips to allow within an observation (default 0)', default=0)
    parser.add_argument('-q', '--qual_thresh', type=float, required=False, help='quality threshold for reads (default none)', default=0)
    parser.add_argument('-c', '--classifier', type=str, required=False, help='use alternative classifier: options = NN (default), RF, LR, or NBC (non-default may significantly increase runtime)', default='NN')
    parser.add_argument('--plot_training', action='store_true', required=False, help='plot probabilities distributions for training positions (requires labels in positions file and --train)', default=False)
    parser.add_argument('-v', '--version', action='version', help='print version', version='%(prog)s v1.0')
    parser.add_argument('--gpus', type=int, required=False, default=int(os.environ.get('MCALLER_GPUS', '1')),
                        help='(mcaller_amd) GPUs of this node to shard the reads over (default 1, or $MCALLER_GPUS)')
    parser.add_argument('--bed', action='store_true', required=False, default=False,
                        help='(mcaller_amd) also write <stem>.methylation.summary.bed from the per-site reduction of the calls '
                             '(device-side counts, ncclAllReduce over the GPUs): what make_bed.py -f <diffs> writes')
    parser.add_argument('--bed_vo', action='store_true', required=False, default=False,
                        help='(mcaller_amd) with --bed: append the per-read probabilities of every site, like make_bed.py --vo')
    parser.add_argument('--bed_min_depth', type=int, required=False, default=15, help='(mcaller_amd) make_bed -d for --bed')
    parser.add_argument('--bed_mod_threshold', type=float, required=False, default=0.5, help='(mcaller_amd) make_bed -t for --bed')
    args = parser.parse_args(argv)

    if args.base == 'A':
        mod = 'm6A'
    elif args.base == 'C':
        mod = 'm5C'
    else:
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
