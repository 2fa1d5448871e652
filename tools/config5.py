"""BASELINE.json configs[4] timed as ONE workload: `--train` on labelled positions + the RF classifier, 10^7 event rows.

  python tools/config5.py [rows] [--runs N] [--json] [--keep DIR]

What it measures, all through the product (the CLI, `mcaller_amd.mCaller.main`, and the C ABI behind it), on one GPU:
  train_file_to_file   `mCaller -p positions.txt --train -c NN`: eventalign text -> `.diffs.6.train` + the fitted model file.
                       Split: the feature-matrix build (extract_features(train=True): text streamed through the GPU in shards,
                       positions-mode scan + emit, the reference's train dicts built per record, extract_contexts.py:210-215,
                       302-303) and the fit (train_model.py:33-113: six runs of the MLP optimiser in one mc_mlp_fit call, k4_mlp_fit)
  predict_rf_file_to_file   `mCaller -p positions.txt -c RF -d <forest>`: text -> `.diffs.6` scored by k3_forest
  kernels              hipEvents around the stages of synchronous passes over the resident 10^7-row table in positions mode
                       (strand resolve, scan, ordering + emit) and around the classifier (k3_forest; k2_mlp beside it)
The forest is the committed fixture tests/golden/models/rf_twobase_model_RF_6_m6A.pkl (fitted by tests/golden/make_golden_train.py with
the reference's hyper-parameters, train_model.py:39-45).  bench.py runs this in a process of its own and adds the CPU leg."""
import contextlib, io, json, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np


def write_inputs(n_rows, d):
    """10^7-row synthetic eventalign file + FASTA + FASTQ + labelled positions: the A of every GATC on both strands, a third
    of them labelled m6A (what tests/test_gpu_config5.py scans)."""
    from mcaller_amd import synth
    codes = synth.genome()
    table, qual = synth.make_table(n_rows, seed=55, codes=codes)
    paths = synth.write_inputs(table, qual, codes, d)
    seq = np.frombuffer(synth.codes_to_str(codes).encode('ascii'), dtype=np.uint8)
    hit = np.flatnonzero((seq[:-3] == ord('G')) & (seq[1:-2] == ord('A')) & (seq[2:-1] == ord('T')) & (seq[3:] == ord('C')))
    paths['positions'] = os.path.join(d, 'positions.txt')
    with open(paths['positions'], 'w') as fh:
        for p in hit:
            fh.write('ecoli_syn\t%d\t+\t%s\n' % (p + 1, 'm6A' if (p + 1) % 3 == 0 else 'A'))
            fh.write('ecoli_syn\t%d\t-\t%s\n' % (p + 2, 'm6A' if (p + 2) % 3 == 0 else 'A'))
    return paths, len(hit) * 2


def main():
    args = sys.argv[1:]
    as_json = '--json' in args
    real_stdout = sys.stdout
    if as_json:
        sys.stdout.flush()
        real_stdout = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    runs = int(args[args.index('--runs') + 1]) if '--runs' in args else 3
    n_rows = int(float(args[0])) if args and not args[0].startswith('--') else 10000000
    keep = args[args.index('--keep') + 1] if '--keep' in args else None
    d = keep or tempfile.mkdtemp(prefix='mc_config5_')
    os.makedirs(d, exist_ok=True)
    t0 = time.perf_counter()
    paths, n_positions = write_inputs(n_rows, d)
    t_inputs = time.perf_counter() - t0
    os.sync()
    from mcaller_amd import mCaller, train_model
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.device import Device
    stem = paths['tsv'][:-4]
    rf_file = os.path.join(REPO, 'tests', 'golden', 'models', 'rf_twobase_model_RF_6_m6A.pkl')
    os.environ.setdefault('MCALLER_SEED', '7')              # (the fit's length depends on its seed: the same six fits every run)

    # (what the fit costs inside the CLI: train_classifier around the one mc_mlp_fit call)
    clock = {}
    real_train, real_fit = train_model.train_classifier, Device.mlp_fit

    def timed_train(*a, **kw):
        t = time.perf_counter()
        try:
            return real_train(*a, **kw)
        finally:
            clock['train_classifier'] = time.perf_counter() - t

    def timed_fit(self, X, y, jobs, **kw):
        t = time.perf_counter()
        try:
            fits = real_fit(self, X, y, jobs, **kw)
            clock['fit_rows'], clock['fit_jobs'] = int(len(y)), len(jobs)
            clock['fit_iters'] = [int(f['n_iter']) for f in fits]
            return fits
        finally:
            clock['mlp_fit'] = time.perf_counter() - t
    train_model.train_classifier, Device.mlp_fit = timed_train, timed_fit

    def run_cli(argv, out_file):
        for f in (out_file,):
            if os.path.exists(f):
                os.remove(f)
        clock.clear()
        buf = io.StringIO()
        t = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            mCaller.main(argv)
        dt = time.perf_counter() - t
        ck = dict(getattr(ec.stream_features, 'last_clock', {}))
        ck.pop('events', None)
        return dt, dict(clock), ck, buf.getvalue()

    common = ['-p', paths['positions'], '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq']]
    train_runs, rf_runs = [], []
    model_out = os.path.join(d, 'trained_model_NN_6_m6A.pkl')
    for _ in range(runs):
        dt, ck, stream, text = run_cli(common + ['--train', '-d', model_out], stem + '.diffs.6.train')
        n_train_rows = sum(1 for _ in open(stem + '.diffs.6.train', 'rb'))
        train_runs.append(dict(seconds=dt, feature_matrix_s=dt - ck.get('train_classifier', 0.0), train_classifier_s=ck.get('train_classifier'),
                               mlp_fit_s=ck.get('mlp_fit'), fit_rows=ck.get('fit_rows'), fit_iters=ck.get('fit_iters'),
                               stream=stream, rows_written=n_train_rows))
    for _ in range(runs):
        dt, ck, stream, text = run_cli(common + ['-c', 'RF', '-d', rf_file], stem + '.diffs.6')
        n_calls = sum(1 for _ in open(stem + '.diffs.6', 'rb'))
        rf_runs.append(dict(seconds=dt, stream=stream, calls=n_calls))

    # ---- the kernels, one pass at a time over the resident table (positions mode: marked sites are sparse, k1_scan<64>) ----
    from mcaller_amd.read_qual import extract_read_quality
    from mcaller_amd.model_io import load_model_file, shipped_model
    r2q = extract_read_quality(paths['fastq'])
    size = os.path.getsize(paths['tsv'])
    with contextlib.redirect_stdout(io.StringIO()):
        P = ec.prepare(paths['tsv'], paths['fasta'], r2q, 0, size, 'A', None, paths['positions'])
    dev = ec.get_device()
    dev.set_reference(P.ref.device_arrays())
    slot = dev.upload_table_async(P.table, P.qual)
    dev.wait_upload(slot)
    kernels = {}
    for name, modelfile in (('forest', rf_file), ('mlp', shipped_model('r95_twobase_model_NN_6_m6A'))):
        ms = load_model_file(modelfile)
        _, weights, _, soc = ec.submodel_setup(ms, 'A')
        dev.set_classifier(weights, soc)
        tms = []
        for i in range(7):
            dev.select_table(slot, as_new=True)
            dev.run(6, 0, 0.0, tail_contig=-1, score=True)
            rec = dev.fetch(copy=False)
            tms.append(dev.times_ms())
        med = {k: float(np.median([t[k] for t in tms[1:]])) for k in tms[0]}
        info = rec.info[:rec.n]
        scored = int(np.isfinite(rec.prob[:rec.n]).sum())
        kernels[name] = dict(ms=med, records=int(rec.n), scored=scored,
                             classifier_us=med['classifier'] * 1e3,
                             classifier_ns_per_scored_record=med['classifier'] * 1e6 / max(scored, 1))
    med_of = lambda rs, key: float(np.median([r[key] for r in rs[1:] or rs]))       # noqa: E731
    alg_bytes = 17.0 * n_rows + 64.0 * kernels['forest']['scored']
    k1_ms = kernels['forest']['ms']['strand_resolve'] + kernels['forest']['ms']['window_scan'] + kernels['forest']['ms']['emit']
    res = dict(rows=n_rows, tsv_bytes=size, labelled_positions=n_positions, inputs_written_s=t_inputs,
               train_file_to_file=dict(seconds_median=med_of(train_runs, 'seconds'), seconds_first_run=train_runs[0]['seconds'],
                                       feature_matrix_s=med_of(train_runs, 'feature_matrix_s'),
                                       train_classifier_s=med_of(train_runs, 'train_classifier_s'),
                                       mlp_fit_ms=med_of(train_runs, 'mlp_fit_s') * 1e3,
                                       mlp_fit_what='six fits (5 GroupKFold folds + the final one) of %s balanced rows in ONE mc_mlp_fit call '
                                                    '(k4_mlp_fit, four workgroups per fit); Adam iterations per fit: %s' % (
                                                        train_runs[-1]['fit_rows'], train_runs[-1]['fit_iters']),
                                       training_rows=train_runs[-1]['rows_written'], events_per_s=n_rows / med_of(train_runs, 'feature_matrix_s'),
                                       stream_last_run=train_runs[-1]['stream'], seconds_all=[r['seconds'] for r in train_runs]),
               predict_rf_file_to_file=dict(seconds_median=med_of(rf_runs, 'seconds'), seconds_first_run=rf_runs[0]['seconds'],
                                            calls=rf_runs[-1]['calls'], calls_per_s=rf_runs[-1]['calls'] / med_of(rf_runs, 'seconds'),
                                            events_per_s=n_rows / med_of(rf_runs, 'seconds'), text_GBps=size / med_of(rf_runs, 'seconds') / 1e9,
                                            stream_last_run=rf_runs[-1]['stream'], seconds_all=[r['seconds'] for r in rf_runs]),
               kernels=kernels,
               roofline_positions_mode=dict(algorithmic_bytes=alg_bytes, kernel_ms=k1_ms, achieved_GBps=alg_bytes / (k1_ms * 1e-3) / 1e9,
                                            frac=alg_bytes / (k1_ms * 1e-3) / 1e9 / 8000.0,
                                            what='k0_first_site + k1_scan<64, validating> + ordering + k1_emit over the 10^7-row table in '
                                                 'positions mode, hipEvents around synchronous passes (at 10^7 rows the kernels are '
                                                 'latency-bound: 160 MB in tens of microseconds)'),
               what='python tools/config5.py %d --runs %d: BASELINE.json configs[4] through the CLI on one GPU; medians over the warm runs' % (n_rows, runs))
    if keep:
        res['inputs'] = paths
    if as_json:
        real_stdout.write(json.dumps(res) + '\n')
        real_stdout.flush()
    else:
        print(json.dumps(res, indent=1))
    if not keep:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
