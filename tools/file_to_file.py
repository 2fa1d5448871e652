"""File-to-file timing (third timing): synthetic eventalign TSV -> .diffs.6 through the CLI.

  python tools/file_to_file.py [rows] [--runs N]            writes its own inputs (synthetic, `rows` rows)
  python tools/file_to_file.py --inputs DIR [--runs N]      inputs already written (synth.write_inputs: syn.eventalign.tsv, ...)
  --gpus N [--bed]   the sharded path (mcaller_amd/multi_gpu.py): N byte ranges cut at read starts, one worker process per GPU
                     streaming its range, with --bed the per-site reduction (ncclAllReduce) and the BED file -- BASELINE.json
                     configs[3].  The workers of one run stay for the next (MCALLER_KEEP_WORKERS): the first run pays for N
                     interpreters and HIP contexts, the later ones are what a file costs
  --motif M          the CLI's -m (default GATC; `A`: every A of both strands is a site -- the dense mode, BASELINE.md's per-phase profile)
  --json: one JSON line (bench.py's 10^8-row legs run this in a process of its own: wall time per run, peak RSS of the process,
          sha-256 of the output, and for the sharded path what multi_gpu.last_run measured in every run)"""
import contextlib, hashlib, io, json, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth, mCaller

def peak_rss_mb():
    """High-water mark of THIS program's resident memory (VmHWM: per address space -- ru_maxrss would carry over the size of the
    process that started us, across fork and exec)."""
    for line in open('/proc/self/status'):
        if line.startswith('VmHWM:'):
            return int(line.split()[1]) / 1024.0
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, 'rb') as fh:
        for block in iter(lambda: fh.read(16 << 20), b''):
            h.update(block)
    return h.hexdigest()


def main():
    # (a function behind a __main__ guard: the workers of a sharded run are spawned, and a spawned interpreter imports the
    # parent's main module)
    args = sys.argv[1:]
    as_json = '--json' in args
    real_stdout = sys.stdout
    if as_json:                   # the one JSON line is all that goes to stdout: RCCL's banner (workers, C stdio) goes to stderr
        sys.stdout.flush()
        real_stdout = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    runs = int(args[args.index('--runs') + 1]) if '--runs' in args else 6
    n_gpus = int(args[args.index('--gpus') + 1]) if '--gpus' in args else 0
    with_bed = '--bed' in args
    motif = args[args.index('--motif') + 1] if '--motif' in args else 'GATC'
    if '--inputs' in args:
        d = args[args.index('--inputs') + 1]
        paths = dict(tsv=os.path.join(d, 'syn.eventalign.tsv'), fasta=os.path.join(d, 'ref.fasta'), fastq=os.path.join(d, 'reads.fastq'))
        n_rows = None
    else:
        n_rows = int(float(args[0])) if args and not args[0].startswith('--') else 1000000
        d = tempfile.mkdtemp(prefix='mc_f2f_')
        codes = synth.genome()
        table, qual = synth.make_table(n_rows, seed=5, codes=codes)
        t = time.time()
        paths = synth.write_inputs(table, qual, codes, d)
        if not as_json:
            print('inputs written in %.1f s: %.1f MB of TSV, %d rows' % (time.time() - t, os.path.getsize(paths['tsv']) / 1e6, table.n_rows))
        del table
    if not as_json:
        os.environ['MCALLER_TIMING'] = os.environ.get('MCALLER_TIMING', '1')
    model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
    out = paths['tsv'][:-4] + '.diffs.6'
    bed = os.path.join(os.path.dirname(paths['tsv']), 'syn.methylation.summary.bed')
    argv = ['-m', motif, '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model]
    stats_path = None
    if n_gpus or with_bed:
        argv += ['--gpus', str(max(1, n_gpus))] + (['--bed', '--bed_min_depth', '1'] if with_bed else [])
        os.environ['MCALLER_KEEP_WORKERS'] = '1'
        stats_path = os.path.join(d, 'sharded_run_stats.json')
        os.environ['MCALLER_STATS_JSON'] = stats_path
    times, calls, stats_all, stdout_tail, phases = [], 0, [], '', []
    from mcaller_amd import extract_contexts as ec
    for rep in range(runs):
        for f in (out, bed, stats_path):
            if f and os.path.exists(f):
                os.remove(f)
        buf = io.StringIO()
        t = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            mCaller.main(argv)
        dt = time.perf_counter() - t
        times.append(dt)
        stdout_tail = buf.getvalue()[-600:]
        # what the one-GPU stream measured of itself (extract_contexts.stream_features: seconds of the main thread's phases)
        ck = dict(getattr(ec.stream_features, 'last_clock', None) or {})
        ck.pop('events', None)
        phases.append(ck if not (n_gpus or with_bed) else None)
        calls = sum(1 for _ in open(out, 'rb'))
        if stats_path:
            stats_all.append(json.load(open(stats_path)) if os.path.exists(stats_path) else None)
            if stats_all[-1] is None:                      # (the sharded path declined and the one-GPU path ran: the later runs would too)
                break
        if os.environ.get('MCALLER_RSS'):                      # (what the process's resident memory is made of, run after run)
            st = dict(l.split(':', 1) for l in open('/proc/self/status').read().splitlines() if ':' in l)
            sys.stderr.write('run %d: VmRSS %s RssAnon %s RssFile %s RssShmem %s VmHWM %s\n' % (
                rep, st['VmRSS'].strip(), st['RssAnon'].strip(), st['RssFile'].strip(), st['RssShmem'].strip(), st['VmHWM'].strip()))
        if not as_json:
            print('run %d: %.3f s wall, %d calls%s' % (rep, dt, calls, '' if n_rows is None else ' -> %.3g events/s, %.3g calls/s (file to file)'
                                                       % (n_rows / dt, calls / dt)))
            if stats_all and stats_all[-1]:
                s = stats_all[-1]
                print('       sharded: %s | workers %s | reduction %s' % (
                    {k: round(v, 3) for k, v in s['seconds'].items()},
                    [(w['rows'], round(w['seconds']['total'], 3), round(w['seconds']['setup'], 3), 'rss %.0f MB' % (w['peak_rss_mb'] or 0), {k: round(v) for k, v in (w.get('rss_mb') or {}).items() if v is not None}) for w in s['workers']],
                    s['site_reduction'] and {k: s['site_reduction'][k] for k in ('backend', 'ms', 'bytes')}))
    if as_json:
        res = {'seconds_all': times, 'calls': calls, 'tsv_bytes': os.path.getsize(paths['tsv']), 'diffs_bytes': os.path.getsize(out),
               'diffs_sha256': sha256_of(out), 'peak_rss_mb': peak_rss_mb(), 'argv': argv[8:], 'motif': motif, 'phases_all': phases}
        if stats_path:
            res['sharded_runs'] = stats_all
            res['stdout_tail'] = stdout_tail
        if with_bed and os.path.exists(bed):
            res['bed_rows'] = sum(1 for _ in open(bed, 'rb'))
            res['bed_sha256'] = sha256_of(bed)
        real_stdout.write(json.dumps(res) + '\n')
        real_stdout.flush()


if __name__ == '__main__':
    main()
