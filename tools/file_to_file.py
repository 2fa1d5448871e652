"""File-to-file timing (third timing): synthetic eventalign TSV -> .diffs.6 through the CLI.

  python tools/file_to_file.py [rows] [--runs N]            writes its own inputs (synthetic, `rows` rows)
  python tools/file_to_file.py --inputs DIR [--runs N]      inputs already written (synth.write_inputs: syn.eventalign.tsv, ...)
  --json: one JSON line (bench.py's 10^8-row leg runs this in a process of its own: wall time per run, peak RSS of the process)"""
import contextlib, io, json, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth, mCaller

def peak_rss_mb():
    """High-water mark of THIS program's resident memory (VmHWM: per address space -- ru_maxrss would carry over the size of the
    process that started us, across fork and exec)."""
    for line in open('/proc/self/status'):
        if line.startswith('VmHWM:'):
            return int(line.split()[1]) / 1024.0
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


args = sys.argv[1:]
as_json = '--json' in args
runs = int(args[args.index('--runs') + 1]) if '--runs' in args else 6
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
times, calls = [], 0
for rep in range(runs):
    if os.path.exists(out):
        os.remove(out)
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        mCaller.main(['-m', 'GATC', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model])
    dt = time.perf_counter() - t
    times.append(dt)
    calls = sum(1 for _ in open(out, 'rb'))
    if os.environ.get('MCALLER_RSS'):                      # (what the process's resident memory is made of, run after run)
        st = dict(l.split(':', 1) for l in open('/proc/self/status').read().splitlines() if ':' in l)
        sys.stderr.write('run %d: VmRSS %s RssAnon %s RssFile %s RssShmem %s VmHWM %s\n' % (
            rep, st['VmRSS'].strip(), st['RssAnon'].strip(), st['RssFile'].strip(), st['RssShmem'].strip(), st['VmHWM'].strip()))
    if not as_json:
        print('run %d: %.3f s wall, %d calls%s' % (rep, dt, calls, '' if n_rows is None else ' -> %.3g events/s, %.3g calls/s (file to file)'
                                                   % (n_rows / dt, calls / dt)))
if as_json:
    print(json.dumps({'seconds_all': times, 'calls': calls, 'tsv_bytes': os.path.getsize(paths['tsv']), 'diffs_bytes': os.path.getsize(out),
                      'peak_rss_mb': peak_rss_mb()}))
