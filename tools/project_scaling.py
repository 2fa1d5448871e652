"""What ONE worker of an N-GPU run costs, measured on one GPU: a PROJECTION of the strong-scaling curve before a node exists.

  python tools/project_scaling.py --inputs DIR [--parts 2,4,8] [--runs 3] [--json]
  python tools/project_scaling.py --rows N ...                                          (writes its own synthetic inputs first)
  python tools/project_scaling.py --inputs DIR --one N [--reader pread|mmap|mmap_keep] [--runs 3] [--json]     (what --parts runs per N)

`mCaller --gpus N` (mcaller_amd/multi_gpu.py; the reference's fan-out: mCaller.py:62-70) cuts the file's consumed byte range
at read starts into N pieces and starts one worker per GPU; a worker streams its piece with stream_features under
MCALLER_HOST_CORES = mc_host_cores() / N host threads, and the workers share nothing until the per-site reduction (0.6 MB,
a fraction of a millisecond).  So the wall time of an N-GPU run, up to that reduction and the join of the parts, is the time of
its slowest worker -- and a worker's time can be measured here: ONE worker, its 1/N piece of the file, 1/N of the host threads,
one GPU.  What this cannot see: N workers contending for the host's memory bandwidth and page cache, N PCIe links at once.
The projection is labelled as such wherever it is quoted."""
import contextlib, io, json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one_worker(inputs, n_parts, runs, reader):
    """This process IS the worker: piece (n_parts // 2) of n_parts, host threads = mc_host_cores() // n_parts (at least 2, as
    multi_gpu._worker sets them)."""
    from mcaller_amd import _lib
    import numpy as np
    L = _lib.lib()
    cores_all = int(L.mc_host_cores())
    share = max(2, cores_all // n_parts) if n_parts > 1 else cores_all
    os.environ['MCALLER_HOST_CORES'] = str(share)
    if reader != 'pread':
        os.environ['MCALLER_READER'] = reader           # mmap | mmap_keep (mcaller_amd._lib.TextBlock)
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.model_io import load_model_file, shipped_model
    from mcaller_amd.read_qual import extract_read_quality
    tsv, fasta, fastq = (os.path.join(inputs, f) for f in ('syn.eventalign.tsv', 'ref.fasta', 'reads.fastq'))
    size = os.path.getsize(tsv)
    lo, hi = _lib.eventalign_consumed_range(tsv, 0, size)
    cuts = _lib.eventalign_read_cuts(tsv, n_parts, lo, hi)
    piece = min(n_parts // 2, n_parts - 1)
    a, b = int(cuts[piece]), int(cuts[piece + 1])
    modelset = load_model_file(shipped_model('r95_twobase_model_NN_6_m6A'))
    r2q = extract_read_quality(fastq)
    out_path = os.path.join(inputs, 'projection.part')
    secs, rows, calls = [], 0, 0
    for _ in range(runs):
        t = time.perf_counter()
        with open(out_path, 'wb') as out, contextlib.redirect_stdout(io.StringIO()):
            res = ec.stream_features(tsv, fasta, r2q, 6, 0, 0.0, modelset, None, 'A', 'GATC', None, byte_range=(a, b), sink=out.write,
                                     tail_of_last=lambda: None, min_shards=1)
        secs.append(time.perf_counter() - t)
        rows, calls = res.n_rows, res.n_obs
    os.remove(out_path)
    warm = secs[1:] if len(secs) > 1 else secs
    ck = dict(getattr(ec.stream_features, 'last_clock', None) or {})
    ck.pop('events', None)
    return dict(n_parts=n_parts, piece=piece, text_bytes=b - a, rows=rows, calls=calls, host_threads=share, host_threads_of=cores_all,
                reader=reader, seconds_median=float(np.median(warm)), seconds_best=min(warm), seconds_all=secs,
                text_GBps=(b - a) / float(np.median(warm)) / 1e9, stream_last_run=ck)


def main():
    args = sys.argv[1:]
    as_json = '--json' in args
    if '--rows' in args:
        import tempfile
        from mcaller_amd import synth
        inputs = tempfile.mkdtemp(prefix='mc_proj_')
        codes = synth.genome()
        table, qual = synth.make_table(int(float(args[args.index('--rows') + 1])), seed=5, codes=codes)
        synth.write_inputs(table, qual, codes, inputs)
        del table
    else:
        inputs = args[args.index('--inputs') + 1]
    runs = int(args[args.index('--runs') + 1]) if '--runs' in args else 3
    reader = args[args.index('--reader') + 1] if '--reader' in args else 'pread'
    real_stdout = sys.stdout
    if as_json:
        sys.stdout.flush()
        real_stdout = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    if '--one' in args:
        res = one_worker(inputs, int(args[args.index('--one') + 1]), runs, reader)
    else:
        parts = [int(x) for x in (args[args.index('--parts') + 1] if '--parts' in args else '2,4,8').split(',')]
        per_n = []
        for n in parts:                              # (a process per N: the worker threads are sized once per process)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--inputs', inputs, '--one', str(n), '--runs', str(runs),
                                '--reader', reader, '--json'], capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                per_n.append(dict(n_parts=n, error=r.stderr[-400:]))
                continue
            per_n.append(json.loads(r.stdout.strip().splitlines()[-1]))
        res = dict(per_n=per_n, what=__doc__.split('\n\n')[1].replace('\n', ' '))
    real_stdout.write(json.dumps(res) + '\n')
    real_stdout.flush()


if __name__ == '__main__':
    main()
