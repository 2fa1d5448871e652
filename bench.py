#!/usr/bin/env python3
"""bench.py -- m6A calls/sec of the hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic eventalign rows that is already resident in HBM, doing
EVERYTHING a table costs when it is scanned once, as every table of a file is: strand resolve, the scan that streams the
position and event-index columns and validates every row, record ordering, window emit, MLP classifier, packing and the D2H
copy of the flush records.  Two distinct tables are resident and taken in turn; before each step the table is declared new
(mc_ctx_select_table(as_new)), so no step uses what an earlier pass over the same rows learned (validation flags, unit
summaries).  The rate of repeated passes over one validated table is reported beside it as config.resident_rescan.
Workload at N=1: BASELINE.json configs[2] -- synthetic 10^8 events, -m GATC, NN classifier (r95 two-base MLP),
skip_thresh 0.  N>1: every rank scans its own 10^8-row shards of reads (weak scaling, no data-path collective).
`python bench.py --gpus N` without WORLD_SIZE in the environment starts the N ranks itself.

After the timed steps the ranks > 0 are done; rank 0 goes on alone (at every N) with the product-shaped legs, among them
  strong_scaling             BASELINE.json configs[3] through the product path: ONE 10^8-row eventalign file ->
                             `python -m mcaller_amd.mCaller --gpus N --bed` (mcaller_amd/multi_gpu.py: N byte ranges cut at read
                             starts, one worker process per GPU streaming its range over its own PCIe link, the per-site
                             ncclAllReduce, .diffs.6 and BED written), wall time of the whole run; "scaling": "strong"

Prints ONE JSON line on rank 0.  Beside the contract's keys it carries:
  config.device_e2e          distinct shards (10^8 rows in total) streamed from pinned host memory through the table slots:
                             H2D + per-table kernel + pass + D2H of the records, next to the measured H2D-only rate
  config.file_to_file        eventalign TSV -> .diffs.6 through the CLI (parser and row formatter included)
  config.per_table_kernel_ms every kernel that touches a table once, hipEvent times, one pass at a time
  roofline                   `achieved` = SURVEY.md 8(d)'s algorithmic bytes (17 B/row + 64 B/call) / the summed live hipEvent
                             times of ALL kernels that touch the table once (strand resolve, validating scan, ordering,
                             emit); `frac` = that / 8 TB/s; `traffic` = the HBM bytes those kernels move (rocprofv3 PMC
                             passes, profiles/), quoted only if the PMC file was collected on the kernel sources that are
                             running (hash recorded in the file), else null; `scan` = the same for the dominant kernel alone
  cpu_baseline               the C oracle on one host core; cpu_baseline_all_cores: the same on all cores;
                             cpu_baseline_reference_like: the Python twin (oracle/py_oracle.py, one predict_proba-equivalent
                             per observation) under multiprocessing on all host cores, byte-range fan-out like the
                             reference's -t (mCaller.py:62-70), CPU model stated
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_FILE = os.path.join(REPO, 'profiles', 'r06_pmc.json')
KERNEL_SOURCES = ['mcaller_amd/csrc/mc_dev.h', 'mcaller_amd/csrc/mc_rows.h', 'mcaller_amd/csrc/mc_rowtext.h', 'mcaller_amd/csrc/mc_rowtext.hip', 'mcaller_amd/csrc/mc_k0.hip', 'mcaller_amd/csrc/mc_scan.hip', 'mcaller_amd/csrc/mc_emit.hip', 'mcaller_amd/csrc/mc_fused.hip', 'mcaller_amd/csrc/mc_literal.hip',
                  'mcaller_amd/csrc/mc_classify.hip', 'mcaller_amd/csrc/mc_stream.hip', 'mcaller_amd/csrc/mc_devparse.inc']


def kernel_source_hash():
    """sha256 over the kernel sources the library was built from (tools/pmc_summary.py records the same in the PMC file)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(REPO, rel), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def dist_setup(n_gpus):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod      # plumbing only: rendezvous, barrier, max-over-ranks
        # (gloo announces its connections on STDOUT: the one JSON line is the only thing this program writes there)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist_mod.init_process_group(backend='gloo', init_method='env://')
            dist_mod.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        dist = dist_mod
    return rank, world, local, dist


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` with no launcher around it: this process starts the N ranks (fresh interpreters, one per GPU;
    it never touches a GPU itself), hands them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like torch.distributed.run would, and
    waits.  Rank 0 prints the JSON line.  A rank that dies takes the others down and the exit code with it."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        env['MCALLER_BENCH_SPAWNED'] = '1'
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        left = list(procs)
        while left:
            for pr in list(left):
                code = pr.poll()
                if code is None:
                    continue
                left.remove(pr)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write('bench.py: rank %d exited with code %d; stopping the others\n' % (procs.index(pr), code))
                    for other in left:
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


# ---- the Python twin under multiprocessing (cpu_baseline_reference_like); workers import the oracle only ----
_twin = {}


def _twin_init(fastq, model_npz):
    sys.path.insert(0, REPO)
    from oracle import py_oracle
    z = np.load(model_npz)
    keys = sorted(set(n.split('.')[0] for n in z.files if not n.startswith('__')))
    models = {k: (z[k + '.W1'], z[k + '.b1'], z[k + '.W2'], z[k + '.b2']) for k in keys}
    models['__twobase__'] = True
    _twin['models'] = models
    _twin['r2q'] = py_oracle.read_fastq_quality(fastq)
    _twin['fn'] = py_oracle.extract_features_oracle


def _twin_ready(_):
    time.sleep(0.05)
    return os.getpid()


def _twin_job(job):
    tsv, fasta, lo, hi, motif = job
    res = _twin['fn'](tsv, fasta, _twin['r2q'], 6, 0, 0.0, _twin['models'], lo, hi, base='A', motif=motif)
    return len(res['rows'])


def python_twin_baseline(paths, model_npz, n_rows_file, motif='GATC', fraction=1.0):
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    size = os.path.getsize(paths['tsv'])
    hc = host_cores_info()
    cores = hc['effective']                               # one process per core this container may USE (256 in the affinity mask, a quota of 16)
    sample = min(size, cores * (64 << 20))                # ~1.3 s per process at the twin's ~4e5 rows/s: 10-30 s of CPU work in all
    sample = max(1 << 20, int(sample * fraction))       # (a one-base motif: a call every eleven rows, one MLP forward each -- a bounded sample)
    n_jobs = cores
    step = sample // n_jobs
    jobs = [(paths['tsv'], paths['fasta'], i * step, (i + 1) * step if i + 1 < n_jobs else sample, motif) for i in range(n_jobs)]
    ctx = multiprocessing.get_context('spawn')           # never fork a process that holds a HIP context
    # (an executor, not a Pool: a worker that dies in its initializer breaks it with an exception instead of being respawned for ever)
    ex = ProcessPoolExecutor(max_workers=cores, mp_context=ctx, initializer=_twin_init, initargs=(paths['fastq'], model_npz))
    try:
        for f in [ex.submit(_twin_ready, i) for i in range(cores * 2)]:     # every worker up, models and qualities loaded
            f.result(timeout=600)
        t0 = time.perf_counter()
        calls = sum(f.result(timeout=900) for f in [ex.submit(_twin_job, j) for j in jobs])
        dt = time.perf_counter() - t0
    finally:
        procs = list((getattr(ex, '_processes', None) or {}).values())
        ex.shutdown(wait=False, cancel_futures=True)
        for p in procs:
            if p.is_alive():
                p.terminate()
    rows = n_rows_file * sample / float(size)
    return {'value': calls / dt, 'unit': 'calls/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
            'cpu_quota': hc['cpu_quota'], 'affinity': hc['affinity'], 'events_per_s': rows / dt,
            'sample': 'Python twin of extract_features -m %s (oracle/py_oracle.py), %d processes over byte ranges of %.0f MB of '
                      'eventalign text (~%.3g rows), %.2f s' % (motif, n_jobs, sample / 1e6, rows, dt)}


def host_cores_info():
    """What the CPU legs may use: the affinity mask says 256 on the GPU box, the container's CPU-time quota says 16
    (mc_host_cores() = min(affinity, cgroup quota, $MCALLER_HOST_CORES) -- what the product's own host threads are sized by)."""
    affinity = len(os.sched_getaffinity(0))
    try:
        from mcaller_amd import _lib
        effective = int(_lib.lib().mc_host_cores())
    except Exception:                                            # noqa
        effective = affinity
    return {'affinity': affinity, 'cpu_quota': effective if effective < affinity else None, 'effective': effective}


def _r(x, digits=5):
    """Numbers of the contract line with `digits` significant digits (the details file keeps them whole)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float('%.*g' % (digits, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(d, *keys):
    d = d or {}
    if 'error' in d:
        return {'error': str(d['error'])[:160]}
    return {k: d.get(k) for k in keys if d.get(k) is not None}


def _cpu_leg(d, sample_chars=150):
    """A CPU leg in the line: the contract's five keys + what the core count means."""
    d = d or {}
    if 'error' in d:
        return {'error': str(d['error'])[:160]}
    if not d:
        return None
    o = _pick(d, 'value', 'unit', 'cores', 'kind', 'cpu_quota', 'affinity', 'events_per_s', 'cpu_model')
    o['sample'] = str(d.get('sample', ''))[:sample_chars]
    return o


LINE_LIMIT = 6000          # characters: the driver keeps the tail of stdout; round 5's 20.8 KB line went unparsed


def contract_line(full):
    """The ONE line of the bench contract, from the full result: the contract's keys, `roofline`, `cpu_baseline`, and one
    number per side leg.  Everything else (phase splits, all runs' seconds, projections per N, prose) is in the details file
    (`details`), not here."""
    cfg, roof = full.get('config') or {}, full.get('roofline') or {}
    strong, dense = full.get('strong_scaling') or {}, full.get('file_to_file_dense') or {}
    c5, f2f = full.get('config5') or {}, cfg.get('file_to_file_1e8') or {}
    line = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_steady',
                                     'ms_per_step_fp64_mlp', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    line['config'] = dict(_pick(cfg, 'workload', 'step', 'passes_in_flight', 'events_per_gpu', 'calls_per_gpu', 'flush_records_per_gpu',
                                'copy_out_bytes_per_pass', 'events_per_s', 'algorithmic_GBps_of_the_step', 'calls_per_s_kernels_only',
                                'device_e2e_events_per_s'),
                          file_to_file_1e8_s=f2f.get('seconds_median'), file_to_file_1e8_text_GBps=f2f.get('text_GBps'),
                          resident_rescan_ms=(cfg.get('resident_rescan') or {}).get('ms_per_pass'))
    line['roofline'] = dict(_pick(roof, 'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes', 'kernel_ms',
                                  'kernels_ms'),
                            pipelined_frac=(roof.get('pipelined') or {}).get('frac'),
                            scan_ms=(roof.get('scan') or {}).get('kernel_ms'),
                            scan_streamed_frac=(roof.get('scan') or {}).get('streamed_frac'))
    line['roofline'].setdefault('traffic', None)
    line['cpu_baseline'] = _cpu_leg(full.get('cpu_baseline'))
    line['cpu_baseline_all_cores'] = _cpu_leg(full.get('cpu_baseline_all_cores'), 80)
    line['cpu_baseline_reference_like'] = _cpu_leg(full.get('cpu_baseline_reference_like'), 120)
    for k in ('file_to_file_calls_per_s', 'file_to_file_vs_cpu_reference_like', 'device_e2e_calls_per_s', 'value_vs_cpu_all_cores',
              'strong_value'):
        if full.get(k) is not None:
            line[k] = full[k]
    big = dense.get('big') or dense.get('1e7') or {}
    if dense:
        rows = float(big.get('rows') or 0)
        line['file_to_file_dense'] = {'error': str(big['error'])[:160]} if 'error' in big else dict(
            s_per_1e8=(big['seconds_median'] * 1e8 / rows) if rows and big.get('seconds_median') else None, rows=big.get('rows'),
            calls_per_s=big.get('calls_per_s'), bound=(big.get('bound') or '').split(' (')[0] or None,
            cpu_reference_like_calls_per_s=(dense.get('cpu_baseline_reference_like') or {}).get('value'))
    if full.get('roofline_fused_dense'):
        line['roofline_fused_dense'] = _pick(full['roofline_fused_dense'], 'frac', 'kernel_ms', 'achieved', 'algorithmic_bytes', 'the_pair_frac')
    if full.get('roofline_parser'):
        line['roofline_parser'] = _pick(full['roofline_parser'], 'frac', 'kernel_ms', 'text_h2d_ms', 'algorithmic_bytes')
    if c5:
        line['config5'] = {'error': str(c5['error'])[:160]} if 'error' in c5 else dict(
            rows=c5.get('rows'), train_s=(c5.get('train_file_to_file') or {}).get('seconds_median'),
            fit_s=(c5.get('train_file_to_file') or {}).get('train_classifier_s'),
            rf_predict_s=(c5.get('predict_rf_file_to_file') or {}).get('seconds_median'),
            cpu_one_core_calls_per_s=((c5.get('cpu_baseline') or {}).get('one_core') or {}).get('calls_per_s'))
    if strong:
        proj = (strong.get('projected') or {}).get('per_n') or []
        line['strong_scaling'] = dict(_pick(strong, 'scaling', 'n_gpus', 'rows', 'calls', 'seconds_median', 'seconds_first_run', 'calls_per_s',
                                            'events_per_s', 'diffs_equal_the_one_gpu_run', 'bed_rows'),
                                      site_reduction_ms=(strong.get('site_reduction') or {}).get('ms'),
                                      site_reduction_backend=str((strong.get('site_reduction') or {}).get('backend') or '')[:60] or None,
                                      projected_s={str(w['n_gpus']): w.get('seconds') for w in proj if not w.get('measured')} or None,
                                      error=str(strong['error'])[:160] if strong.get('error') else None)
    red = cfg.get('site_reduction')
    if red:
        line['site_reduction'] = _pick(red, 'backend', 'ms', 'observations', 'observations_expected', 'bytes', 'error')
    line['details'] = full.get('details')
    line = _r(line)
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:                                   # (never again: shed the optional legs before the contract's keys)
        for k in ('site_reduction', 'config5', 'roofline_parser', 'roofline_fused_dense', 'file_to_file_dense', 'strong_scaling',
                  'cpu_baseline_reference_like', 'cpu_baseline_all_cores'):
            line.pop(k, None)
            if len(json.dumps(line)) <= LINE_LIMIT:
                break
    return line


def write_details(full):
    """The whole result (every leg's phases, runs, prose) to a file beside the run and, indented, to stderr."""
    path = os.environ.get('MCALLER_BENCH_DETAILS') or os.path.join(REPO, 'gpurun_out', 'bench_details.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as fh:
            json.dump(full, fh, indent=1)
            fh.write('\n')
        full['details'] = os.path.relpath(path, REPO)
    except OSError as e:
        full['details'] = 'stderr only (%s)' % e
    sys.stderr.write('---- bench.py details (also in %s) ----\n%s\n' % (full['details'], json.dumps(full, indent=1)))
    sys.stderr.flush()


REAL_STDOUT = None


def print_line(obj):
    out = REAL_STDOUT if REAL_STDOUT is not None else sys.stdout
    out.write(json.dumps(obj) + '\n')
    out.flush()


METRIC = 'm6A calls/sec (GATC motif, E. coli-like synthetic eventalign)'
# what the timed step computes in: K1 (the roofline kernel) integer columns -> fp64 window means; K2, the default classifier, is
# the FAST forward (mc_classify.hip: hidden layer in fp32, output in fp64, every record near a print boundary again in fp64)
DTYPE = ('f64 (K1 window means, everything written); K2 hidden layer f32 + f64 output + f64 re-evaluation near print boundaries '
         '(MCALLER_MLP_FP64=1 = f64 throughout: ms_per_step_fp64_mlp)')
STRONG_KEYS = ('scaling', 'n_gpus', 'projected', 'rows', 'tsv_bytes', 'calls', 'calls_per_s', 'events_per_s', 'text_GBps', 'seconds_median',
               'seconds_best', 'seconds_first_run', 'seconds_all', 'site_reduction', 'workers', 'phases_s', 'bed_rows',
               'diffs_equal_the_one_gpu_run', 'peak_rss_mb', 'what')


def strong_scaling_leg(inputs_dir, n_gpus, dry=False, rows=None, one_gpu_sha=None, one_device=False, runs=4):
    """BASELINE.json configs[3] as it is stated, through the product: one eventalign file of `rows` rows ->
    `mCaller --gpus N --bed` (tools/file_to_file.py runs the CLI `runs` times in a process of its own; the workers of the first
    run stay for the later ones).  Total work is fixed as N grows: "scaling": "strong".  The timed span of a run is the CLI's
    main(): FASTQ qualities, FASTA, the cut into N byte ranges, N workers streaming (read, H2D, parse, passes, rows), the
    per-site reduction (ncclAllReduce when N > 1 communicators come up, the host's sum otherwise), parts joined, BED written."""
    out = dict.fromkeys(STRONG_KEYS)
    out['scaling'], out['n_gpus'] = 'strong', n_gpus
    if dry:
        return out
    import subprocess
    env = dict(os.environ, MCALLER_KEEP_WORKERS='1')
    env.setdefault('MCALLER_COMM_TIMEOUT', '60')        # (a communicator that does not come up: the parent sums on the host after this)
    env.setdefault('MCALLER_WORKER_TIMEOUT', '300')
    if one_device:
        env['MCALLER_SHARD_DEVICES'] = ','.join(['0'] * n_gpus)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'file_to_file.py'), '--inputs', inputs_dir, '--runs', str(runs),
                        '--json', '--gpus', str(n_gpus), '--bed'], capture_output=True, text=True, timeout=1200, env=env)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-800:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    secs = res['seconds_all']
    warm = secs[1:] if len(secs) > 1 else secs
    med = float(np.median(warm))
    stats = [x for x in res.get('sharded_runs', []) if x]
    last = stats[-1] if stats else None
    if last is None:
        raise RuntimeError('the sharded path declined the file (one-GPU fall-back ran): %s | stderr: %s'
                           % (res.get('stdout_tail', '')[-300:], r.stderr[-600:]))
    red = dict(last['site_reduction'] or {})
    if len(stats) > 1:           # (the collective's milliseconds: median over the warm runs)
        ms_all = [x['site_reduction']['ms'] for x in stats[1:] if x['site_reduction'] and x['site_reduction'].get('ms') is not None]
        red['ms_all_runs'] = [x['site_reduction'].get('ms') if x['site_reduction'] else None for x in stats]
        if ms_all:
            red['ms'] = float(np.median(ms_all))
    out.update(rows=rows if rows is not None else last['rows'], tsv_bytes=res['tsv_bytes'], calls=res['calls'],
               calls_per_s=res['calls'] / med, events_per_s=last['rows'] / med, text_GBps=res['tsv_bytes'] / med / 1e9,
               seconds_median=med, seconds_best=min(warm), seconds_first_run=secs[0], seconds_all=secs, site_reduction=red,
               workers=[dict(rank=w['rank'], device=w['device'], rows=w['rows'], text_bytes=w['text_bytes'], shards=w['shards'],
                             seconds=w['seconds']['total'], seconds_setup=w['seconds']['setup'],
                             seconds_site_counts=w['seconds']['site_counts'], peak_rss_mb=w['peak_rss_mb']) for w in last['workers']],
               phases_s=last['seconds'], bed_rows=res.get('bed_rows'),
               diffs_equal_the_one_gpu_run=(res['diffs_sha256'] == one_gpu_sha) if one_gpu_sha else None,
               peak_rss_mb=res['peak_rss_mb'],
               what='python tools/file_to_file.py --inputs ... --runs %d --gpus %d --bed: the CLI (mcaller_amd.mCaller --gpus N --bed) '
                    '%d times in one process of its own, page cache warm; seconds_first_run includes starting the %d worker '
                    'processes (interpreter, HIP context, pinned buffers), the later runs find them waiting; median / best over the '
                    'later runs; workers / phases_s / site_reduction: the last run' % (runs, n_gpus, runs, n_gpus))
    return out


def dense_file_to_file_leg(inputs_dir, rows, runs):
    """`mCaller -m A` (every A of both strands is a site: a call every eleven rows -- the mode BASELINE.md's per-phase profile of the
    reference is quoted on, extract_contexts.py:186,199,207,216) file to file through the CLI on one GPU, in a process of its own,
    with the phase split the stream measures of itself."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'file_to_file.py'), '--inputs', inputs_dir, '--runs', str(runs),
                        '--json', '--motif', 'A'], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-500:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    secs = res['seconds_all']
    warm = secs[1:] if len(secs) > 1 else secs
    med = float(np.median(warm))
    ph = [p for p in res.get('phases_all', [])[1:] if p] or [p for p in res.get('phases_all', []) if p]
    phases = {k: float(np.median([p[k] for p in ph])) for k in ('parse', 'wait_parser', 'enqueue', 'hand_out', 'wait_records', 'wait_formatter', 'format', 'write')
              if ph and all(k in p for p in ph)}
    fmt_threads = max(1, int(ph[-1].get('format_threads', 1) or 1)) if ph else 1
    dev_rows = int(ph[-1].get('device_rows', 0) or 0) if ph else 0         # shards whose rows the GPU wrote (mc_rowtext.hip)
    n_shards = int(ph[-1].get('shards', 0) or 0) if ph else 0
    bound = None
    if phases:
        main = {'helper threads (rows the host formatted -- mc_format_diffs -- and the counters; per helper thread)': phases.get('format', 0.0) / fmt_threads,
                'write (the rows appended to the output file)': phases.get('write', 0.0),
                'GPU + copy-out behind the link (records waited for)': phases.get('wait_records', 0.0),
                'reader + link + device parser (next table waited for)': phases.get('wait_parser', 0.0)}
        bound = max(main, key=main.get)
    return {'rows': rows, 'motif': 'A', 'tsv_bytes': res['tsv_bytes'], 'diffs_bytes': res['diffs_bytes'], 'calls': res['calls'],
            'seconds_first_run': secs[0], 'seconds_median': med, 'seconds_best': min(warm), 'seconds_all': secs,
            'events_per_s': rows / med, 'calls_per_s': res['calls'] / med, 'text_in_GBps': res['tsv_bytes'] / med / 1e9,
            'text_out_GBps': res['diffs_bytes'] / med / 1e9, 'peak_rss_mb': res['peak_rss_mb'],
            'phases_s': phases, 'bound': bound,
            'format_threads': fmt_threads, 'shards': n_shards, 'shards_with_rows_written_on_the_gpu': dev_rows,
            'phases_what': 'seconds per run, median of the warm runs.  Of the MAIN thread (they add up to the run; the reader threads and '
                           'the GPU work beside it): wait_parser = the next shard\'s table waited for (read, H2D, device parser), enqueue '
                           '= upload + passes enqueued, hand_out = wait_records (kernels + copy-out of the oldest pass) + wait_formatter '
                           '(the helper two shards back).  Of the helper threads: format = what is left of a shard for the host -- the '
                           'counters, and the rows of the shards the GPU did not write (native row formatter on all host cores); the sum '
                           'over format_threads helpers --, write = rows appended to the file',
            'what': 'python tools/file_to_file.py --inputs ... --motif A --runs %d --json: the CLI in a process of its own, page cache warm' % runs}


def config5_leg(keep_dir, rows):
    """BASELINE.json configs[4] -- `--train` on labelled positions + the RF classifier, 10^7 rows -- through the CLI on one GPU
    (tools/config5.py, a process of its own; the inputs stay in keep_dir for the CPU leg)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'config5.py'), str(int(rows)), '--runs', '3', '--json', '--keep', keep_dir],
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-600:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def config5_cpu_leg(paths):
    """The CPU leg of config 5: the C oracle's window machine + forest forward over the same 10^7-row table, on one core and
    sharded by read over all cores (the checker, timed as the baseline -- never part of the product path)."""
    import contextlib
    import io
    from concurrent.futures import ThreadPoolExecutor
    from tests import helpers as H
    from tests import shard
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    r2q = extract_read_quality(paths['fastq'])
    with contextlib.redirect_stdout(io.StringIO()):
        P = ec.prepare(paths['tsv'], paths['fasta'], r2q, 0, os.path.getsize(paths['tsv']), 'A', None, paths['positions'])
    arrays = P.ref.device_arrays()
    ms = H.load_rf_modelset()
    _, forests, _, soc = ec.submodel_setup(ms, 'A')
    t1 = time.perf_counter()
    orc = H.oracle_records(P.table, arrays, P.qual, 6, 0, 0.0)
    t2 = time.perf_counter()
    H.oracle_score(orc, P.table, P.qual, forests, soc, 6)
    t3 = time.perf_counter()
    scored = int(np.isfinite(orc.prob[:orc.n]).sum())
    hc = host_cores_info()
    cores = hc['effective']
    bounds = [b for b in shard.shard_bounds(P.table, cores) if b[1] > b[0]]
    subs = [(P.table.slice_segments(lo, hi), shard.tail_contig(P.table, P.qual, 0.0, hi)) for lo, hi in bounds]

    def one(job):
        st, tail = job
        o = H.oracle_records(st, arrays, P.qual, 6, 0, 0.0, tail_contig=tail)
        H.oracle_score(o, st, P.qual, forests, soc, 6)
        return int(np.isfinite(o.prob[:o.n]).sum())
    t4 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(subs)) as ex:
        mt = sum(ex.map(one, subs))
    t5 = time.perf_counter()
    return {'kind': 'port', 'cpu_model': cpu_model(), 'rows': int(P.table.n_rows), 'scored_records': scored,
            'one_core': {'window_machine_s': t2 - t1, 'forest_forward_s': t3 - t2, 'forest_ns_per_record': (t3 - t2) * 1e9 / max(scored, 1),
                         'calls_per_s': scored / (t3 - t1), 'events_per_s': P.table.n_rows / (t3 - t1)},
            'all_cores': {'threads': len(subs), 'cpu_quota': hc['cpu_quota'], 'affinity': hc['affinity'], 'seconds': t5 - t4,
                          'scored_records': mt, 'calls_per_s': mt / (t5 - t4),
                          'events_per_s': P.table.n_rows / (t5 - t4)},
            'sample': 'the parsed 10^7-row table of the config-5 file (columns in host memory): C oracle window machine + forest '
                      'forward (50 trees), one core %.2f s; sharded by read over %d threads %.2f s' % (t3 - t1, len(subs), t5 - t4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)     # 60 ms of timed work: the pipeline's fill and drain (one pass) amortise
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--events', type=float, default=1e8, help='event rows per GPU')
    ap.add_argument('--motif', default='GATC')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='one pass at a time (mc_extract_features) instead of the pipelined passes')
    ap.add_argument('--depth', type=int, default=4, help='pipelined passes in flight (1: one pass at a time through the pipelined interface; at most 5)')
    ap.add_argument('--time-every', type=int, default=8,
                    help='pipelined passes: hipEvents that time a pass go with every n-th pass (one costs the queue ~9 us)')
    ap.add_argument('--cpu-events', type=float, default=1e8, help='rows of the same workload timed on the CPU oracle')
    ap.add_argument('--stream-shards', type=int, default=10, help='device end-to-end: distinct shards the rows arrive in (0: skip)')
    ap.add_argument('--f2f-events', type=float, default=1e7, help='file to file: rows of eventalign text (0: skip)')
    ap.add_argument('--f2f-big-events', type=float, default=1e8,
                    help='file to file at the headline size, in a process of its own: rows of eventalign text (0: skip)')
    ap.add_argument('--strong-events', type=float, default=1e8,
                    help='strong-scaling leg (one file through mCaller --gpus N --bed): rows of eventalign text (0: skip)')
    ap.add_argument('--dense-events', type=float, default=1e8,
                    help='file to file with -m A (dense mode), at the 10^7-row and at this size: rows of eventalign text (0: skip both)')
    ap.add_argument('--config5-events', type=float, default=1e7,
                    help='BASELINE.json configs[4] (--train on labelled positions + RF): rows of eventalign text (0: skip)')
    ap.add_argument('--kernels-only', action='store_true', help='skip device end-to-end, file to file and the CPU legs (profiling runs)')
    ap.add_argument('--rescan-only', action='store_true',
                    help='profiling runs: the timed steps re-scan ONE validated table (config.resident_rescan) instead of full passes')
    ap.add_argument('--dry-ranks', action='store_true',
                    help='launch plumbing only (CPU test): the ranks rendezvous, rank 0 prints n_gpus, nothing touches a GPU')
    args = ap.parse_args()
    # The one JSON line is the only thing this program writes to its stdout: from here on file descriptor 1 is stderr (RCCL and
    # gloo announce themselves on the C library's stdout, buffered until exit), the line goes to a private copy of the real one.
    global REAL_STDOUT
    if REAL_STDOUT is None and not (args.gpus > 1 and 'WORLD_SIZE' not in os.environ):
        sys.stdout.flush()
        REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    if args.kernels_only:
        args.stream_shards, args.f2f_events, args.f2f_big_events, args.strong_events, args.no_cpu_baseline = 0, 0, 0, 0, True
        args.dense_events = args.config5_events = 0
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank, world, local, dist = dist_setup(args.gpus)
    if args.dry_ranks:
        if os.environ.get('MCALLER_BENCH_FAIL_RANK') == str(rank) and world > 1:      # (tests: a rank that dies)
            sys.exit(3)
        total = world
        if dist is not None:
            import torch
            t = torch.tensor([1.0], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            total = int(t[0])
            dist.destroy_process_group()
        if rank == 0:
            # (the real line's schema through the real line's compaction, values empty: what tests/test_shards.py checks the
            # contract on without a GPU)
            full = {'metric': METRIC, 'value': None, 'unit': 'calls/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                    'ms_per_step': None, 'ms_per_step_steady': None, 'ms_per_step_fp64_mlp': None, 'higher_is_better': True,
                    'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
                    'config': {'workload': 'dry run: the ranks rendezvous, nothing touches a GPU'},
                    'roofline': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'},
                    'cpu_baseline': dict(host_cores_info(), value=None, unit='calls/s', cores=1, kind='port', sample='dry run'),
                    'strong_scaling': strong_scaling_leg(None, world, dry=True)}
            line = contract_line(full)
            line.update(dry_ranks=True, ranks_seen=total, launcher='bench.py itself' if os.environ.get('MCALLER_BENCH_SPAWNED') else 'environment')
            print_line(line)
        return
    from mcaller_amd import synth, _lib
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model

    n_rows = int(args.events)
    t_gen = time.time()
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif=args.motif)
    # two distinct tables per rank, resident side by side: consecutive steps never touch the same rows
    n_tables = 1 if args.rescan_only else 2
    tables = [synth.make_table(n_rows, seed=1000 + rank + 500 * i, codes=codes) for i in range(n_tables)]
    table, qual = tables[0]
    t_gen = (time.time() - t_gen) / n_tables
    model_npz = shipped_model('r95_twobase_model_NN_6_m6A')
    modelset = load_model_file(model_npz)
    _, weights, _, soc = submodel_setup(modelset, 'A')

    dev_index = 0 if os.environ.get('MCALLER_BENCH_ONE_DEVICE') else local      # (one-GPU boxes: test the N>1 plumbing)
    affinity_at_start = os.sched_getaffinity(0)
    numa_node = Device.bind_host_to_numa_node(dev_index) if (world > 1 or os.environ.get('MCALLER_BENCH_BIND')) else None  # pinned buffers next to the rank's GPU
    dev = Device(dev_index)
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    t_up = time.time()
    slots = []
    for t_i, q_i in tables:                                      # (the read qualities travel with their table)
        slots.append(dev.upload_table_async(t_i.pinned() if world == 1 else t_i, q_i))
        dev.wait_upload(slots[-1])
    dev.sync()
    t_up = (time.time() - t_up) / n_tables

    # A step = one FULL pass of the hot path over a resident table -- what a table costs when it is scanned once: the table is
    # declared new (nothing an earlier pass learned is used), the scan streams positions and event indices and validates every
    # row; records (slot means, sites, probabilities) land in pinned host memory.  The two tables are taken in turn.  Passes
    # are pipelined (the library's streaming interface, mc_extract_features_async / mc_wait_records): strand resolve, scan and
    # ordering of consecutive passes back to back on one stream; the emit, K2 + packing on a side stream beside the next pass's
    # scan; copy-outs back to back on a third; --depth passes kept in flight (default four, the library allows six); every
    # pass's records are complete in host memory before the timed region ends.
    # --no-pipeline times mc_extract_features instead (one pass at a time, host sync inside).
    step_no = [0]

    def next_table(full):
        i = step_no[0] % n_tables
        step_no[0] += 1
        dev.select_table(slots[i], as_new=full)

    def step_sync(full):
        next_table(full)
        dev.run(6, 0, 0.0, tail_contig=-1, score=True)
        return dev.fetch(copy=False)

    def run_steps(n_steps, on_done, full=True):
        if args.no_pipeline:
            for _ in range(n_steps):
                on_done(step_sync(full))
            return
        depth = max(1, min(args.depth, 5, n_steps))              # passes in flight (the library allows six)

        def enqueue():
            next_table(full)
            dev.run_async(6, 0, 0.0, tail_contig=-1, score=True)
        if depth == 1:                                           # (one pass at a time through the pipelined interface: profiling)
            for _ in range(n_steps):
                enqueue()
                on_done(dev.wait())
            return
        for _ in range(depth):
            enqueue()
        dev.wait_begin()                                         # copy-out of the oldest pass started
        for _ in range(n_steps - depth):
            dev.wait_begin()                                     # ... and of the one behind it, as soon as it is computed:
            enqueue()                                            # the transfers run back to back; the next pass enqueued;
            on_done(dev.wait())                                  # the oldest pass's records are in host memory
        for _ in range(depth):
            dev.wait_begin()                                     # (the copy-out of the pass behind the oldest one starts before the oldest is waited for)
            on_done(dev.wait())

    def barrier():
        if dist is not None:
            dist.barrier()

    k1_ms, tot_ms, last, calls_seen = [], [], [None], [0, 0]

    # The kernel times come from hipEvents on the ctx stream.  An event between two kernels costs that queue ~9 us (6 % of a
    # pass), so in the pipelined loop only every n-th pass carries the two events that do nothing but time it; kernel_ms is
    # an average over those passes of the timed region (one pass at a time: every pass).
    time_every = 1 if args.no_pipeline else max(1, min(args.time_every, max(1, args.steps // 2)))
    dev.set_pass_timing(time_every)

    done_at = []

    def on_done(rec):
        done_at.append(time.perf_counter())
        last[0] = rec
        calls_seen[0] += int(rec.n_calls) if getattr(rec, '_compacted', False) else int(((rec.info[:rec.n] & _lib.I_TOO_MANY) == 0).sum())
        calls_seen[1] += 1
        if args.no_pipeline or dev.last_pass_timed():
            tm = dev.times_ms()
            k1_ms.append(tm['window_scan'] + tm['emit'])
            tot_ms.append(tm)

    full = not args.rescan_only
    if args.warmup:
        run_steps(args.warmup, on_done, full)
    del k1_ms[:], tot_ms[:]
    calls_seen[:] = [0, 0]
    barrier()
    dev.sync()                              # nothing of the warm-up is left on any stream
    del done_at[:]
    t0 = time.perf_counter()
    run_steps(args.steps, on_done, full)    # the last wait() returns when the last pass's records are in host memory
    dev.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    # the step time without the pipeline's fill and drain: the MEAN interval between the completions of consecutive passes
    # from the first completion to the last one that had a full pipeline behind it (a 20-step run pays one fill and one drain in
    # 20 steps, a 200-step run in 200: this figure is the same for both; completions come in pairs since the emit went to the side
    # stream -- two streams hand over in turn --, so the median interval is not a step time)
    done = np.array(done_at[:args.steps])
    depth_now = 1 if args.no_pipeline else max(1, min(args.depth, 5, args.steps))
    last_full = len(done) - depth_now                      # (the passes behind it were still being enqueued)
    steady_ms = float((done[last_full] - done[0]) / last_full) * 1e3 if last_full >= 3 else None
    calls_in_region = calls_seen[0]         # calls of all timed steps of this rank (the two tables differ by a few)
    rec = last[0]
    info = rec.info[:rec.n]
    n_calls = int(((info & _lib.I_TOO_MANY) == 0).sum())
    n_records = int(rec.n)
    # what a pipelined pass sends over PCIe (mc_calls_view): 16 B per record; per call k 32-bit slot words, the probability and
    # the mask of its wide slots; the high words of the wide slots
    packed = getattr(rec, '_packed', None)
    copy_out_bytes = None if packed is None else 16 * n_records + n_calls * (4 * 6 + 8 + 1) + 4 * int(len(packed[1]))

    calls_total, elapsed_max, calls_region_total = n_calls, elapsed, calls_in_region
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([n_calls, calls_in_region], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        elapsed_max, calls_total, calls_region_total = float(t[0]), int(c[0]), int(c[1])

    # ---- the same steps with the classifier in fp64 throughout (MCALLER_MLP_FP64=1: how rounds 1-4 scored every record);
    #      the timed region above runs the default, the fast forward (hidden layer f32, output and near-boundary records f64) ----
    fp64_ms = None
    if full and not args.no_pipeline and not os.environ.get('MCALLER_MLP_FP64'):
        saved = (list(k1_ms), list(tot_ms), last[0], list(calls_seen), list(done_at))
        os.environ['MCALLER_MLP_FP64'] = '1'
        try:
            dev.set_mlp(weights, soc)                           # (the switch is read when the classifier is set)
            run_steps(max(args.warmup, 3), on_done, True)
            dev.sync()
            t_f = time.perf_counter()
            run_steps(args.steps, on_done, True)
            dev.sync()
            fp64_ms = (time.perf_counter() - t_f) / args.steps * 1e3
        finally:
            del os.environ['MCALLER_MLP_FP64']
            dev.set_mlp(weights, soc)
        k1_ms[:], tot_ms[:] = saved[0], saved[1]
        last[0] = saved[2]
        calls_seen[:] = saved[3]
        done_at[:] = saved[4]

    # ---- the rate of repeated passes over ONE validated table (config.resident_rescan): what round 2 reported as `value` ----
    rescan = None
    if full and not args.no_pipeline:
        dev.select_table(slots[0])
        n_tables_saved, n_tables = n_tables, 1
        step_no[0] = 0
        saved = (list(k1_ms), list(tot_ms), last[0], list(calls_seen))
        run_steps(max(args.warmup, 4), on_done, False)         # (the second pass streams the positions, the third builds the summaries)
        dev.sync()
        barrier()
        t_r = time.perf_counter()
        run_steps(args.steps, on_done, False)
        dev.sync()
        barrier()
        t_r = time.perf_counter() - t_r
        rescan = {'ms_per_pass': t_r / args.steps * 1e3, 'calls_per_s': n_calls * args.steps / t_r,
                  'what': 'pipelined passes over one table that earlier passes have validated and summarised (k_summarize): the '
                          'scan reads the 1 B/row unit summaries; no table of a file is ever scanned like this'}
        n_tables = n_tables_saved
        rec_red, table_red = last[0], tables[0][0]            # (the records the device holds now: what the site reduction reduces)
        k1_ms[:], tot_ms[:] = saved[0], saved[1]
        last[0] = saved[2]
        calls_seen[:] = saved[3]
    else:
        rec_red, table_red = rec, tables[(step_no[0] - 1) % n_tables][0]
    n_calls_red = int(((rec_red.info[:rec_red.n] & _lib.I_TOO_MANY) == 0).sum())

    # the one exchange step of the multi-GPU job: per-site counts summed over ranks (feeds make_bed).  Outside the timed
    # steps.  Counted on the device from the records of the last step and all-reduced with RCCL through the C ABI
    # (mc_site_counts / mc_site_allreduce); if that fails (e.g. a plumbing test with two ranks on one GPU) the same
    # reduction goes through torch.distributed's gloo group from the host copy of the records.
    reduction = None
    reduction_hung = False
    if dist is not None:
        import threading
        box = {}

        def site_reduction_leg():
            reduction = None
            from mcaller_amd import make_bed
            index = make_bed.SiteIndex(ref.meth, 1)
            e_native = None
            try:
                uid = [None]
                try:
                    if rank == 0:
                        uid = [Device.comm_unique_id()]           # loads librccl.so
                finally:
                    dist.broadcast_object_list(uid, src=0)
                if uid[0] is None:
                    raise RuntimeError('rank 0 could not create an RCCL unique id')
                dev.comm_init(world, rank, uid[0])
                dev.site_counts(row_offset=rank * n_rows)
                n_meth, n_total, fmin, ms = dev.site_allreduce()
            except Exception as e:                                 # noqa
                e_native = e
            flags = [None] * world
            dist.all_gather_object(flags, (e_native is None, n_calls_red))     # every rank takes the same branch
            expected = sum(f[1] for f in flags)
            flags = [f[0] for f in flags]
            if all(flags):
                reduction = {'backend': 'rccl (ncclAllReduce through the C ABI)', 'ms': ms, 'observations': int(n_total.sum()),
                             'observations_expected': expected, 'bytes': int(index.n * 16), 'sites': int(index.n)}
            else:
                try:
                    import torch
                    counts = make_bed.site_counts(rec_red, table_red, index, row_offset=rank * n_rows)

                    # (the same reduction through torch.distributed's gloo group, from the host copy of the records)
                    packed = torch.from_numpy(np.stack([counts[0], counts[1]]))
                    fmin = torch.from_numpy(counts[2].copy())
                    t_r = time.perf_counter()
                    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
                    dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
                    ms = (time.perf_counter() - t_r) * 1e3
                    backend = 'gloo (native RCCL path: %s)' % e_native
                    reduction = {'backend': backend, 'ms': ms, 'observations': int(packed[1].sum().item()),
                                 'observations_expected': expected,
                                 'bytes': int(packed.numel() * packed.element_size() + fmin.numel() * 8)}
                except Exception as e:                             # noqa
                    reduction = {'error': '%s: %s' % (type(e).__name__, e)}

            box['reduction'] = reduction

        # (a collective that never returns -- a rank that died, a fabric that does not come up -- must not take the measured
        # line with it: the leg runs beside the main thread and is given two minutes)
        th = threading.Thread(target=site_reduction_leg, daemon=True)
        th.start()
        th.join(float(os.environ.get('MCALLER_BENCH_REDUCTION_TIMEOUT', '120')))
        if th.is_alive():
            reduction_hung = True
            reduction = {'error': 'the per-site reduction did not return within its time limit'}
        else:
            reduction = box.get('reduction')

    # ---- the ranks > 0 are done.  Rank 0 goes on alone with the product-shaped legs -- at every N -- once the others have let
    #      go of their GPUs (the workers of the strong-scaling leg take all N) ----
    if dist is not None:
        if rank != 0:
            dev.close()
            dev = None
        if reduction_hung:               # (a thread is stuck inside a collective: no orderly shutdown, no barrier to trust)
            if rank != 0:
                sys.stdout.flush()
                os._exit(4)
        else:
            barrier()
            dist.destroy_process_group()
        dist = None
        if rank != 0:
            return
        try:                             # (rank 0 is alone now: its solo legs and the workers it starts may use every core again;
            os.sched_setaffinity(0, affinity_at_start)          # a worker binds itself to its own GPU's node)
        except OSError:
            pass

    # ---- the kernels one at a time (outside the timed region): hipEvents around every stage of a synchronous FULL pass ----
    sync_ms = []
    for _ in range(7):
        step_sync(full)
        sync_ms.append(dev.times_ms())
    sync_ms = {k: float(np.median([t[k] for t in sync_ms[1:]])) for k in sync_ms[0]}
    # ... and through the pipelined interface, one pass in flight: what the timed steps launch (a one-base motif: K0 + k1_fused
    # in place of the scan, the ordering kernels and k1_emit_runs of the synchronous pass)
    dev.set_pass_timing(1)
    one_ms = []
    for _ in range(7):
        next_table(full)
        dev.run_async(6, 0, 0.0, tail_contig=-1, score=True)
        dev.wait()
        one_ms.append(dev.times_ms())
    fused_room = dev.last_pass_info()[0]
    one_ms = {k: float(np.median([t[k] for t in one_ms[1:]])) for k in one_ms[0]}
    dev.set_pass_timing(time_every)

    # ---- device end to end: distinct shards from pinned host memory through the table slots (one GPU: rank 0's) ----
    device_e2e, shards = None, []
    if args.stream_shards > 0:
        try:
            S = args.stream_shards
            per = n_rows // S
            for i in range(S):
                t_i, q_i = synth.make_table(per, seed=5000 + i, codes=codes)
                shards.append((t_i.pinned(), q_i, t_i if i == 0 else None))     # (the first one is also the file-to-file table)
            dev.reserve_tables(max(s[0].n_rows for s in shards), max(s[0].n_seg for s in shards), max(s[0].n_reads for s in shards))
            total_rows = sum(s[0].n_rows for s in shards)
            total_bytes = 17.0 * total_rows

            def stream_once(with_passes):
                calls, in_flight = 0, 0
                t_s = time.perf_counter()
                slot = -1
                for i, (tp, q, _) in enumerate(shards):
                    slot = dev.upload_table_async(tp, q)
                    if with_passes:
                        dev.run_async(6, 0, 0.0, tail_contig=(0 if i + 1 < S else -1), score=True)
                        in_flight += 1
                        if in_flight > 2:
                            r = dev.wait()
                            calls += r.n_calls
                            in_flight -= 1
                while in_flight:
                    r = dev.wait()
                    calls += r.n_calls
                    in_flight -= 1
                if not with_passes:
                    dev.wait_upload(slot)
                dev.sync()
                return time.perf_counter() - t_s, calls

            stream_once(True)                                   # warm-up: record sets and pinned buffers allocated
            h2d = sorted(stream_once(False)[0] for _ in range(3))
            e2e = sorted((stream_once(True) for _ in range(5)), key=lambda x: x[0])
            best, med = e2e[0], e2e[len(e2e) // 2]
            device_e2e = {'shards': S, 'rows_per_shard': per, 'passes_in_flight': 2, 'source': 'pinned host memory (mc_host_alloc)',
                          'seconds_best': best[0], 'seconds_median': med[0], 'calls': best[1],
                          'events_per_s': total_rows / best[0], 'events_per_s_median': total_rows / med[0],
                          'calls_per_s': best[1] / best[0],
                          'h2d_only_seconds': h2d[0], 'h2d_only_events_per_s': total_rows / h2d[0],
                          'h2d_only_GBps': total_bytes / h2d[0] / 1e9,
                          'fraction_of_h2d_rate': h2d[0] / best[0]}
        except Exception as e:                                  # noqa
            device_e2e = {'error': '%s: %s' % (type(e).__name__, e)}

    # ---- file to file: eventalign TSV -> .diffs.6 through the CLI (one GPU: rank 0's) ----
    file_to_file, f2f_dir, f2f_paths, f2f_rows = None, None, None, 0
    if args.f2f_events > 0:
        try:
            import contextlib
            import io
            from mcaller_amd import mCaller
            f2f_rows = int(args.f2f_events)
            if shards and shards[0][2] is not None and shards[0][2].n_rows == f2f_rows:
                t_f, q_f = shards[0][2], shards[0][1]
            else:
                t_f, q_f = synth.make_table(f2f_rows, seed=5000, codes=codes)
            f2f_dir = tempfile.mkdtemp(prefix='mc_f2f_')
            t_w = time.perf_counter()
            f2f_paths = synth.write_inputs(t_f, q_f, codes, f2f_dir)
            t_w = time.perf_counter() - t_w
            os.sync()                                            # (the runs below read the page cache, not a file still being written back)
            if dev is not None:
                dev.close()                                      # the CLI makes its own context
                dev = None
            runs = []
            out_path = f2f_paths['tsv'][:-4] + '.diffs.6'
            for _ in range(10):                                  # (the first runs still pin memory and warm the page cache)
                if os.path.exists(out_path):
                    os.remove(out_path)
                t_r = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    mCaller.main(['-m', 'GATC', '-r', f2f_paths['fasta'], '-e', f2f_paths['tsv'], '-f', f2f_paths['fastq'], '-d', model_npz])
                runs.append(time.perf_counter() - t_r)
            calls_f = sum(1 for _ in open(out_path, 'rb'))
            best, med = min(runs[1:]), float(np.median(runs[1:]))
            file_to_file = {'rows': f2f_rows, 'tsv_bytes': os.path.getsize(f2f_paths['tsv']), 'calls': calls_f,
                            'seconds_first_run': runs[0], 'seconds_best': best, 'seconds_median': med, 'seconds_all': runs,
                            'events_per_s': f2f_rows / med, 'events_per_s_best': f2f_rows / best, 'calls_per_s': calls_f / med,
                            'host_cores': host_cores_info(), 'inputs_written_s': t_w,
                            'what': 'python -m mcaller_amd.mCaller -m GATC: FASTQ qualities, FASTA marking, the text read into pinned '
                                    'memory and parsed on the GPU (mc_ctx_parse_*), shards streamed through the table slots, native row '
                                    'formatter, .diffs.6 written (page cache warm)'}
        except Exception as e:                                  # noqa
            file_to_file = {'error': '%s: %s' % (type(e).__name__, e)}
    file_to_file_dense = {}
    if args.dense_events > 0 and f2f_paths is not None:
        try:
            file_to_file_dense['1e7'] = dense_file_to_file_leg(f2f_dir, f2f_rows, 4)
        except Exception as e:                                  # noqa
            file_to_file_dense['1e7'] = {'error': '%s: %s' % (type(e).__name__, e)}

    # ---- the headline size (10^8 rows, 11.7 GB of text), in processes of their own: file to file on one GPU (wall time per run,
    #      peak RSS), then BASELINE.json configs[3] -- the same file through `mCaller --gpus N --bed` on all N GPUs (strong scaling) ----
    file_to_file_big, strong = None, None
    big_rows = int(max(args.f2f_big_events, args.strong_events, args.dense_events))
    if big_rows > 0:
        big_dir = None
        try:
            import subprocess
            if dev is not None:
                dev.close()                                      # (the legs below run in processes of their own)
                dev = None
            t_b, q_b = (table, qual) if big_rows == n_rows else synth.make_table(big_rows, seed=1000, codes=codes)
            big_dir = tempfile.mkdtemp(prefix='mc_f2f_big_')
            t_w = time.perf_counter()
            synth.write_inputs(t_b, q_b, codes, big_dir)
            t_w = time.perf_counter() - t_w
            os.sync()
            one_gpu_sha = None
            if args.f2f_big_events > 0:
                try:
                    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'file_to_file.py'), '--inputs', big_dir, '--runs', '4',
                                        '--json'], capture_output=True, text=True, timeout=900)
                    if r.returncode != 0:
                        raise RuntimeError(r.stderr[-500:])
                    res = json.loads(r.stdout.strip().splitlines()[-1])
                    runs_b = res['seconds_all']
                    med_b = float(np.median(runs_b[1:]))
                    one_gpu_sha = res['diffs_sha256']
                    file_to_file_big = {'rows': big_rows, 'tsv_bytes': res['tsv_bytes'], 'diffs_bytes': res['diffs_bytes'], 'calls': res['calls'],
                                        'seconds_first_run': runs_b[0], 'seconds_best': min(runs_b[1:]), 'seconds_median': med_b,
                                        'seconds_all': runs_b, 'events_per_s': big_rows / med_b, 'calls_per_s': res['calls'] / med_b,
                                        'text_GBps': res['tsv_bytes'] / med_b / 1e9, 'peak_rss_mb': res['peak_rss_mb'], 'inputs_written_s': t_w,
                                        'what': 'python tools/file_to_file.py --inputs ... --runs 4 (the CLI, four times in one process of its '
                                                'own; page cache warm): rows are appended to the output shard by shard, memory is bounded by '
                                                'the shards in flight'}
                except Exception as e:                          # noqa
                    file_to_file_big = {'error': '%s: %s' % (type(e).__name__, e)}
            if args.dense_events > 0:
                try:
                    file_to_file_dense['big'] = dense_file_to_file_leg(big_dir, big_rows, 3)
                except Exception as e:                          # noqa
                    file_to_file_dense['big'] = {'error': '%s: %s' % (type(e).__name__, e)}
            if args.strong_events > 0:
                try:
                    strong = strong_scaling_leg(big_dir, world, rows=big_rows, one_gpu_sha=one_gpu_sha,
                                                one_device=bool(os.environ.get('MCALLER_BENCH_ONE_DEVICE')))
                except Exception as e:                          # noqa
                    strong = dict(strong_scaling_leg(None, world, dry=True), error='%s: %s' % (type(e).__name__, e))
                if world == 1:
                    # a PROJECTION of the curve: one worker, its 1/N piece of the file, 1/N of the host threads, on this GPU
                    try:
                        r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'project_scaling.py'), '--inputs', big_dir,
                                            '--parts', '2,4,8', '--runs', '3', '--json'], capture_output=True, text=True, timeout=900)
                        if r.returncode != 0:
                            raise RuntimeError(r.stderr[-500:])
                        proj = json.loads(r.stdout.strip().splitlines()[-1])
                        t1 = strong.get('seconds_median') if strong else None
                        rows_p = [dict(n_gpus=1, seconds=t1, measured=True)]
                        for w in proj['per_n']:
                            if 'error' in w:
                                rows_p.append(dict(n_gpus=w['n_parts'], error=w['error']))
                                continue
                            rows_p.append(dict(n_gpus=w['n_parts'], seconds=w['seconds_median'], measured=False, host_threads_per_worker=w['host_threads'],
                                               worker_text_GBps=w['text_GBps'], rows=w['rows'], speedup_over_one_gpu=(t1 / w['seconds_median']) if t1 else None,
                                               efficiency=(t1 / w['seconds_median'] / w['n_parts']) if t1 else None,
                                               main_thread_s={k: w['stream_last_run'].get(k) for k in ('wait_parser', 'enqueue', 'hand_out', 'parse')}))
                        strong['projected'] = {'label': 'PROJECTION, not a measurement of N GPUs', 'per_n': rows_p, 'what': proj['what'],
                                               'host_threads_total': proj['per_n'][0].get('host_threads_of') if proj['per_n'] and 'error' not in proj['per_n'][0] else None}
                    except Exception as e:                      # noqa
                        strong['projected'] = {'error': '%s: %s' % (type(e).__name__, e)}
        except Exception as e:                                  # noqa
            file_to_file_big = file_to_file_big or {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            if big_dir:
                shutil.rmtree(big_dir, ignore_errors=True)

    # ---- BASELINE.json configs[4]: --train on labelled positions + the RF classifier, in a process of its own ----
    config5, config5_dir = None, None
    if args.config5_events > 0:
        try:
            if dev is not None:
                dev.close()
                dev = None
            config5_dir = tempfile.mkdtemp(prefix='mc_config5_')
            config5 = config5_leg(config5_dir, args.config5_events)
        except Exception as e:                                  # noqa
            config5 = {'error': '%s: %s' % (type(e).__name__, e)}

    # ---- text end to end: the same file, its text already in pinned host memory, parsed on the GPU shard after shard, a pass
    #      over every shard, records back in host memory (N = 1): what the link allows for eventalign TEXT ----
    text_e2e, roofline_parser = None, None
    if file_to_file and 'error' not in file_to_file:
        try:
            from mcaller_amd import _lib
            from mcaller_amd import extract_contexts as ec
            from mcaller_amd.read_qual import extract_read_quality
            from mcaller_amd.refmark import MarkedReference
            d2 = ec.get_device()                                 # (the CLI runs above left reference and classifier in place)
            tsv = f2f_paths['tsv']
            ref2 = MarkedReference(f2f_paths['fasta'], 'A', 'GATC', None)
            ref2.quiet = True
            ref2.mark(0)
            d2.set_reference(ref2.device_arrays())
            r2q = extract_read_quality(f2f_paths['fastq'])
            lo, hi = _lib.eventalign_consumed_range(tsv, 0, os.path.getsize(tsv))
            cuts = _lib.eventalign_read_cuts(tsv, 8, lo, hi)
            texts = [_lib.TextBlock(tsv, cuts[i], cuts[i + 1]) for i in range(8) if cuts[i + 1] > cuts[i]]
            rows_cap = max(t.n_bytes for t in texts) // 48 + 65536
            d2.reserve_tables(rows_cap, rows_cap // 16, rows_cap // 16)

            def text_once():
                calls, rows, in_flight, parsing = 0, 0, 0, []
                t_s = time.perf_counter()

                def finish_one(last):
                    nonlocal calls, rows, in_flight
                    slot, text = parsing.pop(0)
                    table = d2.parse_end(slot, text)
                    P_t = ec.prepare_table(ec.Prepared(), table, ref2, r2q, quiet=True)
                    rows += table.n_rows
                    d2.upload_table_async(P_t.table, P_t.qual)
                    d2.run_async(6, 0, 0.0, tail_contig=(-1 if last else 0), score=True)
                    in_flight += 1
                    if in_flight > 2:
                        calls += d2.wait().n_calls
                        in_flight -= 1
                for i, text in enumerate(texts):
                    parsing.append((d2.parse_begin(text, ref2.names, rows_cap), text))
                    if len(parsing) > 3:
                        finish_one(False)
                while parsing:
                    finish_one(len(parsing) == 1)
                while in_flight:
                    calls += d2.wait().n_calls
                    in_flight -= 1
                d2.sync()
                return time.perf_counter() - t_s, calls, rows

            text_once()
            runs_t = sorted((text_once() for _ in range(5)), key=lambda x: x[0])
            best_t, n_bytes_t = runs_t[0], sum(t.n_bytes for t in texts)
            text_e2e = {'shards': len(texts), 'text_bytes': n_bytes_t, 'rows': best_t[2], 'calls': best_t[1],
                        'seconds_best': best_t[0], 'seconds_median': runs_t[len(runs_t) // 2][0],
                        'events_per_s': best_t[2] / best_t[0], 'calls_per_s': best_t[1] / best_t[0],
                        'text_GBps': n_bytes_t / best_t[0] / 1e9,
                        'fraction_of_h2d_rate': (n_bytes_t / best_t[0] / 1e9) / device_e2e['h2d_only_GBps']
                        if device_e2e and 'h2d_only_GBps' in device_e2e else None,
                        'what': 'eventalign text in pinned host memory -> mc_ctx_parse_begin/_end/_finish (line starts, tokens, '
                                'numbers, segments on the GPU) -> K0-K2 -> records in host memory; three shards of text ahead'}
            # ---- the device parser alone (roofline_parser): ONE shard's text parsed with nothing else in flight, hipEvents
            #      around the parser's kernels (mc_ctx_parse_times_ms) ----
            try:
                text0 = max(texts, key=lambda t: t.n_bytes)
                pm, hm, rows0 = [], [], 0
                for _ in range(6):
                    slot = d2.parse_begin(text0, ref2.names, rows_cap)
                    h2d_ms, parse_ms = d2.parse_times_ms(slot)
                    tab = d2.parse_end(slot, text0)
                    if tab is None:
                        raise RuntimeError('the device parser declined the shard')
                    rows0 = tab.n_rows
                    d2.parse_abandon(slot)
                    pm.append(parse_ms)
                    hm.append(h2d_ms)
                p_ms, alg = float(np.median(pm[1:])), float(text0.n_bytes + 17.0 * rows0)
                roofline_parser = {'bound': 'hbm', 'kernel': 'kp_count + kp_scan + kp_starts + kp_parse + kp_count_rows + kp_scan + kp_place (+ the '
                                   'small copies that hand the result out), one shard of text, nothing else in flight',
                                   'achieved': alg / (p_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': alg / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                                   'algorithmic_bytes': alg, 'algorithmic_bytes_what': 'the shard\'s text read once (%d B) + 17 B/row of columns '
                                   'written (%d rows)' % (text0.n_bytes, rows0),
                                   'kernel_ms': p_ms, 'text_h2d_ms': float(np.median(hm[1:])),
                                   'text_GBps_parser': text0.n_bytes / (p_ms * 1e-3) / 1e9,
                                   'text_GBps_link': text0.n_bytes / (float(np.median(hm[1:])) * 1e-3) / 1e9,
                                   'note': 'one PCIe link feeds one GPU: the parser is hidden behind the text\'s transfer as long as '
                                           'kernel_ms < text_h2d_ms; low priority while that holds (DESIGN.md section 9)'}
            except Exception as e:                              # noqa
                roofline_parser = {'error': '%s: %s' % (type(e).__name__, e)}
            del texts
        except Exception as e:                                  # noqa
            text_e2e = {'error': '%s: %s' % (type(e).__name__, e)}

    if rank == 0:
        kernel_ms = {k: float(np.mean([t[k] for t in tot_ms])) for k in tot_ms[0]}
        if kernel_ms.get('emit') == 0.0:      # pipelined passes time scan + ordering + emit as one span (no event in between)
            kernel_ms['window_scan_and_emit'] = kernel_ms.pop('window_scan')
            del kernel_ms['emit']
        calls_per_step = calls_region_total / float(args.steps * world)          # (per rank and step: the two tables differ by a few)
        alg_bytes = 17.0 * n_rows + 64.0 * calls_per_step
        # every kernel that touches a table once, one full pass at a time (hipEvents on the ctx stream)
        scan_ms = sync_ms['window_scan']
        per_table = {'strand_resolve': sync_ms['strand_resolve'], 'window_scan': scan_ms, 'order_and_emit': sync_ms['emit']}
        per_table_ms = float(sum(per_table.values()))
        achieved = alg_bytes / (per_table_ms * 1e-3) / 1e9
        # HBM bytes those kernels move per table: rocprofv3 PMC passes of this workload (tools/collect_profiles.sh), committed under
        # profiles/ -- quoted only if they were collected on the kernel sources this library was built from
        traffic, traffic_source, scan_traffic, per_kernel_traffic = None, None, None, None
        src_hash = kernel_source_hash()
        if os.path.exists(PMC_FILE) and n_rows == 100000000 and args.motif == 'GATC' and full:
            pmc = json.load(open(PMC_FILE))
            if pmc.get('kernel_source_sha16') == src_hash:
                per = pmc.get('per_launch_bytes_corrected', {})
                names = pmc.get('per_table_kernels') or []
                if names and all(n in per for n in names):
                    per_kernel_traffic = {n: per[n].get('hbm_bytes') for n in names}
                    traffic = float(sum(per_kernel_traffic.values()))
                    scan_traffic = per.get('k1_scan', {}).get('hbm_bytes')
                    traffic_source = ('%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` at commit %s, kernel sources %s '
                                      '(FETCH_SIZE x2, gfx950 correction); sum over %s; not re-measured in this run'
                                      % (os.path.relpath(PMC_FILE, REPO), pmc.get('workload', '?'), pmc.get('head', '?'), src_hash,
                                         ' + '.join(names)))
            else:
                traffic_source = ('%s was collected on other kernel sources (%s, running %s): no traffic figure'
                                  % (os.path.relpath(PMC_FILE, REPO), pmc.get('kernel_source_sha16'), src_hash))
        out = {
            'metric': METRIC,
            'value': calls_region_total / elapsed_max,
            'unit': 'calls/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed_max / args.steps * 1e3,
            'ms_per_step_steady': steady_ms,
            'ms_per_step_steady_what': 'mean interval between the completions of consecutive passes of the timed region on rank 0, from the '
                                       'first completion to the last one with a full pipeline behind it (the pipeline\'s fill and drain, which a '
                                       'short run pays inside ms_per_step, left out)',
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': DTYPE,
            'ms_per_step_fp64_mlp': fp64_ms,
            'data': 'synthetic',
            'config': {'workload': 'synthetic %.0e eventalign rows per GPU and step, -m %s, NN classifier (r95 two-base MLP), '
                                   'skip_thresh 0; %s' % (n_rows, args.motif,
                                                          'two resident tables taken in turn, every step a FULL pass (the table '
                                                          'declared new: positions + event indices streamed, every row validated)'
                                                          if full else 'ONE validated resident table re-scanned (--rescan-only)'),
                       'step': 'full pass' if full else 'resident rescan',
                       'passes_in_flight': 1 if args.no_pipeline else max(1, min(args.depth, 5, args.steps)),
                       'events_per_gpu': n_rows, 'calls_per_gpu': calls_per_step, 'flush_records_per_gpu': n_records,
                       'copy_out_bytes_per_pass': copy_out_bytes,
                       'events_per_s': n_rows * world * args.steps / elapsed_max,
                       'algorithmic_GBps_of_the_step': alg_bytes * world / (elapsed_max / args.steps) / 1e9,
                       'kernel_ms': kernel_ms, 'kernel_ms_from_passes': len(tot_ms), 'timing_events_every_n_passes': time_every,
                       'kernel_ms_one_pass_at_a_time': sync_ms,
                       'kernel_ms_pipelined_one_in_flight': dict(one_ms, fused_room=fused_room,
                                                                 what='hipEvents around the stages of pipelined passes, one in flight: strand_resolve = '
                                                                      'k0_first_site, window_scan = scan + ordering + emit as one span'
                                                                      + (' = k1_fused (one kernel, %d record slots per 960-row piece)' % fused_room
                                                                         if fused_room else '')),
                       'per_table_kernel_ms': dict(per_table, total=per_table_ms, classifier=sync_ms['classifier']),
                       'resident_rescan': rescan,
                       'h2d_table_s': t_up, 'generate_s': t_gen, 'site_reduction': reduction, 'numa_node_rank0': numa_node,
                       # SURVEY.md 8(d)'s three timings: kernels only; device end to end; file to file
                       'calls_per_s_kernels_only': calls_per_step / (float(np.mean([t['total'] for t in tot_ms])) * 1e-3),
                       'device_e2e': device_e2e,
                       'device_e2e_events_per_s': (device_e2e or {}).get('events_per_s'),
                       'strong_scaling': 'see the top-level key (BASELINE.json configs[3]: one file, mCaller --gpus N --bed)',
                       'file_to_file': file_to_file, 'file_to_file_1e8': file_to_file_big, 'text_e2e': text_e2e},
            'roofline': {'bound': 'hbm',
                         'kernel': 'every kernel that touches a table once: k0_first_site (makes the name-block templates too) + k1_scan (validating) + '
                                   'k1_group_scan + k1_list + k1_emit',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_source, 'traffic_per_kernel': per_kernel_traffic,
                         'algorithmic_bytes': alg_bytes, 'algorithmic_bytes_what': '17 B/event row + 64 B/call (SURVEY.md 8(d))',
                         'kernels_ms': per_table, 'kernel_ms': per_table_ms,
                         'kernel_ms_source': 'hipEvents on the ctx stream around the stages of synchronous full passes, median of 6 '
                                             'after the timed region (the classifier touches records, not rows: not in the sum)',
                         'scan': {'kernel': 'k1_scan<64, SCAN_VALIDATE>: streams 8 of the 17 B/row (positions, event indices), '
                                            'validates every row, lists and decides the candidate units',
                                  'kernel_ms': scan_ms, 'streamed_bytes': 8.0 * n_rows,
                                  'streamed_GBps': 8.0 * n_rows / (scan_ms * 1e-3) / 1e9,
                                  'streamed_frac': 8.0 * n_rows / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  'traffic': scan_traffic,
                                  'traffic_frac': (scan_traffic / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if scan_traffic else None},
                         'pipelined': {'what': 'the same algorithmic bytes over the steady step time of the timed (pipelined) steps -- everything a pass '
                                               'costs the pipeline: the ctx stream\'s kernels, the emit and the side kernel beside the next scan, the copy-out',
                                       'kernel_ms': steady_ms if steady_ms else elapsed_max / args.steps * 1e3,
                                       'frac': alg_bytes / ((steady_ms if steady_ms else elapsed_max / args.steps * 1e3) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       'ctx_stream_spans_ms': float(np.mean([t['total'] - t['classifier'] for t in tot_ms]))},
                         'note': 'the event/model pairs (8 of the 17 B/row) are only read for the rows of closed windows and the flag '
                                 'bytes only for the listed units, so the kernels move about half the algorithmic bytes: a frac '
                                 'near 1 would not mean 8 TB/s of traffic'},
        }
        # the story in the order a user meets it: a FILE goes through at file_to_file_calls_per_s (parser, link and row formatter
        # included), columns that are already parsed at device_e2e_calls_per_s (the link is the bound), and `value` is the rate
        # of the resident-table passes the roofline is quoted on; each with the CPU leg that does the same work
        out['file_to_file_calls_per_s'] = (file_to_file_big or {}).get('calls_per_s')
        out['file_to_file_what'] = ('config.file_to_file_1e8: 10^8 rows of eventalign text -> .diffs.6 through the CLI on one GPU, median '
                                    'of the warm runs; like-for-like CPU leg: cpu_baseline_reference_like')
        out['device_e2e_calls_per_s'] = (device_e2e or {}).get('calls_per_s')
        out['device_e2e_what'] = ('config.device_e2e: parsed columns from pinned host memory, records back in host memory; like-for-like '
                                  'CPU leg: cpu_baseline_all_cores')
        out['strong_scaling'] = strong
        if fused_room:
            f_ms = one_ms['strand_resolve'] + one_ms['window_scan'] + one_ms['emit']
            out['roofline_fused_dense'] = {'bound': 'hbm', 'kernel': 'k0_first_site + k1_fused (validating): every kernel that touches a table once in a '
                                           'pipelined pass over a dense reference', 'achieved': alg_bytes / (f_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                                           'unit': 'GB/s', 'frac': alg_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                                           'algorithmic_bytes': alg_bytes, 'kernel_ms': f_ms,
                                           'the_pair_ms': per_table_ms, 'the_pair_frac': achieved / HBM_PEAK_GBS,
                                           'what': 'the same algorithmic bytes over K0 + k1_fused (hipEvents, pipelined passes one in flight); the_pair_*: '
                                                   'K0 + k1_scan<130> + ordering + k1_emit_runs of the synchronous pass (roofline.frac)'}
        out['file_to_file_dense'] = file_to_file_dense or None
        out['file_to_file_dense_what'] = ('`mCaller -m A` file to file through the CLI on one GPU at 10^7 rows ("1e7") and at the headline size '
                                          '("big"): seconds, phase split of the main thread, which phase bounds the mode')
        out['config5'] = config5
        out['roofline_parser'] = roofline_parser
        if world > 1 and strong and strong.get('calls_per_s'):
            out['strong_value'] = strong['calls_per_s']
        out['scaling_note'] = ('`value` is the weak leg: every rank passes over its own resident tables, nothing crosses GPUs inside the timed '
                               'region, so it grows N-fold by construction.  What BASELINE.json configs[3] asks -- ONE 10^8-row file over N '
                               'GPUs, per-site reduction included -- is strong_scaling.calls_per_s (top-level strong_value at N > 1); on one '
                               'GPU strong_scaling.projected carries measured per-worker seconds at 1/2, 1/4, 1/8 of the file and of the '
                               'host threads')
        if not args.no_cpu_baseline:                         # the CPU legs: on rank 0, at every N
            from tests import helpers as H                   # the checker (oracle/), timed as the CPU baseline
            n_cpu = min(n_rows, int(args.cpu_events))
            sub = table if n_cpu == n_rows else table.slice_segments(
                0, int(np.searchsorted(table.seg_row_begin, n_cpu, side='left')))
            arrays = ref.device_arrays()
            t1 = time.perf_counter()
            orc = H.oracle_records(sub, arrays, qual, 6, 0, 0.0)
            H.oracle_score(orc, sub, qual, weights, soc, 6)
            dt = time.perf_counter() - t1
            oc = int(((orc.info[:orc.n] & _lib.I_TOO_MANY) == 0).sum())
            # the same port on all host cores: shards by read (what the reference's -t does with processes), one thread each
            try:
                from concurrent.futures import ThreadPoolExecutor
                from tests import shard                    # (the checker side: shards of a table for the CPU legs)
                hc = host_cores_info()
                cores = hc['effective']                    # threads = the cores the container may use (quota), not the affinity mask's 256
                bounds = [b for b in shard.shard_bounds(sub, cores) if b[1] > b[0]]
                subs = [(sub.slice_segments(lo, hi), shard.tail_contig(sub, qual, 0.0, hi)) for lo, hi in bounds]

                def one(job):
                    st, tail = job
                    o = H.oracle_records(st, arrays, qual, 6, 0, 0.0, tail_contig=tail)
                    H.oracle_score(o, st, qual, weights, soc, 6)
                    return int(((o.info[:o.n] & _lib.I_TOO_MANY) == 0).sum())
                reps, mt_calls = 0, 0
                t2 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=len(subs)) as ex:
                    while reps < 3 or (time.perf_counter() - t2 < 2.0 and reps < 64):      # (one pass is a 0.1-s burst: a few of them)
                        mt_calls += sum(ex.map(one, subs))
                        reps += 1
                dt2 = time.perf_counter() - t2
                out['cpu_baseline_all_cores'] = {'value': mt_calls / dt2, 'unit': 'calls/s', 'cores': len(subs), 'kind': 'port',
                                                 'cpu_quota': hc['cpu_quota'], 'affinity': hc['affinity'], 'cpu_model': cpu_model(),
                                                 'sample': 'same rows, sharded by read over %d threads, %d times, %.2f s' % (len(subs), reps, dt2),
                                                 'events_per_s': sub.n_rows * reps / dt2}
            except Exception as e:                              # noqa
                out['cpu_baseline_all_cores'] = {'error': str(e)}
            out['cpu_baseline'] = {'value': oc / dt, 'unit': 'calls/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(),
                                   'cpu_quota': host_cores_info()['cpu_quota'],
                                   'sample': '%d event rows of the same workload (C oracle oracle/mc_oracle.c: literal window machine + '
                                             'MLP, ONE host core, %.2f s)' % (sub.n_rows, dt),
                                   'events_per_s': sub.n_rows / dt}
            if f2f_paths is not None:
                try:
                    out['cpu_baseline_reference_like'] = python_twin_baseline(f2f_paths, model_npz, f2f_rows)
                except Exception as e:                          # noqa
                    out['cpu_baseline_reference_like'] = {'error': '%s: %s' % (type(e).__name__, e)}
                if file_to_file_dense:                           # the dense mode's like-for-like CPU leg, on the same 10^7-row file
                    try:
                        file_to_file_dense['cpu_baseline_reference_like'] = python_twin_baseline(f2f_paths, model_npz, f2f_rows, motif='A',
                                                                                                 fraction=0.25)
                    except Exception as e:                      # noqa
                        file_to_file_dense['cpu_baseline_reference_like'] = {'error': '%s: %s' % (type(e).__name__, e)}
            if config5 and 'error' not in config5 and config5.get('inputs'):
                try:
                    config5['cpu_baseline'] = config5_cpu_leg(config5['inputs'])
                except Exception as e:                          # noqa
                    config5['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if config5:
            config5.pop('inputs', None)
        # the pairing SURVEY 8(d) asks for: the file-to-file headline beside the CPU leg that does the same work (the reference's
        # byte-range fan-out, one predict_proba-equivalent per observation); `value` beside the C port on all cores.
        # vs_baseline stays null: BASELINE.md holds no published number for this metric
        twin, allc = out.get('cpu_baseline_reference_like') or {}, out.get('cpu_baseline_all_cores') or {}
        if out.get('file_to_file_calls_per_s') and twin.get('value'):
            out['file_to_file_vs_cpu_reference_like'] = out['file_to_file_calls_per_s'] / twin['value']
        if allc.get('value'):
            out['value_vs_cpu_all_cores'] = out['value'] / world / allc['value']
        write_details(out)
        print_line(contract_line(out))
    if f2f_dir:
        shutil.rmtree(f2f_dir, ignore_errors=True)
    if config5_dir:
        shutil.rmtree(config5_dir, ignore_errors=True)
    if reduction_hung:                   # (a thread is stuck inside a collective: no orderly shutdown, and not a clean exit either)
        sys.stdout.flush()
        sys.stderr.write('bench.py: the per-site reduction hung; the line above carries its error\n')
        sys.stderr.flush()
        os._exit(4)
    if dev is not None:
        dev.close()


if __name__ == '__main__':
    main()
