"""One eventalign file on several GPUs of a node: reads shard embarrassingly.

The byte range the reference's loop consumes (extract_contexts.py:141-148) is cut at the first lines of reads into one
piece per GPU (`mc_eventalign_read_cuts_range`); one worker process per GPU STREAMS its piece through its GPU exactly as the
one-GPU path streams a whole file (`extract_contexts.stream_features`: the text read into pinned memory, parsed on the device,
two passes in flight, every shard's rows appended to the worker's part file as they come back); the parent concatenates the
parts in file order, which is the reference's `-t 1` order (extract_contexts.py:179,242: a window never spans two reads).
Two things cross a cut: the first unfiltered row after a piece closes that piece's last window (R6) and supplies its `chrom`
column (R8) -> `tail`, which a worker reports as soon as its first shard has been parsed and needs only when it enqueues its
last; and `last_read` (:161-174), which can only matter for the reads behind a cut up to the first one with a site row against
the reads in front of it from the last one with a site row on (extract_contexts.cut_names) -> the parent compares those names
when the workers are done; a name on both sides of a cut, and the file is not cut at all (the caller falls back to one GPU).  A
read name that comes back anywhere else in the file changes nothing: the pieces stand.  No collective is needed for the
`.diffs` file; the per-site reduction feeding make_bed is the one exchange step (mc_site_allreduce, RCCL).

The protocol between the parent and its workers: every step that all workers take together is decided by the parent for ALL
of them at once -- a worker never enters a collective (ncclCommInitRank, ncclAllReduce) that another worker may not reach.

  1. worker -> parent: the contig of its first unfiltered row ("head"), as soon as it is known
     parent -> worker: go(tail) -- the head of the next piece that has one -- or abort
  2. worker -> parent: done (counters, read names, rows) or the reference's exit path (`unstreamable`)
     parent: a read name on both sides of a cut (cut_names), an exit path -> abort (the caller runs the one-GPU path)
  with --bed, everything rank-local (the device-side counts, shard after shard) has happened inside step 2; then
  3. parent -> worker: probe            worker -> parent: can RCCL be loaded here?  (rank 0: the unique id)
  4. parent -> worker: init(uid) | host worker -> parent: communicator up / not
  5. parent -> worker: rccl | host      worker -> parent: its own counts, and (rccl) the all-reduced ones
     the parent uses rank 0's all-reduced counts only if EVERY worker reports the collective done; else it sums the workers'
     own counts on the host.

Every wait of the parent has a deadline (MCALLER_WORKER_TIMEOUT seconds, default 600; the communicator steps 120); a worker
that misses it, dies or reports an error makes the parent terminate all workers and return False: the caller runs the one-GPU
path.

Workers are spawned (never forked: the parent must not hand a HIP context down), one per device in MCALLER_SHARD_DEVICES
(default 0..n-1).  A worker takes jobs until it is told to end: with MCALLER_KEEP_WORKERS set the workers of one file stay for
the next (their HIP contexts, pinned buffers and table slots with them) -- what a caller that runs file after file wants, and
what bench.py's strong-scaling leg times as the warm runs.  `last_run` holds what the last run measured (per worker: rows,
seconds; the reduction: backend, milliseconds, bytes); mCaller.py writes it to $MCALLER_STATS_JSON.
"""
import multiprocessing
import multiprocessing.connection
import os
import sys
import time

import numpy as np

ROW_STRIDE = 1 << 40          # first-seen rows of the per-site reduction: piece number * ROW_STRIDE + row inside the piece


def _devices(n_gpus):
    env = os.environ.get('MCALLER_SHARD_DEVICES', '')
    if env:
        devs = [int(x) for x in env.split(',')]
        if len(devs) != n_gpus:
            raise ValueError('MCALLER_SHARD_DEVICES names %d devices for %d workers' % (len(devs), n_gpus))
        return devs
    return list(range(n_gpus))


def _peak_rss_mb(breakdown=False):
    """High-water mark of this process's resident memory in MB (breakdown: what it is made of right now -- pinned host memory
    counts as RssShmem)."""
    try:
        st = dict(l.split(':', 1) for l in open('/proc/self/status').read().splitlines() if ':' in l)
        mb = lambda key: int(st[key].split()[0]) / 1024.0 if key in st else None
        if breakdown:
            return {k: mb(k) for k in ('VmHWM', 'VmRSS', 'RssAnon', 'RssFile', 'RssShmem')}
        return mb('VmHWM')
    except (OSError, ValueError):
        return None


class _Abort(Exception):
    """The parent told the worker to stop."""


def _worker(conn, device, n_workers=1, index=0):
    """A worker process: one GPU, jobs from the parent until it says None (or its end of the pipe closes).  A plain run sends
    one job; a caller that runs file after file (MCALLER_KEEP_WORKERS: bench.py's strong-scaling leg) finds the worker, its HIP
    context, its pinned buffers and its table slots where the last file left them."""
    os.environ['MCALLER_DEVICE'] = str(device)
    cache = {}
    try:
        if n_workers > 1:                       # (the workers of a run share the host's cores -- and its CPU-time quota, if it has one)
            from . import _lib
            share = max(2, int(_lib.lib().mc_host_cores()) // n_workers)
            share = min(share, int(os.environ.get('MCALLER_HOST_CORES', share)))
            os.environ['MCALLER_HOST_CORES'] = str(share)
            # (workers whose GPUs hang off one NUMA node are bound to the same CPUs: each takes its own stretch of them)
            os.environ['MCALLER_HOST_CORE_OFFSET'] = str(index * share)
        while True:
            try:
                job = conn.recv()
            except (EOFError, OSError):
                return
            if job is None or not _job(conn, device, job, cache):
                return
    except BaseException as e:                                   # noqa
        try:
            import traceback
            conn.send(dict(error='%s: %s\n%s' % (type(e).__name__, e, traceback.format_exc(limit=6))))
        except Exception:
            pass
    finally:
        conn.close()


def _job(conn, device, job, cache):
    """One piece of one file.  -> True: done, the next job may come; False: the parent said stop."""
    import contextlib
    import io
    t_job = time.perf_counter()
    from . import _lib
    if job['world'] > 1 and not cache.get('bound'):
        _lib.lib().mc_bind_to_device_numa_node(int(device))     # reader threads and pinned buffers next to this worker's GPU
        cache['bound'] = True
    from . import extract_contexts as ec
    from . import make_bed
    from .device import get_device
    from .model_io import load_model_file
    read2qual = job['read2qual']
    if read2qual is None:                 # (every worker reads the FASTQ itself, natively: a pickled dict per worker costs more)
        key = ('fastq', job['fastq'], os.path.getmtime(job['fastq']), os.path.getsize(job['fastq']))
        if cache.get('fastq_key') != key:
            from .read_qual import extract_read_quality
            cache['fastq_key'], cache['read2qual'] = key, extract_read_quality(job['fastq'])
        read2qual = cache['read2qual']
    train = bool(job.get('train'))
    modelset = None if train else load_model_file(job['modelfile'])      # (train mode: features only, :131-134)
    dev = get_device()
    rank, k = job['rank'], job['k']
    state = dict(head_sent=False, go_seen=False, index=None, extras=[], t_reduce=0.0)
    t_ready = time.perf_counter()

    def on_head(name):
        state['head_sent'] = True
        conn.send(dict(head=name))

    def tail_of_last():
        go = conn.recv()
        state['go_seen'] = True
        if go is None:
            raise _Abort()
        return go['tail']

    def on_shard(P, rec, fin, tail, rows_before):
        """The per-site reduction of one shard's records (make_bed.py:86-96), on the device, added to the worker's counts.
        Records the host scored itself (NaN on the device) are folded in; records whose row names another contig than their
        site (R8: closed by a row of the next contig) are not sites of the numbering: they travel as `extras`."""
        t_r = time.perf_counter()
        if state['index'] is None:
            state['index'] = make_bed.SiteIndex(P.ref.meth, len(P.ref.names))
            dev.site_counts_reset()
        offset = rank * ROW_STRIDE + rows_before
        extras = make_bed.cross_contig_records(rec, P.table, P.ref, k, fin.host_scored, tail, row_offset=offset)
        tail_id = P.ref.names.index(tail) if tail is not None else -1
        if dev.site_counts_accumulate(row_offset=offset, tail_contig=tail_id):
            make_bed.add_pending_site_counts(dev, rec, P.table, state['index'], row_offset=offset, prob=fin.host_prob(rec),
                                             skip=extras['records'])
        state['extras'].extend(extras['rows'])
        state['t_reduce'] += time.perf_counter() - t_r

    # ---- steps 1 and 2: the piece, streamed ----
    buf = io.StringIO()
    try:
        with open(job['part'], 'wb') as out, contextlib.redirect_stdout(buf):
            res = ec.stream_features(job['tsv'], job['fasta'], read2qual, k, job['skip_thresh'], job['qual_thresh'], modelset,
                                     None, job['base'], job['motif'], job['positions_list'], byte_range=(job['lo'], job['hi']),
                                     sink=out.write, tail_of_last=tail_of_last, on_head=on_head,
                                     on_shard=on_shard if job['bed'] else None, mark_all=bool(job['bed']), min_shards=1,
                                     train=train, pos_label=job.get('pos_label'))
    except _Abort:
        return False
    except ec._Unstreamable as e:
        if not state['head_sent']:
            conn.send(dict(head=None))
        conn.send(dict(stop='unstreamable: %s' % e, stdout=buf.getvalue()))
        conn.recv()
        return False
    t_done = time.perf_counter()
    clock = getattr(ec.stream_features, 'last_clock', None) or {}
    conn.send(dict(stop=None, stdout=buf.getvalue(), messages=res.messages, head_names=list(res.head_names), tail_names=list(res.tail_names),
                   had_records=bool(res.had_records), n_rows=res.n_rows,
                   n_obs=res.n_obs, positions=res.positions, n_multi=res.n_multi, n_wskips=res.n_wskips,
                   n_skipped=res.n_skipped, n_bytes_out=res.n_bytes, signals=res.signals, contexts=res.contexts,
                   seconds=dict(total=t_done - t_job, setup=t_ready - t_job, stream=t_done - t_ready,
                                site_counts=state['t_reduce'], reader_threads=clock.get('parse', 0.0),
                                wait_for_table=clock.get('wait_parser', 0.0), enqueue=clock.get('enqueue', 0.0),
                                hand_out=clock.get('hand_out', 0.0)),
                   shards=clock.get('shards', 0), text_bytes=job['hi'] - job['lo'], peak_rss_mb=_peak_rss_mb(), rss_mb=_peak_rss_mb(True)))
    if not state['go_seen'] and conn.recv() is None:     # (a piece without a pass never asked for its tail: the answer is still there)
        return False
    if not job['bed']:
        return True
    # ---- steps 3-5: the per-site reduction; every step is the parent's decision for everybody ----
    if conn.recv() is None:
        return False
    if state['index'] is None:                # (a piece without a pass: counts of zero over the same site numbering)
        from .refmark import MarkedReference
        ref = MarkedReference(job['fasta'], job['base'], job['motif'], job['positions_list'])
        ref.quiet = True
        for cid in range(len(ref.names)):
            ref.mark(cid)
        dev.set_reference(ref.device_arrays())
        dev.site_counts_reset()
    # (kept workers keep their communicator: ncclCommInitRank is paid by the first file only)
    keep = bool(os.environ.get('MCALLER_KEEP_WORKERS'))
    have = cache.get('comm') == (job['world'], rank)
    uid, can = None, job['world'] > 1      # (one worker: nothing to exchange, and loading librccl.so costs seconds and ~11 GB of
    try:                                   #  transient host memory while the HIP runtime unpacks its code objects)
        if can:
            from .device import Device
            uid = Device.comm_unique_id() if rank == 0 else None   # (loads librccl.so; rank 0: ncclGetUniqueId)
            if rank != 0:
                Device.comm_probe()
    except Exception as e:                                      # noqa
        can, uid = False, None
    conn.send(dict(can=can, uid=uid, have=have))
    how = conn.recv()
    if how is None:
        return False
    up = False
    t_init = time.perf_counter()
    if how.get('reuse') and have:
        up = True
    elif how.get('init'):
        cache.pop('comm', None)
        try:
            dev.comm_init(job['world'], rank, how['uid'])      # (ncclCommInitRank: every worker was told to, every worker can)
            up = True
        except Exception:                                       # noqa
            up = False
    t_init = time.perf_counter() - t_init
    conn.send(dict(up=up))
    how = conn.recv()
    if how is None:
        return False
    if not how.get('rccl') and up:
        dev.comm_destroy()
        cache.pop('comm', None)
        up = False
    own = dev.site_counts_fetch()         # this worker's own counts first: what the parent adds up if the collective fails anywhere
    reduced, ms, err = None, 0.0, None
    if up:
        try:
            reduced = dev.site_allreduce()
            ms = reduced[3]
        except Exception as e:                                  # noqa
            err, reduced = str(e), None
        if keep and reduced is not None:
            cache['comm'] = (job['world'], rank)
        else:
            dev.comm_destroy()
            cache.pop('comm', None)
    lead = rank == 0
    conn.send(dict(bed=dict(own=own, reduced=reduced[:3] if (reduced is not None and lead) else None,
                            collective_done=reduced is not None, ms=ms, comm_init_ms=t_init * 1e3,
                            comm_reused=bool(how.get('rccl')) and have, err=err,
                            extras=state['extras'])))
    return True


class _Workers(object):
    """The worker processes and the parent's ends of their pipes; every wait has a deadline."""

    def __init__(self, ctx, devices):
        self.procs, self.conns, self.jobs, self.devices = [], [], [], list(devices)
        self.timeout = float(os.environ.get('MCALLER_WORKER_TIMEOUT', '600'))
        self.files = 0
        for dev in devices:
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, dev, len(self.devices), len(self.procs)))
            p.start()
            child.close()
            self.procs.append(p)
            self.conns.append(parent)

    def alive(self):
        return all(p.is_alive() for p in self.procs)

    def start(self, jobs):
        self.jobs = jobs
        self.files += 1
        self.tell(jobs)

    def gather(self, timeout=None, on_message=None):
        """One message from every worker -> list, or None if a worker died, reported an error or missed the deadline.
        on_message(i, msg): called as the messages arrive (the parent may have something to send meanwhile)."""
        out = [None] * len(self.conns)
        waiting = {c: i for i, c in enumerate(self.conns)}
        deadline = time.monotonic() + (self.timeout if timeout is None else timeout)
        while waiting:
            ready = multiprocessing.connection.wait(list(waiting), timeout=max(0.0, deadline - time.monotonic()))
            if not ready:
                return None                                      # deadline
            for c in ready:
                try:
                    msg = c.recv()
                except (EOFError, OSError):
                    return None                                  # the worker died
                if 'error' in msg:
                    sys.stderr.write('mcaller_amd worker %d: %s\n' % (waiting[c], msg['error']))
                    return None
                i = waiting.pop(c)
                out[i] = msg
                if on_message is not None:
                    on_message(i, msg)
        return out

    def tell(self, messages):
        for c, m in zip(self.conns, messages):
            try:
                c.send(m)
            except Exception:                                    # noqa
                pass

    def stop(self, remove_parts=True):
        """Abort: workers that wait for the parent get `None`, everything still alive after a moment is terminated."""
        global _kept
        if _kept is self:
            _kept = None
        self.tell([None] * len(self.conns))
        t_end = time.monotonic() + 5.0
        for p in self.procs:
            p.join(max(0.0, t_end - time.monotonic()))
        for p in self.procs:
            if p.is_alive():
                p.terminate()
                p.join(5.0)
        for c in self.conns:
            c.close()
        if remove_parts:
            for job in self.jobs:
                if os.path.exists(job['part']):
                    os.remove(job['part'])
        return False

    def release(self):
        """The file is done: the workers end (told so: a worker waits for its next job), or stay for the next file."""
        global _kept
        if os.environ.get('MCALLER_KEEP_WORKERS'):
            if _kept is None:
                import atexit
                atexit.register(_stop_kept)
            _kept = self
            return
        self.tell([None] * len(self.conns))
        for p in self.procs:
            p.join(self.timeout)
            if p.is_alive():
                p.terminate()
        for c in self.conns:
            c.close()


_kept = None          # the workers of the last file, with MCALLER_KEEP_WORKERS


def _stop_kept():
    if _kept is not None:
        _kept.stop(remove_parts=False)


def _workers_for(devices):
    """Workers for these devices: the ones kept from the last file (same devices, all alive) or new ones (spawned, never
    forked: the parent may hold a HIP context of its own and must not hand it down)."""
    global _kept
    if _kept is not None:
        W, _kept = _kept, None
        if W.devices == list(devices) and W.alive():
            return W
        W.stop(remove_parts=False)
    return _Workers(multiprocessing.get_context('spawn'), devices)


last_run = None       # what the last sharded run measured (mCaller.py writes it to $MCALLER_STATS_JSON)
train_dicts = None    # train mode: the (signals, contexts) the last sharded run collected
bed_written = False   # the last sharded run wrote the BED from the workers' reduction (else: the caller makes it from the rows)


def _join_parts(tsv_output, parts):
    """The workers' part files behind whatever `tsv_output` holds (the reference appends, extract_contexts.py:34-38), in file order,
    and the parts removed.  No byte goes through the interpreter: an output that does not exist yet (or is empty) BECOMES the first
    part (a rename -- with one worker that is the whole join), the others are copied inside the kernel (copy_file_range; a one-base
    motif leaves 1.3 GB of rows per 10^8 events: read() / write() in blocks took longer than the workers did)."""
    parts = list(parts)
    if parts and (not os.path.exists(tsv_output) or os.path.getsize(tsv_output) == 0):
        try:
            os.replace(parts[0], tsv_output)
            parts.pop(0)
        except OSError:                                            # (another file system: copied like the rest)
            pass
    if not parts:
        return
    out_fd = os.open(tsv_output, os.O_WRONLY | os.O_CREAT, 0o666)   # (not O_APPEND: copy_file_range refuses it)
    try:
        os.lseek(out_fd, 0, os.SEEK_END)
        for path in parts:
            with open(path, 'rb') as part:
                left = os.fstat(part.fileno()).st_size
                in_kernel = hasattr(os, 'copy_file_range')
                while left > 0:
                    n = 0
                    if in_kernel:
                        try:
                            n = os.copy_file_range(part.fileno(), out_fd, min(left, 1 << 30))
                        except OSError:                            # (EXDEV / ENOSYS / EINVAL: this pair of files the plain way)
                            in_kernel = False
                            continue
                        if n == 0:
                            in_kernel = False
                            continue
                    else:
                        block = part.read(min(left, 64 << 20))
                        if not block:
                            break
                        view, n = memoryview(block), len(block)
                        while len(view):
                            view = view[os.write(out_fd, view):]
                    left -= n
            os.remove(path)
    finally:
        os.close(out_fd)


def extract_features_sharded(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile, base, motif,
                             positions_list, n_gpus, bed=None, fastq=None, train=False, pos_label=None):
    """One file on n_gpus GPUs.  Returns True when the `.diffs.<k>[.train].tmp0` file has been written and the counter lines
    printed; False when the file cannot be cut (a read name in two pieces, an exit path of the reference, an error or a
    missed deadline in a worker): the caller then runs the one-GPU path, which reproduces the reference's behaviour in
    those cases.  fastq: the workers read the qualities themselves (else `read2qual` is shipped to each).  train (the reference
    fans train-mode extraction out over its `-t` processes too, mCaller.py:72-87): the workers return their pieces of the
    (signals, contexts) dicts, merged here in file order -> `train_dicts`."""
    global last_run
    from . import _lib
    t_start = time.perf_counter()
    last_run = None
    # the bytes the reference's loop reads for (0, file size): the last < 500 bytes of a file can stay unread (:141-148)
    lo, hi = _lib.eventalign_consumed_range(tsv_input, 0, os.path.getsize(tsv_input))
    cuts = _lib.eventalign_read_cuts(tsv_input, n_gpus, lo, hi)
    devices = _devices(n_gpus)
    stem = '.'.join(tsv_input.split('.')[:-1])
    tsv_output = stem + '.diffs.' + str(k) + ('.train' if train else '') + '.tmp0'
    jobs = [dict(tsv=tsv_input, fasta=fasta_input, read2qual=None if fastq else read2qual, fastq=fastq, lo=cuts[r], hi=cuts[r + 1],
                 base=base, motif=motif, positions_list=positions_list, k=k, skip_thresh=skip_thresh, qual_thresh=qual_thresh,
                 modelfile=modelfile, part='%s.diffs.%d.part%d' % (stem, k, r), bed=bool(bed), rank=r, world=n_gpus, train=bool(train),
                 pos_label=pos_label if train else None)
            for r in range(n_gpus)]
    W = _workers_for(devices)
    reused = W.files > 0
    t_cut = time.perf_counter()
    marking = None
    try:
        W.start(jobs)
        if bed:                               # (the site numbering of the BED, made while the workers stream)
            import threading
            marking = {}
            marking['thread'] = threading.Thread(target=_mark_for_bed, args=(marking, fasta_input, base, motif, positions_list), daemon=True)
            marking['thread'].start()
        # ---- step 1: the heads, as they come; go(tail) to everybody once all are known ----
        heads = W.gather()
        if heads is None:
            return W.stop()
        W.tell([dict(tail=next((h['head'] for h in heads[r + 1:] if h['head'] is not None), None)) for r in range(n_gpus)])
        # ---- step 2 ----
        results = W.gather()
        if results is None or any(x.get('stop') for x in results):
            return W.stop()
        # `last_read` across the cuts between the pieces (extract_contexts.cut_names): a name on both sides of one -- among the
        # reads behind it up to the first with a flush record and the reads in front of it from the last with one -- and the
        # file is not cut.  A name that comes back anywhere else changes nothing: the pieces stand.
        tail = set()
        for x in results:
            if tail.intersection(x['head_names']):
                return W.stop()
            tail = set(x['tail_names']) if x['had_records'] else tail | set(x['tail_names'])
        t_streamed = time.perf_counter()
        beds, reduction_failed = None, None
        if bed:
            # ---- steps 3-5: one decision for everybody, three times.  The rows are done by now: if a step of the reduction does not
            # finish (a communicator that never comes up, a worker that dies in it) the workers are stopped, the parts are kept, and
            # the caller makes the BED from the rows -- what make_bed.py does ----
            quick = min(W.timeout, float(os.environ.get('MCALLER_COMM_TIMEOUT', '120')))

            def reduction_steps():
                W.tell([dict(probe=True)] * n_gpus)
                probes = W.gather(quick)
                if probes is None:
                    return None, 'a worker did not answer the probe'
                reuse = n_gpus > 1 and all(p.get('have') for p in probes)       # (kept workers whose communicator is still up)
                init = reuse or (n_gpus > 1 and all(p['can'] for p in probes) and probes[0]['uid'] is not None)
                W.tell([dict(init=init and not reuse, reuse=reuse, uid=probes[0]['uid'])] * n_gpus)
                ups = W.gather(quick)
                if ups is None:
                    return None, 'the communicator did not come up within %.0f s' % quick
                W.tell([dict(rccl=init and all(u['up'] for u in ups))] * n_gpus)
                third = W.gather(quick)
                if third is None:
                    return None, 'the all-reduce did not finish within %.0f s' % quick
                return [x['bed'] for x in third], None
            beds, reduction_failed = reduction_steps()
        t_reduced = time.perf_counter()
        if reduction_failed:
            W.stop(remove_parts=False)
        else:
            W.release()
    except BaseException:
        W.stop()
        raise
    for x in results:
        for line in x['messages']:
            print(line)                                           # 'could not find sequence' lines, in file order
    _join_parts(tsv_output, [job['part'] for job in jobs])
    t_joined = time.perf_counter()
    global bed_written
    reduction, bed_written = None, False
    if bed and beds is not None:
        reduction = _write_bed(bed, beds, fasta_input, base, motif, positions_list, k, marking)
        reduction['seconds_steps_3_to_5'] = t_reduced - t_streamed
        bed_written = True
    elif bed:
        sys.stderr.write('mcaller_amd: the per-site reduction did not finish (%s): the BED is made from the rows\n' % reduction_failed)
        reduction = dict(backend='rows (%s)' % reduction_failed, ms=None, comm_init_ms=None, comm_reused=None, bytes=0, sites=None,
                         observations=None, loci_written=None, cross_contig_rows=None, seconds_steps_3_to_5=t_reduced - t_streamed)
    positions = np.unique(np.concatenate([x['positions'] for x in results])) if results else np.zeros(0)
    print('thread finished processing...:')
    print('%d observations' % sum(x['n_obs'] for x in results))
    print('%d positions' % len(positions))
    print('%d regions with multiple methylated bases' % sum(x['n_multi'] for x in results))
    print('%d observations with skips included' % sum(x['n_wskips'] for x in results))
    print('%d observations with too many skips' % sum(x['n_skipped'] for x in results))
    global train_dicts
    train_dicts = None
    if train:                                  # the pieces' dicts, in file order (:133-134: one entry per sub-model key, lists per label)
        signals, contexts = {}, {}
        for x in results:
            for merged, piece in ((signals, x['signals'] or {}), (contexts, x['contexts'] or {})):
                for key, by_label in piece.items():
                    into = merged.setdefault(key, {})
                    for label, rows in by_label.items():
                        into.setdefault(label, []).extend(rows)
        train_dicts = (signals, contexts)
    last_run = dict(n_gpus=n_gpus, devices=devices, workers_reused=reused, rows=sum(x['n_rows'] for x in results),
                    observations=sum(x['n_obs'] for x in results), text_bytes=hi - lo,
                    seconds=dict(total=time.perf_counter() - t_start, cut_and_start=t_cut - t_start, streamed=t_streamed - t_cut,
                                 reduction=t_reduced - t_streamed, parts_joined=t_joined - t_reduced,
                                 bed_written=time.perf_counter() - t_joined),
                    workers=[dict(rank=r, device=devices[r], rows=x['n_rows'], text_bytes=x['text_bytes'], shards=x['shards'],
                                  bytes_out=x['n_bytes_out'], peak_rss_mb=x['peak_rss_mb'], rss_mb=x.get('rss_mb'), seconds=x['seconds'])
                             for r, x in enumerate(results)],
                    site_reduction=reduction)
    return True


def combine_site_counts(beds):
    """The node-wide per-site counts from what the workers report: rank 0's all-reduced counts if EVERY worker finished the
    collective, else the sum (min for the first-seen rows) of the workers' own counts.  -> (n_meth, n_total, first, how)."""
    if beds and all(b['collective_done'] for b in beds) and beds[0]['reduced'] is not None:
        n_meth, n_total, first = beds[0]['reduced']
        return n_meth, n_total, first, 'ncclAllReduce over %d GPUs' % len(beds)
    n_meth = sum(np.asarray(b['own'][0], dtype=np.int64) for b in beds).astype(np.int32)
    n_total = sum(np.asarray(b['own'][1], dtype=np.int64) for b in beds).astype(np.int32)
    first = np.minimum.reduce([np.asarray(b['own'][2]) for b in beds])
    why = next((b['err'] for b in beds if b.get('err')), None) or ('one worker' if len(beds) == 1 else 'no RCCL communicator')
    return n_meth, n_total, first, 'summed on the host (%s)' % why


def _mark_for_bed(box, fasta_input, base, motif, positions_list):
    """Every contig marked and the sites numbered (the key space of the reduction): what the BED writer needs of the reference."""
    try:
        from . import make_bed
        from .refmark import MarkedReference
        ref = MarkedReference(fasta_input, base, motif, positions_list)
        ref.quiet = True
        for cid in range(len(ref.names)):
            try:
                ref.mark(cid)
            except SystemExit:
                pass
        box['ref'], box['index'] = ref, make_bed.SiteIndex(ref.meth, len(ref.names))
    except BaseException as e:                                   # noqa
        box['error'] = e


def _write_bed(bed, beds, fasta_input, base, motif, positions_list, k, marking=None):
    """BED of the whole file from the workers' reductions."""
    from . import make_bed
    if marking is None:
        marking = {}
        _mark_for_bed(marking, fasta_input, base, motif, positions_list)
    else:
        marking['thread'].join()
    if 'error' in marking:
        raise marking['error']
    ref, index = marking['ref'], marking['index']
    n_meth, n_total, first, how = combine_site_counts(beds)
    extras = [row for b in beds for row in b['extras']]
    count = make_bed.write_bed_from_counts(bed['path'], n_meth, n_total, first, index, ref.names, ref.meth, k,
                                           bed['min_depth'], bed['mod_threshold'], extras=extras)
    print(count, 'methylated loci found with min depth', bed['min_depth'], 'reads')
    print('per-site reduction: %s' % how)
    rccl = how.startswith('ncclAllReduce')
    return dict(backend=how, ms=max(b['ms'] for b in beds) if rccl else None,
                comm_init_ms=max(b.get('comm_init_ms', 0.0) for b in beds) if rccl else None,
                comm_reused=all(b.get('comm_reused') for b in beds) if rccl else None,
                bytes=int(index.n) * 16, sites=int(index.n), observations=int(np.asarray(n_total, dtype=np.int64).sum()),
                loci_written=int(count), cross_contig_rows=len(extras))
