"""One eventalign file on several GPUs of a node: reads shard embarrassingly.

The byte range the reference's loop consumes (extract_contexts.py:141-148) is cut at the first lines of reads into one
piece per GPU (`mc_eventalign_read_cuts_range`); one worker process per GPU parses its piece, runs the HIP path on it and
writes its rows; the parent concatenates the pieces in file order, which is the reference's `-t 1` order
(extract_contexts.py:179,242: a window never spans two reads).  Two things cross a cut and are exchanged through the parent
before the kernels run: the first unfiltered row after a piece closes that piece's last window (R6) and supplies its `chrom`
column (R8) -> `tail`; and `last_read`, which only matters when a read name occurs in two pieces -> then the file is not
cut at all (the caller falls back to one GPU).  No collective is needed for the `.diffs` file; the per-site reduction
feeding make_bed is the one exchange step (mc_site_allreduce, RCCL).

The protocol between the parent and its workers has three rounds, and every round is decided by the parent for ALL workers
at once -- a worker never enters a collective (ncclCommInitRank, ncclAllReduce) that another worker may not reach:

  1. worker -> parent: read names, head contig, rows, `fatal`            parent -> worker: go(tail, row offset) | abort
  2. worker -> parent: rows written, or the reference's exit path        parent -> worker: reduce('rccl' | 'host') | abort
  3. worker -> parent: per-site counts (rank 0 holds the all-reduced ones)

Every wait of the parent has a deadline (MCALLER_WORKER_TIMEOUT seconds, default 600); a worker that misses it, dies or
reports an error makes the parent terminate all workers and return False: the caller runs the one-GPU path.

Workers are spawned (never forked: the parent must not hold a HIP context), one per device in MCALLER_SHARD_DEVICES
(default 0..n-1).
"""
import multiprocessing
import multiprocessing.connection
import os
import sys
import time

import numpy as np


def _devices(n_gpus):
    env = os.environ.get('MCALLER_SHARD_DEVICES', '')
    if env:
        devs = [int(x) for x in env.split(',')]
        if len(devs) != n_gpus:
            raise ValueError('MCALLER_SHARD_DEVICES names %d devices for %d workers' % (len(devs), n_gpus))
        return devs
    return list(range(n_gpus))


def _worker(conn, device, job):
    import contextlib
    import io
    os.environ['MCALLER_DEVICE'] = str(device)
    try:
        from . import _lib
        if job['world'] > 1:
            _lib.lib().mc_bind_to_device_numa_node(int(device))     # parser threads and pinned buffers next to this worker's GPU
        from . import extract_contexts as ec
        from .model_io import load_model_file
        # ---- round 1: parse, report what crosses the cuts ----
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            P = ec.prepare(job['tsv'], job['fasta'], job['read2qual'], job['lo'], job['hi'], job['base'], job['motif'],
                           job['positions_list'], exact_range=True)
        t = P.table
        if job['bed']:                       # the site numbering of the reduction must be the same on every worker:
            for cid in range(len(P.ref.names)):   # mark every contig, not only the ones this piece touches
                P.ref.mark(cid)
        head = ec.head_contig(P, job['qual_thresh'])              # its first unfiltered row closes the previous piece's last window
        uid = None
        if job['bed'] and job['rank'] == 0 and job['world'] > 1:
            try:
                from .device import Device
                uid = Device.comm_unique_id()                      # ncclGetUniqueId: shipped to the other workers by the parent
            except Exception:
                uid = None
        conn.send(dict(names=list(t.read_names), head=None if head is None else P.ref.names[head], n_rows=t.n_rows,
                       fatal=repr(P.fatal) if P.fatal is not None else None, stdout=buf.getvalue(), uid=uid))
        go = conn.recv()
        if go is None:
            return
        # ---- round 2: the kernels, the rows ----
        tail = go['tail']
        modelset = load_model_file(job['modelfile'])
        tail_id = P.ref.names.index(tail) if tail is not None else -1
        rec = ec.compute(P, job['k'], job['skip_thresh'], job['qual_thresh'], modelset, job['base'], False, tail_contig=tail_id)
        fin = ec.Finisher(P, job['k'], job['base'], False, modelset=modelset, tail_chrom=tail)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            stop = fin.run(rec)
        if stop is None:
            with open(job['part'], 'wb') as out:
                out.write(fin.text())
            info = rec.info[:rec.n]
            too = (info & _lib.I_TOO_MANY) != 0
            fin._count(rec.n)
            conn.send(dict(stop=None, stdout=buf.getvalue(), n_obs=fin.num_observations,
                           positions=np.unique(rec.site_pos[:rec.n][~too]), n_multi=fin._n_multi, n_wskips=fin._n_wskips,
                           n_skipped=fin._n_skipped))
        else:
            conn.send(dict(stop=repr(stop), stdout=buf.getvalue()))
        if not job['bed']:
            return
        how = conn.recv()
        if how is None:
            return
        # ---- round 3: the per-site reduction; every worker is here, and every worker was told the same `how` ----
        conn.send(dict(bed=_reduce_sites(job, go, how, P, rec, fin)))
    except BaseException as e:                                   # noqa
        try:
            conn.send(dict(error='%s: %s' % (type(e).__name__, e)))
        except Exception:
            pass
    finally:
        conn.close()


def _reduce_sites(job, go, how, P, rec, fin):
    """The per-site reduction of this worker's records (make_bed.py:86-96): on the device, all-reduced over the workers
    with RCCL (mc_site_allreduce) -- rank 0 then holds the node-wide counts.  Records the host scored itself (NaN on the
    device) are folded into the device-side counts first.  If the communicator cannot be set up (e.g. several workers
    sharing one GPU: RCCL refuses on every rank alike) the worker's own counts go to the parent, which adds them up.
    Records whose row names another contig than their site (R8: closed by a row of the next contig) are not sites of the
    numbering: they travel as `extras` and the parent adds them."""
    from . import make_bed
    from .device import get_device
    index = make_bed.SiteIndex(P.ref.meth, len(P.ref.names))
    dev = get_device()
    offset = go['row_offset']
    extras = make_bed.cross_contig_records(rec, P.table, P.ref, job['k'], fin.host_scored, fin.tail_chrom, row_offset=offset)
    if how == 'rccl':
        try:
            if job['world'] > 1:
                dev.comm_init(job['world'], job['rank'], go['uid'])
            if dev.site_counts(row_offset=offset, tail_contig=go['tail_id']):
                make_bed.add_pending_site_counts(dev, rec, P.table, index, row_offset=offset, prob=fin.host_prob(rec),
                                                 skip=extras['records'])
            n_meth, n_total, first, ms = dev.site_allreduce()
            dev.comm_destroy()
            lead = job['rank'] == 0
            return dict(mode='rccl', n_meth=n_meth if lead else None, n_total=n_total if lead else None,
                        first=first if lead else None, ms=ms, extras=extras['rows'])
        except Exception as e:                                   # noqa
            why = str(e)
    else:
        why = 'no RCCL communicator'
    n_meth, n_total, first = make_bed.site_counts(rec, P.table, index, row_offset=offset, prob=fin.host_prob(rec),
                                                  skip=extras['records'])
    return dict(mode='host', why=why, n_meth=n_meth, n_total=n_total, first=first, extras=extras['rows'])


class _Workers(object):
    """The worker processes and the parent's ends of their pipes; every wait has a deadline."""

    def __init__(self, ctx, devices, jobs):
        self.procs, self.conns, self.jobs = [], [], jobs
        self.timeout = float(os.environ.get('MCALLER_WORKER_TIMEOUT', '600'))
        for dev, job in zip(devices, jobs):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, dev, job))
            p.start()
            child.close()
            self.procs.append(p)
            self.conns.append(parent)

    def gather(self):
        """One message from every worker -> list, or None if a worker died, reported an error or missed the deadline."""
        out = [None] * len(self.conns)
        waiting = {c: i for i, c in enumerate(self.conns)}
        deadline = time.monotonic() + self.timeout
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
                out[waiting.pop(c)] = msg
        return out

    def tell(self, messages):
        for c, m in zip(self.conns, messages):
            try:
                c.send(m)
            except Exception:                                    # noqa
                pass

    def stop(self, remove_parts=True):
        """Abort: workers that wait for the parent get `None`, everything still alive after a moment is terminated."""
        self.tell([None] * len(self.conns))
        t_end = time.monotonic() + 5.0
        for p in self.procs:
            p.join(max(0.0, t_end - time.monotonic()))
        for p in self.procs:
            if p.is_alive():
                p.terminate()
                p.join(5.0)
        if remove_parts:
            for job in self.jobs:
                if os.path.exists(job['part']):
                    os.remove(job['part'])
        return False

    def join(self):
        for p in self.procs:
            p.join(self.timeout)
            if p.is_alive():
                p.terminate()


def extract_features_sharded(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile, base, motif,
                             positions_list, n_gpus, bed=None):
    """Predict mode on n_gpus GPUs.  Returns True when the `.diffs.<k>.tmp0` file has been written and the counter lines
    printed; False when the file cannot be cut (a read name in two pieces, an exit path of the reference, an error or a
    missed deadline in a worker): the caller then runs the one-GPU path, which reproduces the reference's behaviour in
    those cases."""
    from . import _lib
    # the bytes the reference's loop reads for (0, file size): the last < 500 bytes of a file can stay unread (:141-148)
    lo, hi = _lib.eventalign_consumed_range(tsv_input, 0, os.path.getsize(tsv_input))
    cuts = _lib.eventalign_read_cuts(tsv_input, n_gpus, lo, hi)
    devices = _devices(n_gpus)
    stem = '.'.join(tsv_input.split('.')[:-1])
    tsv_output = stem + '.diffs.' + str(k) + '.tmp0'
    ctx = multiprocessing.get_context('spawn')
    jobs = [dict(tsv=tsv_input, fasta=fasta_input, read2qual=read2qual, lo=cuts[r], hi=cuts[r + 1], base=base, motif=motif,
                 positions_list=positions_list, k=k, skip_thresh=skip_thresh, qual_thresh=qual_thresh, modelfile=modelfile,
                 part='%s.diffs.%d.part%d' % (stem, k, r), bed=bool(bed), rank=r, world=n_gpus) for r in range(n_gpus)]
    W = _Workers(ctx, devices, jobs)
    try:
        # ---- round 1 ----
        heads = W.gather()
        if heads is None or any(h.get('fatal') for h in heads):
            return W.stop()
        seen = set()
        for h in heads:
            if seen.intersection(h['names']):
                return W.stop()                                  # a read name in two pieces: `last_read` crosses the cut
            seen.update(h['names'])
        names = None
        go = []
        for r in range(n_gpus):
            tail = next((h['head'] for h in heads[r + 1:] if h['head'] is not None), None)
            go.append(dict(tail=tail, row_offset=sum(h['n_rows'] for h in heads[:r]), uid=heads[0].get('uid'), tail_id=-1))
        if bed:                                                  # (contig ids for the device-side reduction)
            from .refmark import read_fasta
            names = [n for n, _ in read_fasta(fasta_input)]
            for g in go:
                g['tail_id'] = names.index(g['tail']) if g['tail'] is not None else -1
        W.tell(go)
        # ---- round 2 ----
        results = W.gather()
        if results is None or any(x.get('stop') for x in results):
            return W.stop()
        beds = None
        if bed:
            # ---- round 3: one decision for everybody ----
            how = 'rccl' if (n_gpus == 1 or heads[0].get('uid') is not None) else 'host'
            W.tell([how] * n_gpus)
            third = W.gather()
            if third is None:
                return W.stop()
            beds = [x['bed'] for x in third]
        W.join()
    except BaseException:
        W.stop()
        raise
    for h in heads:
        sys.stdout.write(h['stdout'])                             # 'could not find sequence' lines, in file order
    with open(tsv_output, 'ab') as out:
        for job in jobs:
            with open(job['part'], 'rb') as part:
                out.write(part.read())
            os.remove(job['part'])
    if bed:
        _write_bed(bed, beds, fasta_input, base, motif, positions_list, k)
    positions = np.unique(np.concatenate([x['positions'] for x in results])) if results else np.zeros(0)
    print('thread finished processing...:')
    print('%d observations' % sum(x['n_obs'] for x in results))
    print('%d positions' % len(positions))
    print('%d regions with multiple methylated bases' % sum(x['n_multi'] for x in results))
    print('%d observations with skips included' % sum(x['n_wskips'] for x in results))
    print('%d observations with too many skips' % sum(x['n_skipped'] for x in results))
    return True


def _write_bed(bed, beds, fasta_input, base, motif, positions_list, k):
    """BED of the whole file from the workers' reductions: rank 0's all-reduced counts, or the sum of per-worker counts."""
    from . import make_bed
    from .refmark import MarkedReference
    ref = MarkedReference(fasta_input, base, motif, positions_list)
    for cid in range(len(ref.names)):
        try:
            ref.mark(cid)
        except SystemExit:
            pass
    index = make_bed.SiteIndex(ref.meth, len(ref.names))
    if all(b['mode'] == 'rccl' for b in beds):
        n_meth, n_total, first = beds[0]['n_meth'], beds[0]['n_total'], beds[0]['first']
    elif all(b['mode'] == 'host' for b in beds):
        n_meth = sum(b['n_meth'] for b in beds)
        n_total = sum(b['n_total'] for b in beds)
        first = np.minimum.reduce([b['first'] for b in beds])
    else:
        raise RuntimeError('workers disagree on how the per-site counts were reduced')
    extras = [row for b in beds for row in b['extras']]
    count = make_bed.write_bed_from_counts(bed['path'], n_meth, n_total, first, index, ref.names, ref.meth, k,
                                           bed['min_depth'], bed['mod_threshold'], extras=extras)
    print(count, 'methylated loci found with min depth', bed['min_depth'], 'reads')
    print('per-site reduction: %s' % ('ncclAllReduce over %d GPUs' % len(beds) if beds[0]['mode'] == 'rccl'
                                      else 'summed on the host (%s)' % beds[0].get('why', '')))
