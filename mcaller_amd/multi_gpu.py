"""One eventalign file on several GPUs of a node: reads shard embarrassingly.

The file is cut at the first lines of reads into one piece per GPU (`mc_eventalign_read_cuts`); one worker process per GPU
parses its piece, runs the HIP path on it and writes its rows; the parent concatenates the pieces in file order, which is
the reference's `-t 1` order (extract_contexts.py:179,242: a window never spans two reads).  Two things cross a cut and
are exchanged through the parent before the kernels run: the first unfiltered row after a piece closes that piece's last
window (R6) and supplies its `chrom` column (R8) -> `tail`; and `last_read`, which only matters when a read name occurs
in two pieces -> then the file is not cut at all (the caller falls back to one GPU).  No collective is needed for the
`.diffs` file; the per-site reduction feeding make_bed is the one exchange step (mc_site_allreduce, RCCL).

Workers are spawned (never forked: the parent must not hold a HIP context), one per device in MCALLER_SHARD_DEVICES
(default 0..n-1).
"""
import multiprocessing
import os
import sys

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
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            P = ec.prepare(job['tsv'], job['fasta'], job['read2qual'], job['lo'], job['hi'], job['base'], job['motif'],
                           job['positions_list'], exact_range=True)
        t = P.table
        if job['bed']:                       # the site numbering of the reduction must be the same on every worker:
            for cid in range(len(P.ref.names)):   # mark every contig, not only the ones this piece touches
                P.ref.mark(cid)
        # the first unfiltered row of this piece: it closes the previous piece's last window
        head = None
        for seg in range(t.n_seg):
            if P.qual[t.seg_read[seg]] < job['qual_thresh']:
                continue
            r0, r1 = int(t.seg_row_begin[seg]), int(t.seg_row_begin[seg + 1])
            ok = (t.flags[r0:r1] & _lib.F_MODEL_N) == 0
            if ok.any():
                head = P.ref.names[int(t.seg_contig[seg])]
                break
        uid = None
        if job['bed'] and job['rank'] == 0 and job['world'] > 1:
            try:
                from .device import Device
                uid = Device.comm_unique_id()                      # ncclGetUniqueId: shipped to the other workers by the parent
            except Exception:
                uid = None
        conn.send(dict(names=list(t.read_names), head=head, n_rows=t.n_rows, fatal=repr(P.fatal) if P.fatal is not None else None,
                       stdout=buf.getvalue(), uid=uid))
        go = conn.recv()
        if go is None:
            return
        tail = go['tail']
        modelset = load_model_file(job['modelfile'])
        tail_id = P.ref.names.index(tail) if tail is not None else -1
        rec = ec.compute(P, job['k'], job['skip_thresh'], job['qual_thresh'], modelset, job['base'], False, tail_contig=tail_id)
        fin = ec.Finisher(P, job['k'], job['base'], False, modelset=modelset, tail_chrom=tail)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            stop = fin.run(rec)
        if stop is not None:
            conn.send(dict(stop=repr(stop), stdout=buf.getvalue()))
            return
        with open(job['part'], 'wb') as out:
            out.write(fin.text())
        bed = None
        if job['bed']:
            bed = _reduce_sites(job, go, P, rec)
        info = rec.info[:rec.n]
        too = (info & _lib.I_TOO_MANY) != 0
        fin._count(rec.n)
        conn.send(dict(stop=None, stdout=buf.getvalue(), n_obs=fin.num_observations,
                       positions=np.unique(rec.site_pos[:rec.n][~too]), n_multi=fin._n_multi, n_wskips=fin._n_wskips,
                       n_skipped=fin._n_skipped, bed=bed))
    except BaseException as e:                                   # noqa
        try:
            conn.send(dict(error='%s: %s' % (type(e).__name__, e)))
        except Exception:
            pass
    finally:
        conn.close()


def _reduce_sites(job, go, P, rec):
    """The per-site reduction of this worker's records (make_bed.py:86-96): on the device, all-reduced over the workers
    with RCCL (mc_site_allreduce) -- rank 0 then holds the node-wide counts; if the communicator cannot be set up (e.g.
    several workers sharing one GPU) or the host scored some records itself, the worker's own counts go to the parent,
    which adds them up."""
    from . import make_bed
    from .device import get_device
    index = make_bed.SiteIndex(P.ref.meth, len(P.ref.names))
    dev = get_device()
    offset = go['row_offset']
    try:
        if job['world'] > 1:
            if go.get('uid') is None:
                raise RuntimeError('no RCCL unique id')
            dev.comm_init(job['world'], job['rank'], go['uid'])
        if dev.site_counts(row_offset=offset):
            raise RuntimeError('records scored on the host')
        n_meth, n_total, first, ms = dev.site_allreduce()
        dev.comm_destroy()
        return dict(mode='rccl', n_meth=n_meth if job['rank'] == 0 else None, n_total=n_total if job['rank'] == 0 else None,
                    first=first if job['rank'] == 0 else None, ms=ms)
    except Exception as e:                                       # noqa
        n_meth, n_total, first = make_bed.site_counts(rec, P.table, index, row_offset=offset)
        return dict(mode='host', why=str(e), n_meth=n_meth, n_total=n_total, first=first)


def extract_features_sharded(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile, base, motif,
                             positions_list, n_gpus, bed=None):
    """Predict mode on n_gpus GPUs.  Returns True when the `.diffs.<k>.tmp0` file has been written and the counter lines
    printed; False when the file cannot be cut (a read name in two pieces, an exit path of the reference, an error in a
    worker): the caller then runs the one-GPU path, which reproduces the reference's behaviour in those cases."""
    from . import _lib
    cuts = _lib.eventalign_read_cuts(tsv_input, n_gpus)
    devices = _devices(n_gpus)
    stem = '.'.join(tsv_input.split('.')[:-1])
    tsv_output = stem + '.diffs.' + str(k) + '.tmp0'
    ctx = multiprocessing.get_context('spawn')
    workers = []
    for r in range(n_gpus):
        parent, child = ctx.Pipe()
        job = dict(tsv=tsv_input, fasta=fasta_input, read2qual=read2qual, lo=cuts[r], hi=cuts[r + 1], base=base, motif=motif,
                   positions_list=positions_list, k=k, skip_thresh=skip_thresh, qual_thresh=qual_thresh, modelfile=modelfile,
                   part='%s.diffs.%d.part%d' % (stem, k, r), bed=bool(bed), rank=r, world=n_gpus)
        p = ctx.Process(target=_worker, args=(child, devices[r], job))
        p.start()
        child.close()
        workers.append((p, parent, job))

    def abort():
        for p, conn, job in workers:
            try:
                conn.send(None)
            except Exception:
                pass
        for p, conn, job in workers:
            p.join()
            if os.path.exists(job['part']):
                os.remove(job['part'])
        return False

    heads = []
    for p, conn, job in workers:
        try:
            heads.append(conn.recv())
        except EOFError:
            heads.append(dict(error='worker died'))
    if any('error' in h or h.get('fatal') for h in heads):
        return abort()
    seen = set()
    for h in heads:
        if seen.intersection(h['names']):
            return abort()                                       # a read name in two pieces: `last_read` crosses the cut
        seen.update(h['names'])
    for r, (p, conn, job) in enumerate(workers):
        tail = None
        for h in heads[r + 1:]:
            if h['head'] is not None:
                tail = h['head']
                break
        conn.send(dict(tail=tail, row_offset=sum(h['n_rows'] for h in heads[:r]), uid=heads[0].get('uid')))
    results = []
    for p, conn, job in workers:
        try:
            results.append(conn.recv())
        except EOFError:
            results.append(dict(error='worker died'))
    for p, conn, job in workers:
        p.join()
    if any('error' in x or x.get('stop') for x in results):
        for p, conn, job in workers:
            if os.path.exists(job['part']):
                os.remove(job['part'])
        return False
    for h in heads:
        sys.stdout.write(h['stdout'])                             # 'could not find sequence' lines, in file order
    with open(tsv_output, 'ab') as out:
        for p, conn, job in workers:
            with open(job['part'], 'rb') as part:
                out.write(part.read())
            os.remove(job['part'])
    if bed:
        _write_bed(bed, results, fasta_input, base, motif, positions_list, k)
    positions = np.unique(np.concatenate([x['positions'] for x in results])) if results else np.zeros(0)
    print('thread finished processing...:')
    print('%d observations' % sum(x['n_obs'] for x in results))
    print('%d positions' % len(positions))
    print('%d regions with multiple methylated bases' % sum(x['n_multi'] for x in results))
    print('%d observations with skips included' % sum(x['n_wskips'] for x in results))
    print('%d observations with too many skips' % sum(x['n_skipped'] for x in results))
    return True


def _write_bed(bed, results, fasta_input, base, motif, positions_list, k):
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
    beds = [x['bed'] for x in results]
    if all(b['mode'] == 'rccl' for b in beds):
        n_meth, n_total, first = beds[0]['n_meth'], beds[0]['n_total'], beds[0]['first']
    elif all(b['mode'] == 'host' for b in beds):
        n_meth = sum(b['n_meth'] for b in beds)
        n_total = sum(b['n_total'] for b in beds)
        first = np.minimum.reduce([b['first'] for b in beds])
    else:
        raise RuntimeError('workers disagree on how the per-site counts were reduced')
    count = make_bed.write_bed_from_counts(bed['path'], n_meth, n_total, first, index, ref.names, ref.meth, k,
                                           bed['min_depth'], bed['mod_threshold'])
    print(count, 'methylated loci found with min depth', bed['min_depth'], 'reads')
    print('per-site reduction: %s' % ('ncclAllReduce over %d GPUs' % len(beds) if beds[0]['mode'] == 'rccl'
                                      else 'summed on the host (%s)' % beds[0].get('why', '')))
