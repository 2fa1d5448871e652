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
        from . import extract_contexts as ec
        from .model_io import load_model_file
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            P = ec.prepare(job['tsv'], job['fasta'], job['read2qual'], job['lo'], job['hi'], job['base'], job['motif'],
                           job['positions_list'], exact_range=True)
        t = P.table
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
        conn.send(dict(names=list(t.read_names), head=head, n_rows=t.n_rows, fatal=repr(P.fatal) if P.fatal is not None else None,
                       stdout=buf.getvalue()))
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
        info = rec.info[:rec.n]
        too = (info & _lib.I_TOO_MANY) != 0
        fin._count(rec.n)
        conn.send(dict(stop=None, stdout=buf.getvalue(), n_obs=fin.num_observations,
                       positions=np.unique(rec.site_pos[:rec.n][~too]), n_multi=fin._n_multi, n_wskips=fin._n_wskips,
                       n_skipped=fin._n_skipped))
    except BaseException as e:                                   # noqa
        try:
            conn.send(dict(error='%s: %s' % (type(e).__name__, e)))
        except Exception:
            pass
    finally:
        conn.close()


def extract_features_sharded(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile, base, motif,
                             positions_list, n_gpus):
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
                   part='%s.diffs.%d.part%d' % (stem, k, r))
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
        conn.send(dict(tail=tail))
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
    positions = np.unique(np.concatenate([x['positions'] for x in results])) if results else np.zeros(0)
    print('thread finished processing...:')
    print('%d observations' % sum(x['n_obs'] for x in results))
    print('%d positions' % len(positions))
    print('%d regions with multiple methylated bases' % sum(x['n_multi'] for x in results))
    print('%d observations with skips included' % sum(x['n_wskips'] for x in results))
    print('%d observations with too many skips' % sum(x['n_skipped'] for x in results))
    return True
