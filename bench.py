#!/usr/bin/env python3
"""bench.py -- m6A calls/sec of the hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (strand resolve + window scan + record ordering + MLP classifier, then the
D2H copy of the flush records) over one batch of synthetic eventalign rows that is already resident in HBM.
Workload at N=1: BASELINE.json configs[2] -- synthetic 10^8 events, -m GATC, NN classifier (r95 two-base MLP),
skip_thresh 0.  N>1: every rank scans its own 10^8-row shard of reads (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the window-scan kernel
(algorithmic bytes = 17 B/event row + 64 B/emitted call) and `cpu_baseline` (the C oracle,
single host core, same workload sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def dist_setup(n_gpus):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod      # plumbing only: rendezvous, barrier, max-over-ranks
        dist_mod.init_process_group(backend='gloo', init_method='env://')
        dist = dist_mod
    return rank, world, local, dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)     # 60 ms of timed work: the pipeline's fill and drain (one pass) amortise
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--events', type=float, default=1e8, help='event rows per GPU')
    ap.add_argument('--motif', default='GATC')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='one pass at a time (mc_extract_features) instead of the pipelined passes')
    ap.add_argument('--time-every', type=int, default=8,
                    help='pipelined passes: hipEvents that time a pass go with every n-th pass (one costs the queue ~9 us)')
    ap.add_argument('--cpu-events', type=float, default=1e8, help='rows of the same workload timed on the CPU oracle')
    args = ap.parse_args()

    rank, world, local, dist = dist_setup(args.gpus)
    from mcaller_amd import synth, _lib
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from tests import helpers as H

    n_rows = int(args.events)
    t_gen = time.time()
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif=args.motif)
    table, qual = synth.make_table(n_rows, seed=1000 + rank, codes=codes)
    t_gen = time.time() - t_gen
    modelset = H.load_modelset('r95')
    _, weights, _, soc = submodel_setup(modelset, 'A')

    dev_index = 0 if os.environ.get('MCALLER_BENCH_ONE_DEVICE') else local      # (one-GPU boxes: test the N>1 plumbing)
    numa_node = Device.bind_host_to_numa_node(dev_index) if (world > 1 or os.environ.get('MCALLER_BENCH_BIND')) else None  # pinned buffers next to the rank's GPU
    dev = Device(dev_index)
    dev.set_reference(ref.device_arrays())
    t_up = time.time()
    dev.upload_table(table)
    t_up = time.time() - t_up
    dev.set_read_quality(qual)
    dev.set_mlp(weights, soc)

    # A step = one pass of the hot path over the resident table, records (slot means, sites, probabilities) landing in
    # pinned host memory.  Passes are pipelined (the library's streaming interface, mc_extract_features_async /
    # mc_wait_records): K0 + K1 of consecutive passes back to back on one stream, K2 + packing on a side stream, copy-outs
    # back to back on a third, four passes in flight at most; every pass's records are complete in host memory before the
    # timed region ends.
    # --no-pipeline times mc_extract_features instead (one pass at a time, host sync inside).
    def step_sync():
        dev.run(6, 0, 0.0, tail_contig=-1, score=True)
        return dev.fetch(copy=False)

    def run_steps(n_steps, on_done):
        if args.no_pipeline:
            for _ in range(n_steps):
                on_done(step_sync())
            return
        depth = min(3, n_steps)                                  # passes in flight (the library allows four)
        for _ in range(depth):
            dev.run_async(6, 0, 0.0, tail_contig=-1, score=True)
        dev.wait_begin()                                         # copy-out of the oldest pass started
        for _ in range(n_steps - depth):
            dev.wait_begin()                                     # ... and of the one behind it, as soon as it is computed:
            dev.run_async(6, 0, 0.0, tail_contig=-1, score=True) # the transfers run back to back; the next pass enqueued;
            on_done(dev.wait())                                  # the oldest pass's records are in host memory
        for _ in range(depth):
            on_done(dev.wait())

    def barrier():
        if dist is not None:
            dist.barrier()

    k1_ms, tot_ms, last = [], [], [None]

    # The kernel times come from hipEvents on the ctx stream.  An event between two kernels costs that queue ~9 us (6 % of a
    # pass), so in the pipelined loop only every n-th pass carries the two events that do nothing but time it; kernel_ms and
    # the roofline are averages over those passes of the timed region (one pass at a time: every pass).
    time_every = 1 if args.no_pipeline else max(1, min(args.time_every, max(1, args.steps // 4)))
    dev.set_pass_timing(time_every)

    def on_done(rec):
        last[0] = rec
        if args.no_pipeline or dev.last_pass_timed():
            tm = dev.times_ms()
            k1_ms.append(tm['window_scan'] + tm['emit'])
            tot_ms.append(tm)

    if args.warmup:
        run_steps(args.warmup, on_done)
    del k1_ms[:], tot_ms[:]
    barrier()
    dev.sync()                              # hipDeviceSynchronize: nothing of the warm-up is left on any stream
    t0 = time.perf_counter()
    run_steps(args.steps, on_done)          # the last wait() returns when the last pass's records are in host memory
    dev.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    rec = last[0]
    info = rec.info[:rec.n]
    n_calls = int(((info & _lib.I_TOO_MANY) == 0).sum())

    calls_total, elapsed_max = n_calls, elapsed
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([n_calls], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        elapsed_max, calls_total = float(t[0]), int(c[0])

    # the one exchange step of the multi-GPU job: per-site counts summed over ranks (feeds make_bed).  Outside the timed
    # steps.  Counted on the device from the records of the last step and all-reduced with RCCL through the C ABI
    # (mc_site_counts / mc_site_allreduce); if that fails (e.g. a plumbing test with two ranks on one GPU) the same
    # reduction goes through torch.distributed (nccl, then gloo) from the host copy of the records.
    reduction = None
    if dist is not None:
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
        dist.all_gather_object(flags, e_native is None)        # every rank takes the same branch
        if all(flags):
            reduction = {'backend': 'rccl (ncclAllReduce through the C ABI)', 'ms': ms, 'observations': int(n_total.sum()),
                         'observations_expected': calls_total, 'bytes': int(index.n * 16), 'sites': int(index.n)}
        else:
            try:
                import torch
                counts = make_bed.site_counts(rec, table, index, row_offset=rank * n_rows)

                def reduce_with(backend):
                    group = dist.new_group(backend=backend) if backend == 'nccl' else None
                    packed = torch.from_numpy(np.stack([counts[0], counts[1]]))
                    fmin = torch.from_numpy(counts[2].copy())
                    if backend == 'nccl':
                        torch.cuda.set_device(0 if os.environ.get('MCALLER_BENCH_ONE_DEVICE') else local)
                        packed, fmin = packed.cuda(), fmin.cuda()
                    t_r = time.perf_counter()
                    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
                    dist.all_reduce(fmin, op=dist.ReduceOp.MIN, group=group)
                    if backend == 'nccl':
                        torch.cuda.synchronize()
                    return packed.cpu(), fmin.cpu(), (time.perf_counter() - t_r) * 1e3

                backend = 'torch nccl (native RCCL path failed: %s)' % e_native
                try:
                    packed, fmin, ms = reduce_with('nccl')
                except Exception as e_nccl:                    # noqa
                    backend = 'gloo (native: %s; torch nccl: %s)' % (e_native, type(e_nccl).__name__)
                    packed, fmin, ms = reduce_with('gloo')
                reduction = {'backend': backend, 'ms': ms, 'observations': int(packed[1].sum().item()),
                             'bytes': int(packed.numel() * packed.element_size() + fmin.numel() * 8)}
            except Exception as e:                             # noqa
                reduction = {'error': '%s: %s' % (type(e).__name__, e)}

    if rank == 0:
        k1 = float(np.mean(k1_ms))
        kernel_ms = {k: float(np.mean([t[k] for t in tot_ms])) for k in tot_ms[0]}
        if kernel_ms.get('emit') == 0.0:      # pipelined passes time scan + ordering + emit as one span (no event in between)
            kernel_ms['window_scan_and_emit'] = kernel_ms.pop('window_scan')
            del kernel_ms['emit']
        alg_bytes = 17.0 * n_rows + 64.0 * n_calls
        traffic = None      # HBM bytes per step of the same kernels from the committed rocprofv3 PMC passes (same workload)
        pmc = os.path.join(REPO, 'profiles', 'r01_pmc.json')
        if os.path.exists(pmc) and n_rows == 100000000 and args.motif == 'GATC':
            traffic = json.load(open(pmc)).get('feature_extraction_hbm_bytes_per_step')
        achieved = alg_bytes / (k1 * 1e-3) / 1e9
        out = {
            'metric': 'm6A calls/sec (GATC motif, E. coli-like synthetic eventalign)',
            'value': calls_total * args.steps / elapsed_max,
            'unit': 'calls/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed_max / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'synthetic %.0e eventalign rows per GPU, -m %s, NN classifier (r95 two-base MLP), '
                                   'skip_thresh 0, table resident in HBM' % (n_rows, args.motif),
                       'passes_in_flight': 1 if args.no_pipeline else min(3, args.steps),
                       'events_per_gpu': n_rows, 'calls_per_gpu': n_calls, 'flush_records_per_gpu': int(rec.n),
                       'events_per_s': n_rows * world * args.steps / elapsed_max,
                       'kernel_ms': kernel_ms, 'kernel_ms_from_passes': len(tot_ms), 'timing_events_every_n_passes': time_every,
                       'h2d_table_s': t_up, 'generate_s': t_gen, 'site_reduction': reduction, 'numa_node_rank0': numa_node,
                       # SURVEY.md §8(d)'s three timings, calls/s on one GPU: kernels only; H2D of the table + one pass +
                       # D2H of the records; file to file is measured by tools/file_to_file.py (profiles/r01_file_to_file.log)
                       'calls_per_s_kernels_only': n_calls / (float(np.mean([t['total'] for t in tot_ms])) * 1e-3),
                       'calls_per_s_with_h2d': n_calls / (t_up + elapsed / args.steps)},
            'roofline': {'bound': 'hbm', 'kernel': 'k1_scan + k1_group_scan + k1_list + k1_emit (feature extraction)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_gbs': (traffic / (k1 * 1e-3) / 1e9) if traffic else None,
                         'algorithmic_bytes': alg_bytes, 'kernel_ms': k1},
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU leg: rank 0 at N=1 only
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
                from mcaller_amd import shard
                cores = min(os.cpu_count() or 1, 64)
                bounds = [b for b in shard.shard_bounds(sub, cores) if b[1] > b[0]]
                subs = [(sub.slice_segments(lo, hi), shard.tail_contig(sub, qual, 0.0, hi)) for lo, hi in bounds]

                def one(job):
                    st, tail = job
                    o = H.oracle_records(st, arrays, qual, 6, 0, 0.0, tail_contig=tail)
                    H.oracle_score(o, st, qual, weights, soc, 6)
                    return int(((o.info[:o.n] & _lib.I_TOO_MANY) == 0).sum())
                t2 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=len(subs)) as ex:
                    mt_calls = sum(ex.map(one, subs))
                dt2 = time.perf_counter() - t2
                out['cpu_baseline_all_cores'] = {'value': mt_calls / dt2, 'unit': 'calls/s', 'cores': len(subs), 'kind': 'port',
                                                 'sample': 'same rows, sharded by read over %d threads, %.2f s' % (len(subs), dt2),
                                                 'events_per_s': sub.n_rows / dt2}
            except Exception as e:                              # noqa
                out['cpu_baseline_all_cores'] = {'error': str(e)}
            out['cpu_baseline'] = {'value': oc / dt, 'unit': 'calls/s', 'cores': 1, 'kind': 'port',
                                   'sample': '%d event rows of the same workload (C oracle: literal window machine + '
                                             'MLP, one host core, %.2f s)' % (sub.n_rows, dt),
                                   'events_per_s': sub.n_rows / dt}
        print(json.dumps(out))
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
