"""Worker for tests/test_shards.py::test_site_reduction_gloo_world2 (one process per rank, gloo)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from tests.test_shards import make_workload


def allreduce_site_counts(n_meth, n_total, first, dist=None):
    """Sum / min over ranks through torch.distributed (gloo, CPU tensors) -- the reduction the product does with ncclAllReduce
    through the C ABI (mc_site_allreduce); here only what it must equal.  Messages: 2 x 4 B + 8 B per site."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return n_meth, n_total, first
    import torch
    packed = torch.from_numpy(np.stack([n_meth, n_total]))
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    fmin = torch.from_numpy(first.copy())
    dist.all_reduce(fmin, op=dist.ReduceOp.MIN)
    packed = packed.numpy()
    return packed[0], packed[1], fmin.numpy()


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo', init_method='env://')
    from mcaller_amd import make_bed
    from tests import shard
    from mcaller_amd.extract_contexts import submodel_setup
    codes, ref, table, qual = make_workload(n_rows=120000, seed=33, genome_len=40000)
    _, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
    lo, hi = shard.shard_bounds(table, world)[rank]
    sub = table.slice_segments(lo, hi)
    rec = H.oracle_records(sub, ref.device_arrays(), qual, 6, 0, 0.0, tail_contig=shard.tail_contig(table, qual, 0.0, hi))
    H.oracle_score(rec, sub, qual, weights, soc, 6)
    index = make_bed.SiteIndex(ref.meth, 1)
    counts = make_bed.site_counts(rec, sub, index, row_offset=int(table.seg_row_begin[lo]))
    n_meth, n_total, first = allreduce_site_counts(*counts, dist=dist)
    if rank == 0:
        make_bed.write_bed_from_counts(out_path, n_meth, n_total, first, index, ref.names, ref.meth, 6, 1, 0.0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
