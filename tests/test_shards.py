"""Sharding by read and the per-site reduction, on CPU: the C oracle produces the records of every
shard; concatenated they must equal the records of the whole table.  The N>1 reduction runs under torch.distributed with
the gloo backend, world size 2."""
import os
import socket
import subprocess
import sys

import numpy as np

from tests import helpers as H


def make_workload(n_rows=150000, seed=21, motif='GATC', genome_len=300000):
    from mcaller_amd import synth
    codes = synth.genome(length=genome_len, seed=9)
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(n_rows, seed=seed, codes=codes, read_len=(800, 3000))
    return codes, ref, table, qual


def sharded_records(table, ref, qual, n_shards, k=6, skip=0, qthresh=0.0):
    from tests import shard
    bounds = shard.shard_bounds(table, n_shards)
    parts, ro, so, nr, tr = [], [], [], [], []
    for lo, hi in bounds:
        sub = table.slice_segments(lo, hi)
        trow, tail = shard.tail_close(table, qual, qthresh, hi)
        parts.append(H.oracle_records(sub, ref.device_arrays(), qual, k, skip, qthresh, tail_contig=tail))
        ro.append(int(table.seg_row_begin[lo]))
        so.append(lo)
        nr.append(sub.n_rows)
        tr.append(trow)
    return shard.concat_records(parts, k, ro, so, nr, tr), bounds


def test_shards_concatenate_to_the_whole():
    codes, ref, table, qual = make_workload()
    whole = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    for n in (2, 3, 8):
        rec, bounds = sharded_records(table, ref, qual, n)
        assert len(bounds) == n and bounds[0][0] == 0 and bounds[-1][1] == table.n_seg
        rows = [table.seg_row_begin[hi] - table.seg_row_begin[lo] for lo, hi in bounds]
        assert max(rows) < 2.5 * table.n_rows / n                 # balanced by rows
        H.assert_records_equal(rec, whole, 6)


def test_shards_with_quality_filter_and_skips():
    codes, ref, table, qual = make_workload(seed=22, motif='A')
    whole = H.oracle_records(table, ref.device_arrays(), qual, 6, 1, 9.0)
    rec, _ = sharded_records(table, ref, qual, 4, skip=1, qthresh=9.0)
    H.assert_records_equal(rec, whole, 6)


def test_repeated_read_names_are_not_cut():
    from mcaller_amd import _lib
    from tests import shard
    codes, ref, table, qual = make_workload(n_rows=30000)
    table.seg_read[-1] = table.seg_read[0]                        # the last read reuses the first read's name
    assert shard.has_repeated_names(table)
    bounds = shard.shard_bounds(table, 4)
    assert bounds[0] == (0, table.n_seg) and all(lo == hi for lo, hi in bounds[1:])


def test_site_reduction_gloo_world2(tmp_path):
    """Two ranks (gloo): each reduces its shard's records per site; all-reduce; rank 0 writes the BED.  Must equal the BED
    of the single-process run."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = [str(tmp_path / ('bed%d' % w)) for w in (1, 2)]
    for world, path in zip((1, 2), out):
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port + world),
                       PYTHONPATH=H.REPO)
            procs.append(subprocess.Popen([sys.executable, os.path.join(H.REPO, 'tests', '_reduce_worker.py'), path], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
    a, b = open(out[0]).read(), open(out[1]).read()
    assert a == b and a.count('\n') > 50


def test_bed_from_reduced_counts_equals_make_bed_on_the_diffs_text(tmp_path):
    """The per-site reduction (site_counts -> write_bed_from_counts, what the all-reduce feeds) writes the same BED as the
    reference-compatible text path (records -> .diffs rows -> make_bed), for depth and fraction thresholds."""
    import contextlib
    import io
    from mcaller_amd import make_bed
    from mcaller_amd import extract_contexts as ec
    codes, ref, table, qual = make_workload(n_rows=120000, seed=44, motif='GATC', genome_len=30000)
    modelset = H.load_modelset('r95')
    _, weights, _, soc = ec.submodel_setup(modelset, 'A')
    rec = H.oracle_records(table, ref.device_arrays(), qual, 6, 0, 0.0)
    H.oracle_score(rec, table, qual, weights, soc, 6)

    class P(object):
        pass
    P.table, P.ref, P.qual = table, ref, qual
    P.qual_obj = [np.float64(q) for q in qual]
    P.fatal = None
    fin = ec.Finisher(P, 6, 'A', False, modelset=modelset)
    with contextlib.redirect_stdout(io.StringIO()):
        assert fin.run(rec) is None
    diffs = tmp_path / 'syn.eventalign.diffs.6'
    diffs.write_bytes(fin.text())
    index = make_bed.SiteIndex(ref.meth, 1)
    counts = make_bed.site_counts(rec, table, index)
    for depth, thresh in ((1, 0.5), (5, 0.5), (3, 0.2), (8, 0.8)):
        with contextlib.redirect_stdout(io.StringIO()):
            make_bed.main(['-f', str(diffs), '-d', str(depth), '-t', str(thresh)])
        want = (tmp_path / 'syn.methylation.summary.bed').read_text()
        out = tmp_path / 'reduced.bed'
        make_bed.write_bed_from_counts(str(out), counts[0], counts[1], counts[2], index, ref.names, ref.meth, 6, depth, thresh)
        assert out.read_text() == want and want.count('\n') >= (3 if depth == 1 else 0)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE in the environment) starts two rank processes
    itself and rank 0 reports n_gpus = 2 (--dry-ranks: the rendezvous only, nothing touches a GPU); under a launcher's
    environment it does not spawn."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(H.REPO, 'bench.py'), '--gpus', '2', '--dry-ranks'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['launcher'] == 'bench.py itself'
    # both legs' schema: the weak leg is the contract's keys, the strong leg (BASELINE.json configs[3]: ONE file through
    # `mCaller --gpus N --bed`) a top-level object of its own -- summarised; its phases and workers are in the details file
    assert line['scaling'] == 'weak' and {'value', 'ms_per_step', 'ms_per_step_steady', 'ms_per_step_fp64_mlp', 'metric', 'unit', 'dtype',
                                          'roofline', 'cpu_baseline', 'vs_baseline', 'config'} <= set(line)
    assert 'f32' in line['dtype'] and 'f64' in line['dtype']          # (the classifier's hidden layer is fp32: the line says so)
    strong = line['strong_scaling']
    assert strong['scaling'] == 'strong' and strong['n_gpus'] == 2
    r = subprocess.run([sys.executable, os.path.join(H.REPO, 'bench.py'), '--gpus', '1', '--dry-ranks'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1])['n_gpus'] == 1


def test_bench_rank_failure_is_the_exit_code():
    """A rank that dies takes the others down and makes the launcher exit non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['MCALLER_BENCH_FAIL_RANK'] = '1'
    r = subprocess.run([sys.executable, os.path.join(H.REPO, 'bench.py'), '--gpus', '2', '--dry-ranks'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'rank 1 exited' in r.stderr


def test_site_counts_of_the_workers_are_combined_on_the_host_when_the_collective_did_not_finish():
    """multi_gpu.combine_site_counts: rank 0's all-reduced counts only if EVERY worker finished the collective; otherwise the
    workers' own counts are added up (min of the first-seen rows) -- never an error after the rows have been written."""
    from mcaller_amd.multi_gpu import combine_site_counts
    big = np.iinfo(np.int64).max
    own = [(np.array([1, 0, 2], np.int32), np.array([2, 0, 3], np.int32), np.array([7, big, 4], np.int64)),
           (np.array([0, 1, 1], np.int32), np.array([1, 1, 1], np.int32), np.array([big, 9, 2], np.int64))]
    summed = (np.array([1, 1, 3], np.int32), np.array([3, 1, 4], np.int32), np.array([7, 9, 2], np.int64))
    reduced = tuple(a + 0 for a in summed)
    ok = [dict(own=own[0], reduced=reduced, collective_done=True, err=None), dict(own=own[1], reduced=None, collective_done=True, err=None)]
    got = combine_site_counts(ok)
    assert all(np.array_equal(g, w) for g, w in zip(got[:3], reduced)) and 'ncclAllReduce' in got[3]
    bad = [dict(own=own[0], reduced=reduced, collective_done=True, err=None),
           dict(own=own[1], reduced=None, collective_done=False, err='ncclAllReduce failed: unhandled system error')]
    got = combine_site_counts(bad)
    assert all(np.array_equal(g, w) for g, w in zip(got[:3], summed)) and 'summed on the host' in got[3] and 'unhandled' in got[3]
    host = [dict(own=o, reduced=None, collective_done=False, err=None) for o in own]
    got = combine_site_counts(host)
    assert all(np.array_equal(g, w) for g, w in zip(got[:3], summed))


def test_bench_line_is_small_strict_json():
    """The driver keeps the tail of stdout: round 5's 20.8 KB line went unparsed.  The contract line is strict JSON (no NaN /
    Infinity), under 6000 characters, and carries the contract's keys + roofline + cpu_baseline -- checked on the dry-run line
    and on the compaction of the largest full results on record (round 5's lines, 20 KB each, and this round's details)."""
    import glob
    import json
    import bench

    def strict(text):
        return json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError('not strict JSON: ' + c)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(H.REPO, 'bench.py'), '--dry-ranks'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out_lines = r.stdout.splitlines()
    assert len(out_lines) == 1 and len(out_lines[0]) < bench.LINE_LIMIT
    strict(out_lines[0])
    fulls = sorted(glob.glob(os.path.join(H.REPO, 'profiles', 'r05_bench*.json')) + glob.glob(os.path.join(H.REPO, 'profiles', 'r06_bench*details*.json')))
    assert len(fulls) >= 4
    for path in fulls:
        full = json.load(open(path))
        full['poison'] = float('nan')
        full['roofline']['traffic_per_kernel'] = {'k%d' % i: 'x' * 100 for i in range(100)}          # (whatever a leg grows)
        text = json.dumps(bench.contract_line(full))
        line = strict(text)
        assert len(text) < bench.LINE_LIMIT, (path, len(text))
        assert {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'} <= set(line), path
        assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(line['roofline']), path
        assert 'workload' in line['config'] and isinstance(line['value'], float)
        if line['cpu_baseline']:
            assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline']), path
    # a leg that balloons is shed before the contract's keys are
    full = json.load(open(fulls[0]))
    full['config']['workload'] = 'w' * 3000
    full['cpu_baseline_reference_like']['cpu_model'] = 'c' * 3500
    text = json.dumps(bench.contract_line(full))
    assert len(text) < bench.LINE_LIMIT and 'roofline' in json.loads(text) and 'cpu_baseline' in json.loads(text)


def test_worker_parts_are_joined_behind_what_the_output_holds(tmp_path):
    """multi_gpu._join_parts: the parts in order behind whatever the output file holds (the reference appends), a fresh output is
    the first part renamed, the rest copied in the kernel or -- where copy_file_range is missing or refuses -- block by block."""
    import random
    from mcaller_amd.multi_gpu import _join_parts
    rng = random.Random(3)
    blobs = [bytes(rng.getrandbits(8) for _ in range(n)) for n in (1000, 0, 70000, 5)]

    def files(tag):
        paths = []
        for i, b in enumerate(blobs):
            p = str(tmp_path / ('%s%d' % (tag, i)))
            open(p, 'wb').write(b)
            paths.append(p)
        return paths

    out = str(tmp_path / 'fresh')
    parts = files('a')
    _join_parts(out, parts)
    assert open(out, 'rb').read() == b''.join(blobs) and not any(os.path.exists(p) for p in parts)
    out = str(tmp_path / 'appended')
    open(out, 'wb').write(b'rows of an earlier run\n')
    _join_parts(out, files('b'))
    assert open(out, 'rb').read() == b'rows of an earlier run\n' + b''.join(blobs)
    out = str(tmp_path / 'empty')
    open(out, 'wb').close()
    _join_parts(out, files('c')[2:3])
    assert open(out, 'rb').read() == blobs[2]
    saved = getattr(os, 'copy_file_range', None)
    if saved is not None:
        del os.copy_file_range
    try:
        out = str(tmp_path / 'plain')
        open(out, 'wb').write(b'X')
        _join_parts(out, files('d'))
        assert open(out, 'rb').read() == b'X' + b''.join(blobs)
    finally:
        if saved is not None:
            os.copy_file_range = saved
