"""The command line end to end on the GPU: same files as the reference's README commands (README.md:126-148)."""
import contextlib
import io
import os
import shutil
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def td(tmp_path_factory):
    return H.testdata_paths(str(tmp_path_factory.mktemp('testdata')))


def run_cli(td, tmp_path, extra):
    from mcaller_amd import mCaller
    tsv = str(tmp_path / 'masonread1.eventalign.tsv')
    shutil.copy(td['tsv'], tsv)
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        mCaller.main(extra + ['-r', td['fasta'], '-e', tsv, '-f', td['fastq'], '-d', model])
    return str(tmp_path / 'masonread1.eventalign.diffs.6'), buf.getvalue()


def test_readme_command_positions(td, tmp_path):
    out, stdout = run_cli(td, tmp_path, ['-p', td['test_positions_m6A.txt']])
    assert open(out).read() == open(os.path.join(H.GOLDEN, 'ref_outputs', 'config1_positions_m6A.diffs.6')).read()
    want = open(os.path.join(H.GOLDEN, 'ref_outputs', 'config1_positions_m6A.stdout')).read().replace('<DIR>/', str(tmp_path) + '/')
    assert [l for l in stdout.split('\n') if l.strip()] == [l for l in want.split('\n') if l.strip()]


def test_readme_command_motif_then_make_bed(td, tmp_path):
    from mcaller_amd import make_bed
    out, _ = run_cli(td, tmp_path, ['-m', 'GATC'])
    assert open(out).read() == open(os.path.join(H.GOLDEN, 'ref_outputs', 'motif_GATC.diffs.6')).read()
    with contextlib.redirect_stdout(io.StringIO()):
        make_bed.main(['-f', out, '-d', '1', '-t', '0.5'])
    bed = str(tmp_path / 'masonread1.methylation.summary.bed')
    assert open(bed).read() == open(os.path.join(H.GOLDEN, 'ref_outputs', 'motif_GATC.bed')).read()


def test_threads_flag_sorts_like_the_reference(td, tmp_path):
    out, _ = run_cli(td, tmp_path, ['-m', 'GATC', '-t', '4'])
    want = sorted(set(open(os.path.join(H.GOLDEN, 'ref_outputs', 'motif_GATC.diffs.6'), 'rb').read().splitlines(True)))
    assert open(out, 'rb').read() == b''.join(want)


def test_train_mode_feature_matrix(td, tmp_path):
    """--train: the labelled rows and the returned dicts (the fit itself is scikit-learn's and stochastic)."""
    import json
    from mcaller_amd.extract_contexts import extract_features
    from mcaller_amd.read_qual import extract_read_quality
    tsv = str(tmp_path / 'masonread1.eventalign.tsv')
    shutil.copy(td['tsv'], tsv)
    posf = td['test_positions.txt']
    with contextlib.redirect_stdout(io.StringIO()):
        sig, ctx = extract_features(tsv, td['fasta'], extract_read_quality(td['fastq']), 6, 0, 0, None, 'NN', 0,
                                    endline=os.path.getsize(tsv), train=True, pos_label=H.pos2label(posf), base='A',
                                    motif=None, positions_list=posf)
    out = open(str(tmp_path / 'masonread1.eventalign.diffs.6.train.tmp0')).read()
    assert out == open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.diffs.6.train')).read()
    gold = json.load(open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.dicts.json')))
    assert H.plain_signals(sig) == gold['signals'] and ctx == gold['contexts']


def test_sharded_run_on_one_gpu_equals_the_whole():
    """Shards scanned one after the other on the same GPU (tail_contig path of the kernels) == the whole table."""
    from mcaller_amd import synth
    from tests import shard
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    codes = synth.genome(length=500000, seed=9)
    ref = synth.SynthRef(codes)
    table, qual = synth.make_table(600000, seed=77, codes=codes, read_len=(2000, 9000))
    _, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
    dev = Device(0)
    dev.set_reference(ref.device_arrays())
    dev.set_read_quality(qual)
    dev.set_mlp(weights, soc)
    dev.upload_table(table)
    whole = dev.extract(6, 0, 0.0)
    parts, ro, so, nr, tr = [], [], [], [], []
    for lo, hi in shard.shard_bounds(table, 4):
        sub = table.slice_segments(lo, hi)
        trow, tail = shard.tail_close(table, qual, 0.0, hi)
        dev.upload_table(sub)
        parts.append(dev.extract(6, 0, 0.0, tail_contig=tail))
        ro.append(int(table.seg_row_begin[lo])); so.append(lo); nr.append(sub.n_rows); tr.append(trow)
    rec = shard.concat_records(parts, 6, ro, so, nr, tr)
    H.assert_records_equal(rec, whole, 6, prob_tol=0.0)
    dev.close()


def test_site_reduction_on_device_equals_host(tmp_path):
    """mc_site_counts + mc_site_allreduce (RCCL communicator of one rank: loads librccl, ncclCommInitRank, no exchange)
    == the numpy reduction of the same records; and the BED written from it == make_bed on the .diffs text."""
    from mcaller_amd import synth, make_bed
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    codes = synth.genome(length=60000, seed=4)
    ref = synth.SynthRef(codes, motif='A')                 # dense: every site is hit by several reads
    table, qual = synth.make_table(400000, seed=12, codes=codes, read_len=(1500, 6000))
    _, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
    dev = Device(0)
    dev.set_reference(ref.device_arrays())
    dev.set_read_quality(qual)
    dev.set_mlp(weights, soc)
    dev.upload_table(table)
    rec = dev.extract(6, 0, 0.0)
    index = make_bed.SiteIndex(ref.meth, 1)
    want = make_bed.site_counts(rec, table, index, row_offset=1000)
    dev.comm_init(1, 0, Device.comm_unique_id())
    pending = dev.site_counts(row_offset=1000)
    assert pending == int(np.isnan(rec.prob[:rec.n][(rec.info[:rec.n] & 0x200) == 0]).sum())
    make_bed.add_pending_site_counts(dev, rec, table, index, row_offset=1000)
    n_meth, n_total, first, ms = dev.site_allreduce()
    assert index.n == len(n_meth) > 1000 and n_total.sum() > 10000
    assert (n_meth == want[0]).all() and (n_total == want[1]).all() and (first == want[2]).all()
    dev.comm_destroy()
    dev.close()


def test_train_cli_fits_on_the_gpu(td, tmp_path):
    """`mCaller.py --train`: feature matrix from the HIP path, six fits in one mc_mlp_fit launch, model file written in
    the reference's format (or the neutral .npz without scikit-learn) and readable by model_io."""
    from mcaller_amd import mCaller, model_io
    tsv = str(tmp_path / 'masonread1.eventalign.tsv')
    shutil.copy(td['tsv'], tsv)
    model = str(tmp_path / 'trained.pkl')
    os.environ['MCALLER_SEED'] = '4'
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            mCaller.main(['-p', td['test_positions.txt'], '-r', td['fasta'], '-e', tsv, '-f', td['fastq'], '--train',
                          '-d', model])
    finally:
        del os.environ['MCALLER_SEED']
    out = buf.getvalue()
    assert 'NN general model scores: ' in out and 'Cross validation accuracy: ' in out and 'Finished training' in out
    scores = [float(x) for x in out.split('NN general model scores: ')[1].split('\n')[0].split(',')]
    assert len(scores) == 5 and all(0.0 <= s <= 1.0 for s in scores)
    assert open(str(tmp_path / 'masonread1.eventalign.diffs.6.train')).read() == \
        open(os.path.join(H.GOLDEN, 'ref_outputs', 'train_positions_all.diffs.6.train')).read()
    ms = model_io.load_model_file(model)
    w = ms.models['general']
    assert ms.twobase and w.W1.shape == (7, 100) and np.isfinite(w.W1).all() and np.abs(w.W1).max() > 0
    # the same seed gives the same model (deterministic kernel)
    model2 = str(tmp_path / 'trained2.pkl')
    os.environ['MCALLER_SEED'] = '4'
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mCaller.main(['-p', td['test_positions.txt'], '-r', td['fasta'], '-e', tsv, '-f', td['fastq'], '--train',
                          '-d', model2])
    finally:
        del os.environ['MCALLER_SEED']
    assert (model_io.load_model_file(model2).models['general'].W1 == w.W1).all()


def test_cli_sharded_over_two_workers_equals_one_gpu(tmp_path):
    """`--gpus 2`: the file is cut at read starts, one worker process per piece (both on GPU 0 here), the pieces' rows
    concatenated; output and counter lines equal the one-GPU run."""
    from mcaller_amd import synth, mCaller
    codes = synth.genome(length=300000, seed=21)
    table, qual = synth.make_table(400000, seed=5, codes=codes, read_len=(1500, 6000))
    d = str(tmp_path)
    tsv = os.path.join(d, 'syn.eventalign.tsv')
    synth.write_tsv(table, codes, tsv)
    seq = synth.codes_to_str(codes)
    with open(os.path.join(d, 'ref.fasta'), 'w') as fa:
        fa.write('>ecoli_syn\n' + '\n'.join(seq[i:i + 60] for i in range(0, len(seq), 60)) + '\n')
    with open(os.path.join(d, 'reads.fastq'), 'w') as fq:
        for i, name in enumerate(table.read_names):
            fq.write('@%s\nACGTACGTAC\n+\n%s\n' % (name, chr(33 + int(round(qual[i]))) * 10))
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    common = ['-m', 'GATC', '-r', os.path.join(d, 'ref.fasta'), '-e', tsv, '-f', os.path.join(d, 'reads.fastq'), '-d', model,
              '-q', '8']
    outs = []
    from mcaller_amd import multi_gpu
    real, took = multi_gpu.extract_features_sharded, []

    def spy(*a, **kw):
        took.append(real(*a, **kw))
        return took[-1]
    multi_gpu.extract_features_sharded = spy
    # (the last two: every worker streams its piece in four shards -- text parsed on the device, two passes in flight, the
    # tail of a worker's last shard from the next worker's first)
    for extra, shards in (([], None), (['--gpus', '2'], None), (['--gpus', '3'], None), (['--gpus', '2'], '4'), (['--gpus', '3'], '4')):
        out_path = tsv[:-4] + '.diffs.6'
        if os.path.exists(out_path):
            os.remove(out_path)
        os.environ['MCALLER_SHARD_DEVICES'] = ','.join(['0'] * (int(extra[1]) if extra else 1))
        if shards:
            os.environ['MCALLER_STREAM_SHARDS'] = shards
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                mCaller.main(common + extra)
        finally:
            del os.environ['MCALLER_SHARD_DEVICES']
            os.environ.pop('MCALLER_STREAM_SHARDS', None)
        lines = [l for l in buf.getvalue().split('\n') if 'observations' in l or 'positions' in l or 'regions' in l]
        outs.append((open(out_path, 'rb').read(), lines))
    multi_gpu.extract_features_sharded = real
    assert took == [True, True, True, True]                   # the sharded path ran (no fall-back to one GPU)
    assert outs[0][0].count(b'\n') > 200
    assert all(o == outs[0] for o in outs[1:])
    assert not [f for f in os.listdir(d) if '.part' in f or '.tmp' in f]


def test_cli_bed_from_the_per_site_reduction(tmp_path):
    """`--bed`: the BED written from the workers' per-site reductions (one worker: device-side counts; two workers on one
    GPU: RCCL refuses the duplicate device, the parent adds the workers' counts) equals make_bed on the `.diffs` file."""
    from mcaller_amd import synth, mCaller, make_bed
    codes = synth.genome(length=40000, seed=31)
    table, qual = synth.make_table(250000, seed=6, codes=codes, read_len=(1500, 6000))
    d = str(tmp_path)
    tsv = os.path.join(d, 'syn.eventalign.tsv')
    synth.write_tsv(table, codes, tsv)
    seq = synth.codes_to_str(codes)
    with open(os.path.join(d, 'ref.fasta'), 'w') as fa:
        fa.write('>ecoli_syn\n' + '\n'.join(seq[i:i + 60] for i in range(0, len(seq), 60)) + '\n')
    with open(os.path.join(d, 'reads.fastq'), 'w') as fq:
        for i, name in enumerate(table.read_names):
            fq.write('@%s\nACGTACGTAC\n+\n%s\n' % (name, chr(33 + int(round(qual[i]))) * 10))
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    common = ['-m', 'GATC', '-r', os.path.join(d, 'ref.fasta'), '-e', tsv, '-f', os.path.join(d, 'reads.fastq'), '-d', model,
              '--bed', '--bed_min_depth', '3', '--bed_mod_threshold', '0.3']
    bed_path = os.path.join(d, 'syn.methylation.summary.bed')
    diffs = tsv[:-4] + '.diffs.6'
    results = []
    # (one worker; two; one and three whose pieces are streamed in three shards each: the counts accumulate on the device
    # shard after shard, mc_site_counts_accumulate)
    for n, shards in ((1, None), (2, None), (1, '3'), (3, '3')):
        for f in (bed_path, diffs):
            if os.path.exists(f):
                os.remove(f)
        os.environ['MCALLER_SHARD_DEVICES'] = ','.join(['0'] * n)
        if shards:
            os.environ['MCALLER_STREAM_SHARDS'] = shards
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                mCaller.main(common + ['--gpus', str(n)])
        finally:
            del os.environ['MCALLER_SHARD_DEVICES']
            os.environ.pop('MCALLER_STREAM_SHARDS', None)
        results.append((open(bed_path).read(), buf.getvalue()))
    assert all('per-site reduction' in r[1] for r in results)
    assert 'summed on the host' in results[1][1]
    os.rename(bed_path, bed_path + '.reduced')
    with contextlib.redirect_stdout(io.StringIO()):
        make_bed.main(['-f', diffs, '-d', '3', '-t', '0.3'])
    want = open(bed_path).read()
    assert all(r[0] == want for r in results) and want.count('\n') > 5
    # --bed_vo: the same file with make_bed.py --vo's probability lists (two workers)
    os.remove(diffs)
    os.environ['MCALLER_SHARD_DEVICES'] = '0,0'
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mCaller.main(common + ['--gpus', '2', '--bed_vo'])
    finally:
        del os.environ['MCALLER_SHARD_DEVICES']
    got_vo = open(bed_path).read()
    with contextlib.redirect_stdout(io.StringIO()):
        make_bed.main(['-f', diffs, '-d', '3', '-t', '0.3', '--vo'])
    assert got_vo == open(bed_path).read() and got_vo != want and got_vo.count('\n') == want.count('\n')


def test_cli_bed_over_several_contigs_equals_make_bed(tmp_path):
    """`--bed` on files whose reads lie on several contigs: a window closed by a row of the NEXT contig is written with that
    contig in its chrom column (R8), so make_bed files it under a (contig, position) that is no marked site -- the
    reduction leaves such records to the host (mc_site_counts: n_cross_contig) and the BED equals make_bed's on the rows."""
    from mcaller_amd import mCaller, make_bed
    n_done = 0
    cases = [c for c in H.micro_cases() if c['flavour'] in ('multi_contig', 'plain', 'dense', 'quirk_names')]
    for i, case in enumerate(cases):
        a = case['args']
        if a['train'] or case['expected']['outcome'] != 'ok' or not case['expected']['text'] or a['k'] != 6:
            continue
        model = os.path.join(H.MODELS, H.MODEL_STEMS[a['model']] + '.npz')
        d = tmp_path / ('m%d' % i)
        d.mkdir()
        paths = H.materialise(case, str(d))
        argv = ['-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model, '-b', a['base'],
                '-s', str(a['skip_thresh']), '-q', str(a['qual_thresh']), '--bed', '--bed_min_depth', '1',
                '--bed_mod_threshold', '0.0']
        argv += ['-p', paths['positions']] if paths['positions'] else ['-m', a['motif']]
        os.environ['MCALLER_SHARD_DEVICES'] = '0'
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                mCaller.main(argv)
        finally:
            del os.environ['MCALLER_SHARD_DEVICES']
        diffs = paths['tsv'][:-4] + '.diffs.6'
        assert open(diffs).read() == case['expected']['text']
        bed = os.path.join(str(d), 'case.methylation.summary.bed')
        got = open(bed).read()
        assert got.count('\n') > 0
        with contextlib.redirect_stdout(io.StringIO()):
            make_bed.main(['-f', diffs, '-d', '1', '-t', '0.0'])
        assert got == open(bed).read(), case['seed']
        n_done += 1
    assert n_done >= 10


def test_strong_scaling_leg_of_the_bench_equals_the_one_gpu_run(tmp_path):
    """bench.py's strong-scaling leg (BASELINE.json configs[3]): one file -> `mCaller --gpus N --bed` in a process of its own,
    the CLI several times, the workers of the first run kept for the later ones (MCALLER_KEEP_WORKERS).  With 1, 2 and 3 workers
    sharing GPU 0 (RCCL refuses a communicator of one device twice: the host-sum branch) the `.diffs.6` bytes equal the plain
    one-GPU run's, every run; the leg reports per-worker seconds, the reduction's backend and bytes, and the BED's rows."""
    import bench
    from mcaller_amd import synth
    codes = synth.genome(length=300000, seed=23)
    table, qual = synth.make_table(600000, seed=9, codes=codes, read_len=(1500, 6000))
    d = str(tmp_path)
    synth.write_inputs(table, qual, codes, d)
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(H.REPO, 'tools', 'file_to_file.py'), '--inputs', d, '--runs', '2', '--json'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads(r.stdout.strip().splitlines()[-1])
    assert one['calls'] > 100
    beds = []
    for n in (1, 2, 3):
        leg = bench.strong_scaling_leg(d, n, rows=table.n_rows, one_gpu_sha=one['diffs_sha256'], one_device=True, runs=3)
        assert leg['scaling'] == 'strong' and leg['n_gpus'] == n and leg['diffs_equal_the_one_gpu_run'] is True
        assert leg['calls'] == one['calls'] and leg['rows'] == table.n_rows and len(leg['seconds_all']) == 3
        assert len(leg['workers']) == n and sum(w['rows'] for w in leg['workers']) == table.n_rows
        assert all(w['seconds'] > 0 for w in leg['workers'])
        red = leg['site_reduction']
        assert red['bytes'] == red['sites'] * 16 and red['sites'] > 100 and red['observations'] == one['calls']
        assert ('summed on the host' in red['backend']) and leg['bed_rows'] > 50
        assert leg['calls_per_s'] > 0 and leg['events_per_s'] > 0 and set(bench.STRONG_KEYS) <= set(leg)
        beds.append(leg['bed_rows'])
    assert len(set(beds)) == 1


def test_workers_are_kept_between_files_and_replaced_when_the_devices_change(tmp_path):
    """MCALLER_KEEP_WORKERS: the workers of one sharded run take the next file (same processes); another device list, or a
    worker that died, starts new ones; a piece without a read (more GPUs than reads) is a finished piece, not a fall-back."""
    from mcaller_amd import synth, mCaller, multi_gpu
    codes = synth.genome(length=200000, seed=25)
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    outs, pids, reused = {}, [], []
    os.environ['MCALLER_KEEP_WORKERS'] = '1'
    try:
        for tag, rows, seed, devs, read_len in (('a', 200000, 3, '0,0', (1500, 6000)), ('b', 150000, 4, '0,0', (1500, 6000)),
                                                ('a', 200000, 3, '0,0,0', (1500, 6000)), ('c', 30000, 5, '0,0,0', (14000, 15000))):
            d = str(tmp_path / (tag + devs.replace(',', '')))
            os.makedirs(d)
            table, qual = synth.make_table(rows, seed=seed, codes=codes, read_len=read_len)
            paths = synth.write_inputs(table, qual, codes, d)
            os.environ['MCALLER_SHARD_DEVICES'] = devs
            with contextlib.redirect_stdout(io.StringIO()):
                mCaller.main(['-m', 'GATC', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model,
                              '--gpus', str(devs.count(',') + 1), '--bed', '--bed_min_depth', '1'])
            assert multi_gpu.last_run is not None, 'the sharded path declined'
            assert multi_gpu.last_run['rows'] == table.n_rows
            pids.append([p.pid for p in multi_gpu._kept.procs])
            reused.append(multi_gpu.last_run['workers_reused'])
            outs.setdefault(tag, []).append(open(paths['tsv'][:-4] + '.diffs.6', 'rb').read())
            if tag == 'c':          # two reads, three workers: one range is empty
                assert table.n_reads < 3 and min(w['rows'] for w in multi_gpu.last_run['workers']) == 0
        assert pids[0] == pids[1] and len(pids[2]) == 3 and pids[2] == pids[3] and reused == [False, True, False, True]
        assert outs['a'][0] == outs['a'][1] and outs['a'][0].count(b'\n') > 50
    finally:
        del os.environ['MCALLER_KEEP_WORKERS']
        os.environ.pop('MCALLER_SHARD_DEVICES', None)
        multi_gpu._stop_kept()


def test_cli_bed_when_the_reduction_does_not_finish(tmp_path):
    """A step of the per-site reduction that does not finish in time (here: a deadline nobody can meet) must not cost the
    sharded run: the workers are stopped, their rows are kept, the BED is made from the rows -- `.diffs` and BED as ever."""
    from mcaller_amd import synth, mCaller, make_bed, multi_gpu
    codes = synth.genome(length=40000, seed=33)
    table, qual = synth.make_table(200000, seed=7, codes=codes, read_len=(1500, 6000))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    argv = ['-m', 'GATC', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model, '--bed', '--bed_min_depth', '2']
    bed_path = os.path.join(str(tmp_path), 'syn.methylation.summary.bed')
    diffs = paths['tsv'][:-4] + '.diffs.6'
    outs = []
    for timeout in (None, '0.000001'):
        for f in (bed_path, diffs):
            if os.path.exists(f):
                os.remove(f)
        os.environ['MCALLER_SHARD_DEVICES'] = '0,0'
        if timeout:
            os.environ['MCALLER_COMM_TIMEOUT'] = timeout
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                mCaller.main(argv + ['--gpus', '2'])
        finally:
            del os.environ['MCALLER_SHARD_DEVICES']
            os.environ.pop('MCALLER_COMM_TIMEOUT', None)
        assert multi_gpu.last_run is not None                       # the sharded path ran both times
        outs.append((open(diffs, 'rb').read(), open(bed_path).read(), multi_gpu.bed_written, multi_gpu.last_run['site_reduction']['backend']))
    assert outs[0][2] is True and outs[1][2] is False and outs[1][3].startswith('rows (')
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and outs[0][1].count('\n') > 5
    assert not [f for f in os.listdir(str(tmp_path)) if '.part' in f or '.tmp' in f]


def test_train_mode_over_two_and_three_workers_equals_one_gpu(tmp_path):
    """The reference fans train-mode extraction out over its `-t` processes too (mCaller.py:72-87): `extract_features_sharded(...,
    train=True)` -- every worker streams its byte range with features only, the parent merges the pieces' dicts in file order --
    returns the dicts and writes the `.train` rows the one-GPU run does."""
    from mcaller_amd import synth, multi_gpu
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    codes = synth.genome(length=150000, seed=35)
    table, qual = synth.make_table(500000, seed=12, codes=codes, read_len=(1500, 6000))
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    seq = np.frombuffer(synth.codes_to_str(codes).encode('ascii'), dtype=np.uint8)
    hit = np.flatnonzero((seq[:-3] == ord('G')) & (seq[1:-2] == ord('A')) & (seq[2:-1] == ord('T')) & (seq[3:] == ord('C')))
    posfile = str(tmp_path / 'positions.txt')
    with open(posfile, 'w') as fh:
        for p in hit:
            fh.write('ecoli_syn\t%d\t+\t%s\n' % (p + 1, 'm6A' if (p + 1) % 3 == 0 else 'A'))
            fh.write('ecoli_syn\t%d\t-\t%s\n' % (p + 2, 'm6A' if (p + 2) % 3 == 0 else 'A'))
    pos_label = H.pos2label(posfile)
    r2q = extract_read_quality(paths['fastq'])
    size = os.path.getsize(paths['tsv'])
    tmp = paths['tsv'][:-4] + '.diffs.6.train.tmp0'
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sig, ctx = ec.extract_features(paths['tsv'], paths['fasta'], r2q, 6, 0, 0.0, None, 'NN', 0, endline=size, train=True,
                                       pos_label=pos_label, base='A', motif=None, positions_list=posfile)
    want_rows, want_lines = open(tmp, 'rb').read(), [l for l in buf.getvalue().splitlines() if 'observations' in l or 'regions' in l]
    assert want_rows.count(b'\n') > 300 and sum(len(v) for v in sig['general'].values()) == want_rows.count(b'\n')
    for n in (2, 3):
        os.remove(tmp)
        os.environ['MCALLER_SHARD_DEVICES'] = ','.join(['0'] * n)
        os.environ['MCALLER_STREAM_SHARDS'] = '3'
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ok = multi_gpu.extract_features_sharded(paths['tsv'], paths['fasta'], r2q, 6, 0, 0.0, None, 'A', None, posfile, n,
                                                        fastq=paths['fastq'], train=True, pos_label=pos_label)
        finally:
            del os.environ['MCALLER_SHARD_DEVICES']
            del os.environ['MCALLER_STREAM_SHARDS']
        assert ok is True
        assert open(tmp, 'rb').read() == want_rows
        assert multi_gpu.train_dicts[0] == sig and multi_gpu.train_dicts[1] == ctx
        assert [l for l in buf.getvalue().splitlines() if 'observations' in l or 'regions' in l] == want_lines


def test_sharded_run_survives_a_read_name_that_comes_back(tmp_path):
    """`--gpus 3` over a file in which a read name occurs twice, far apart: the pieces stand (the name is on neither side of a
    cut in the sense of extract_contexts.cut_names), no fall-back to one GPU, bytes and counter lines equal the one-GPU run."""
    from mcaller_amd import synth, mCaller, multi_gpu, _lib
    codes = synth.genome(length=300000, seed=23)
    table, qual = synth.make_table(400000, seed=6, codes=codes, read_len=(1500, 6000))
    seg_read = table.seg_read.copy()
    seg_read[table.n_seg - 4] = seg_read[3]
    seg_read[table.n_seg // 2] = seg_read[8]
    table = _lib.Table(table.pos, None, None, table.event_idx, table.flags, table.seg_row_begin, seg_read, table.seg_contig, table.n_reads,
                       read_names=table.read_names, evmu=table.evmu)
    paths = synth.write_inputs(table, qual, codes, str(tmp_path))
    model = os.path.join(H.MODELS, 'r95_twobase_model_NN_6_m6A.npz')
    common = ['-m', 'GATC', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model]
    real, took, outs = multi_gpu.extract_features_sharded, [], []

    def spy(*a, **kw):
        took.append(real(*a, **kw))
        return took[-1]
    multi_gpu.extract_features_sharded = spy
    try:
        for extra in ([], ['--gpus', '3']):
            out_path = paths['tsv'][:-4] + '.diffs.6'
            if os.path.exists(out_path):
                os.remove(out_path)
            os.environ['MCALLER_SHARD_DEVICES'] = ','.join(['0'] * (int(extra[1]) if extra else 1))
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    mCaller.main(common + extra)
            finally:
                del os.environ['MCALLER_SHARD_DEVICES']
            outs.append((open(out_path, 'rb').read(), [l for l in buf.getvalue().split('\n') if 'observations' in l or 'positions' in l or 'regions' in l]))
    finally:
        multi_gpu.extract_features_sharded = real
    assert took == [True]                                        # the sharded run stood: no second pass on one GPU
    assert outs[0] == outs[1] and outs[0][0].count(b'\n') > 200
