"""The rows of streamed shards written on the GPU (mc_rowtext.hip, mc_ctx_row_text) against the host formatter's (mc_format_diffs,
pinned to the reference's own rows by tests/test_host_pipeline.py and the CLI tests): the same bytes, whichever of the two made a
shard's rows -- and the passes the device hands back to the host (a context that leaves its contig, rows that do not fit, no free
block) are the ones that should be."""
import contextlib
import io
import os
import random

import pytest

pytestmark = pytest.mark.gpu

MODEL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}


def write_case(d, seed, n_reads=36, decimals=(2, 2, 2, 4), edge_reads=True):
    """A small eventalign file over three contigs: reads on both strands, some starting at position 0 / ending at the contig's
    last 6-mer (their first windows' contexts leave the contig: the reference's slicing, left to the host), skipped positions,
    NNNNNN rows, event means of two or four decimals, FASTQ qualities whose means have many digits."""
    rng = random.Random(seed)
    contigs = [('contig_1', 5200), ('plasmid.2', 3100), ('c3', 4400)]
    seqs = {name: ''.join(rng.choice('ACGT') for _ in range(L)) for name, L in contigs}
    paths = dict(tsv=os.path.join(d, 'case.eventalign.tsv'), fasta=os.path.join(d, 'ref.fasta'), fastq=os.path.join(d, 'reads.fastq'))
    with open(paths['fasta'], 'w') as fa:
        for name, _ in contigs:
            s = seqs[name]
            fa.write('>%s\n' % name + '\n'.join(s[i:i + 60] for i in range(0, len(s), 60)) + '\n')
    model = {}
    rows = 0
    with open(paths['tsv'], 'w') as out, open(paths['fastq'], 'w') as fq:
        out.write('contig\tposition\treference_kmer\tread_index\tstrand\tevent_index\tevent_level_mean\tevent_stdv\tevent_length\t'
                  'model_kmer\tmodel_mean\tmodel_stdv\tstandardized_level\n')
        order = sorted(range(n_reads), key=lambda i: (i * 3 // n_reads, rng.random()))        # contig by contig, like a sorted BAM
        for i in order:
            name, L = contigs[i * 3 // n_reads]
            seq = seqs[name]
            length = rng.randint(300, 1400)
            kind = rng.random()
            if edge_reads and kind < 0.12:
                s = 0
            elif edge_reads and kind < 0.24:
                s = L - 6 - length + 1
            else:
                s = rng.randint(0, L - 6 - length)
            rev = rng.random() < 0.5
            dec = rng.choice(decimals)
            read = 'read-%04d-%08x_Basecall_2D_template' % (i, rng.getrandbits(32))
            fq.write('@%s\nACGTACGTACGT\n+\n%s\n' % (read, ''.join(chr(33 + rng.randint(3, 40)) for _ in range(12))))
            positions = list(range(s, s + length))
            n_ev = [rng.choices((0, 1, 2, 3, 4, 7), (0.08, 0.5, 0.25, 0.1, 0.05, 0.02))[0] for _ in positions]
            total = sum(n_ev)
            idx = 1000 + (total if rev else 0)
            for p, ne in zip(positions, n_ev):
                ref_kmer = seq[p:p + 6]
                for _ in range(ne):
                    mk = ref_kmer if not rev else ''.join(COMP[c] for c in reversed(ref_kmer))
                    mu = model.setdefault(mk, round(rng.uniform(55.0, 117.0), 2))
                    is_n = rng.random() < 0.04
                    ev = mu + rng.gauss(-0.17, 2.44)
                    out.write('%s\t%d\t%s\t%s\tt\t%d\t%.*f\t1.500\t0.00200\t%s\t%.2f\t1.50\t0.10\n' % (
                        name, p, ref_kmer, read, idx, dec, ev, 'NNNNNN' if is_n else mk, 0.0 if is_n else mu))
                    idx += -1 if rev else 1
                    rows += 1
    return paths, rows


def run_cli(paths, motif, env):
    from mcaller_amd import mCaller, extract_contexts as ec
    keys = ('MCALLER_NO_STREAM', 'MCALLER_STREAM_SHARDS', 'MCALLER_DEVICE_ROWS', 'MCALLER_ROW_TEXT_ROOM', 'MCALLER_HOST_PARSER')
    saved = {k: os.environ.pop(k, None) for k in keys}
    os.environ.update(env)
    out = paths['tsv'][:-4] + '.diffs.6'
    if os.path.exists(out):
        os.remove(out)
    ec.stream_features.last_clock = None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mCaller.main(['-m', motif, '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', MODEL])
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    clock = ec.stream_features.last_clock or {}
    return open(out, 'rb').read(), clock.get('device_rows', 0), clock.get('shards', 0)


# ('AA' can overlap itself: its masks are made by the host contig by contig as the contigs first appear, and every new contig's upload
# waits until no pass is in flight -- the moments at which a shard's rows may still be with the host's writer while nothing is in flight)
@pytest.mark.parametrize('motif', ['A', 'GATC', 'AA'])
def test_rows_written_on_the_device_are_the_host_formatters(tmp_path, motif):
    with_text = without = 0
    for seed in (11, 12, 13):
        d = str(tmp_path / ('case%d' % seed))
        os.makedirs(d)
        paths, rows = write_case(d, seed)
        want, _, _ = run_cli(paths, motif, {'MCALLER_NO_STREAM': '1'})
        assert want.count(b'\n') > (500 if motif == 'A' else 5 if motif == 'GATC' else 50)
        for shards in ('3', '8'):
            host, n_dev0, n0 = run_cli(paths, motif, {'MCALLER_STREAM_SHARDS': shards, 'MCALLER_DEVICE_ROWS': '0'})
            dev, n_dev, n = run_cli(paths, motif, {'MCALLER_STREAM_SHARDS': shards})
            assert n0 >= 2 and n >= 2 and n_dev0 == 0
            assert host == want
            assert dev == want
            with_text += n_dev
            without += n - n_dev
    assert with_text > 0
    if motif == 'A':
        assert without > 0      # (every A is a site: the reads that start at position 0 have windows whose context leaves the contig)


def test_every_shard_on_the_device_when_no_context_leaves_a_contig(tmp_path):
    d = str(tmp_path)
    paths, rows = write_case(d, 5, edge_reads=False, decimals=(2,))
    want, _, _ = run_cli(paths, 'A', {'MCALLER_NO_STREAM': '1'})
    dev, n_dev, n = run_cli(paths, 'A', {'MCALLER_STREAM_SHARDS': '6'})
    assert dev == want
    assert n >= 2 and n_dev == n


def test_rows_that_do_not_fit_their_room_come_from_the_host_and_the_room_grows(tmp_path):
    d = str(tmp_path)
    paths, rows = write_case(d, 7, edge_reads=False, decimals=(2,))
    want, _, _ = run_cli(paths, 'A', {'MCALLER_NO_STREAM': '1'})
    dev, n_dev, n = run_cli(paths, 'A', {'MCALLER_STREAM_SHARDS': '6', 'MCALLER_ROW_TEXT_ROOM': '8'})
    assert dev == want
    assert 0 < n_dev < n                # (the first shard's rows did not fit eight bytes a row: the host's; the room the next ones got was enough)


def test_qualities_that_are_not_floats_are_left_to_the_host(tmp_path):
    """str(quality) on the device is repr of a double: what the FASTQ reader returns.  A caller's dict of ints (str -> '9', not
    '9.0') or of numpy doubles (str the same as repr) must come out as the reference writes them: the first from the host
    formatter, the second from the device."""
    import numpy as np
    from mcaller_amd import extract_contexts as ec
    from mcaller_amd.read_qual import extract_read_quality
    d = str(tmp_path)
    paths, rows = write_case(d, 9, edge_reads=False, decimals=(2,))
    r2q = extract_read_quality(paths['fastq'])
    size = os.path.getsize(paths['tsv'])
    tmp = paths['tsv'][:-4] + '.diffs.6.tmp0'

    def run(qual, env):
        saved = {k: os.environ.pop(k, None) for k in ('MCALLER_NO_STREAM', 'MCALLER_STREAM_SHARDS', 'MCALLER_DEVICE_ROWS')}
        os.environ.update(env)
        ec.stream_features.last_clock = None
        try:
            if os.path.exists(tmp):
                os.remove(tmp)
            with contextlib.redirect_stdout(io.StringIO()):
                ec.extract_features(paths['tsv'], paths['fasta'], qual, 6, 0, 0.0, MODEL, 'NN', 0, endline=size, train=False, base='A', motif='A')
        finally:
            for k in saved:
                os.environ.pop(k, None)
                if saved[k] is not None:
                    os.environ[k] = saved[k]
        clock = ec.stream_features.last_clock or {}
        return open(tmp, 'rb').read(), clock.get('device_rows', 0), clock.get('shards', 0)

    ints = {name: int(q) + 3 for name, q in r2q.items()}
    want, _, _ = run(ints, {'MCALLER_NO_STREAM': '1'})
    got, n_dev, n = run(ints, {'MCALLER_STREAM_SHARDS': '5'})
    assert got == want and n >= 2 and n_dev == 0
    assert b',12\t' in want or b',11\t' in want or b',10\t' in want or any(b',%d\t' % v in want for v in set(ints.values()))
    npq = {name: np.float64(q) for name, q in r2q.items()}
    want, _, _ = run(npq, {'MCALLER_NO_STREAM': '1'})
    got, n_dev, n = run(npq, {'MCALLER_STREAM_SHARDS': '5'})
    assert got == want and n_dev == n
    floats, _, _ = run(r2q, {'MCALLER_STREAM_SHARDS': '5'})
    assert floats == want


def test_two_times_ten_to_the_seven_rows_of_a_one_base_motif(tmp_path):
    """The streamed dense mode at size (2 x 10^7 rows, 2.3 GB of text, 1.8 x 10^6 rows out in the default shard schedule): every
    shard's rows from the GPU, the file the host formatter's byte for byte."""
    import hashlib
    from mcaller_amd import synth
    d = str(tmp_path)
    codes = synth.genome()
    table, qual = synth.make_table(20000000, seed=23, codes=codes)
    paths = synth.write_inputs(table, qual, codes, d)
    del table
    host, n_dev0, n0 = run_cli(paths, 'A', {'MCALLER_DEVICE_ROWS': '0'})
    dev, n_dev, n = run_cli(paths, 'A', {})
    assert n0 == n and n >= 10 and n_dev0 == 0 and n_dev == n
    assert len(dev) == len(host) and dev.count(b'\n') > 1500000
    assert hashlib.sha256(dev).digest() == hashlib.sha256(host).digest()


def test_shards_that_find_no_free_block_come_from_the_host(tmp_path):
    """One pinned block instead of six (MCALLER_ROW_TEXT_BLOCKS, read once per process: a process of its own): a shard whose rows are
    ready while the block is still with the writer gets none and is formatted by the host -- the file is the same."""
    import subprocess
    import sys
    d = str(tmp_path)
    paths, rows = write_case(d, 21, n_reads=48, edge_reads=False, decimals=(2,))
    want, _, _ = run_cli(paths, 'A', {'MCALLER_NO_STREAM': '1'})
    out = paths['tsv'][:-4] + '.diffs.6'
    os.remove(out)
    env = dict(os.environ, MCALLER_ROW_TEXT_BLOCKS='1', MCALLER_STREAM_SHARDS='12', MCALLER_VERBOSE='1')
    for k in ('MCALLER_NO_STREAM', 'MCALLER_DEVICE_ROWS'):
        env.pop(k, None)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'mcaller_amd.mCaller', '-m', 'A', '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', MODEL],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out, 'rb').read() == want
    said = [l for l in r.stderr.splitlines() if 'rows written on the device for' in l]
    assert said, r.stderr[-2000:]
    n_text, n_no_block = int(said[-1].split(' for ')[1].split()[0]), int(said[-1].split('not for ')[1].split()[0])
    assert n_text >= 1 and n_text + n_no_block == 12
