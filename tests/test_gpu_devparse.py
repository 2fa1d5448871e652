"""The eventalign text parsed on the GPU (mc_ctx_parse_*) against the host parser (mc_parse.cpp), column for column, and the
records of a pass over the device-parsed table against the records over the host-parsed one."""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from mcaller_amd.device import Device
    return Device(0)


def both_parsers(dev, path, contigs, lo=0, hi=None):
    from mcaller_amd import _lib
    hi = os.path.getsize(path) if hi is None else hi
    host = _lib.parse_eventalign(path, lo, hi, contigs, exact_range=True)
    text = _lib.TextBlock(path, lo, hi)
    slot = dev.parse_begin(text, contigs, max(1024, (hi - lo) // 40))
    t = dev.parse_end(slot, text)
    return host, t, slot, text


def assert_same_table(dev, host, t, slot):
    assert t is not None, getattr(dev, 'parse_fallback_reason', '')
    assert t.n_rows == host.n_rows and t.n_seg == host.n_seg
    assert t.read_names == host.read_names
    assert t.unknown == host.unknown
    assert np.array_equal(t.seg_row_begin, host.seg_row_begin)
    assert np.array_equal(t.seg_read, host.seg_read) and np.array_equal(t.seg_contig, host.seg_contig)
    assert np.array_equal(t.flags, host.flags)
    pos, evmu, idx, fl = dev.fetch_columns(slot, t.n_rows)
    assert np.array_equal(pos, host.pos) and np.array_equal(idx, host.event_idx)
    assert np.array_equal(evmu, host.evmu) and np.array_equal(fl, host.flags)


def test_micro_cases_column_for_column(dev, tmp_path):
    """Every committed micro-case (header lines, short lines, unknown contigs, quirks): the same table from both parsers, or
    the device parser says the shard needs the host's."""
    from mcaller_amd.refmark import read_fasta
    n_same = n_host = 0
    for i, case in enumerate(H.micro_cases()):
        d = tmp_path / ('c%d' % i)
        d.mkdir()
        paths = H.materialise(case, str(d))
        contigs = [name for name, _ in read_fasta(paths['fasta'])]
        try:
            host, t, slot, text = both_parsers(dev, paths['tsv'], contigs)
        except Exception:                         # the host parser raises (a malformed number): the device must decline
            from mcaller_amd import _lib
            text = _lib.TextBlock(paths['tsv'], 0, os.path.getsize(paths['tsv']))
            slot = dev.parse_begin(text, contigs, 4096)
            assert dev.parse_end(slot, text) is None
            n_host += 1
            continue
        if t is None:
            n_host += 1
            continue
        assert_same_table(dev, host, t, slot)
        dev.parse_abandon(slot)
        n_same += 1
    print('device parser: %d micro-cases column for column, %d left to the host parser' % (n_same, n_host))
    assert n_same > 100, (n_same, n_host)


def test_synthetic_shard_and_records(dev, tmp_path):
    """A synthetic file (10^5 rows, header line in front, a few malformed lines mixed in): same table; and a pass over the
    device-parsed table gives the records of a pass over the host-parsed one, bit for bit."""
    from mcaller_amd import synth
    from mcaller_amd.extract_contexts import submodel_setup
    codes = synth.genome()
    table, qual = synth.make_table(100000, seed=77, codes=codes)
    tsv = str(tmp_path / 'syn.tsv')
    synth.write_tsv_native(table, codes, tsv)
    body = open(tsv, 'rb').read()
    lines = body.split(b'\n')
    lines.insert(0, b'contig\tposition\treference_kmer\tread_name\tstrand\tevent_index\tevent_level_mean\tevent_stdv\tevent_length'
                    b'\tmodel_kmer\tmodel_mean\tmodel_stdv\tstandardized_level')
    lines.insert(5000, b'too\tshort')
    lines.insert(7000, b'')
    lines.insert(9000, b'elsewhere\t5\tAAAAAA\tr\tt\t7\t80.00\t1.0\t0.001\tAAAAAA\t81.00\t1.0\t0.1')
    open(tsv, 'wb').write(b'\n'.join(lines))
    contigs = ['ecoli_syn']
    host, t, slot, text = both_parsers(dev, tsv, contigs)
    assert t.unknown == ['contig', 'elsewhere']
    assert_same_table(dev, host, t, slot)
    # records
    ref = synth.SynthRef(codes, motif='GATC')
    _, weights, _, soc = submodel_setup(H.load_modelset('r95'), 'A')
    dev.set_reference(ref.device_arrays())
    dev.set_mlp(weights, soc)
    q = np.asarray([qual[table.read_names.index(n)] for n in t.read_names], dtype=np.float64)
    dev.upload_table_async(t, q)
    dev.run_async(6, 0, 0.0)
    got = dev.wait().by_record()
    got = [np.array(getattr(got, f)[:got.n * (6 if f == 'feats' else 1)]) for f in ('feats', 'site_pos', 'site_seg', 'close_row', 'info', 'prob')]
    dev.upload_table_async(host.pinned(), q)
    dev.run_async(6, 0, 0.0)
    want = dev.wait().by_record()
    assert want.n > 50
    for a, f in zip(got, ('feats', 'site_pos', 'site_seg', 'close_row', 'info', 'prob')):
        assert np.array_equal(a, getattr(want, f)[:want.n * (6 if f == 'feats' else 1)], equal_nan=True), f


def test_number_forms_that_need_the_host(dev, tmp_path):
    """Exponents, more than four decimals, inf, a malformed integer: the device parser declines the shard."""
    row = 'c1\t{pos}\tAAAAAA\tread1\tt\t{idx}\t{ev}\t1.0\t0.001\tAAAAAA\t{mu}\t1.0\t0.1\n'
    for k, kw in enumerate([dict(pos='5', idx='7', ev='8.1e1', mu='80.0'), dict(pos='5', idx='7', ev='80.12345', mu='80.0'),
                            dict(pos='5', idx='7', ev='inf', mu='80.0'), dict(pos='5x', idx='7', ev='80.0', mu='80.0'),
                            dict(pos='5', idx='7', ev='80.0', mu='1234567890.5')]):
        p = str(tmp_path / ('n%d.tsv' % k))
        open(p, 'w').write(row.format(pos='4', idx='6', ev='79.5', mu='80.25') + row.format(**kw))
        from mcaller_amd import _lib
        text = _lib.TextBlock(p, 0, os.path.getsize(p))
        slot = dev.parse_begin(text, ['c1'], 1024)
        assert dev.parse_end(slot, text) is None, kw
    # ... and the forms it does take: sign, no fraction, no integer part, '\\r\\n', spaces for tabs, no final newline
    p = str(tmp_path / 'ok.tsv')
    open(p, 'wb').write(b'c1 4 AAAAAA read1 t 6 +79.5 1.0 0.001 AAAAAA 80.25 1.0 0.1\r\n'
                        b'c1\t5\tAAAAAA\tread1\tt\t-7\t.5\t1.0\t0.001\tNNNNNN\t0.\t1.0\t0.1\r\n'
                        b'c1\t6\tAAAAAC\tread2\tt\t8\t-80\t1.0\t0.001\tAAAAAA\t80.1234\t1.0\t0.1')
    host, t, slot, text = both_parsers(dev, p, ['c1'])
    assert_same_table(dev, host, t, slot)
    assert t.n_rows == 3 and t.read_names == ['read1', 'read2']
    dev.parse_abandon(slot)


def test_random_text_against_the_host_parser(dev, tmp_path):
    """Seeded random files: odd whitespace, dropped tokens, blank lines, every number form float()/int() accept or reject, unknown
    contigs, read names that come back, with and without a final newline.  Where the host parser raises the device parser must
    decline; where it declines nothing is claimed; everywhere else every column is the host parser's."""
    from mcaller_amd import _lib
    rng = np.random.default_rng(20261002)
    seps = ['\t', '\t', '\t', ' ', '  ', '\t ', '\x0b', '\x0c', '\x1c']
    ints = ['0', '7', '42', '+5', '-3', '123456', '2147483647', '2147483648', '12x', '', '1.0', '1e3', '٣']
    floats = ['80.5', '79.25', '0.1234', '.5', '5.', '+81.0', '-3.75', '100', '80.12345', '8.05e1', 'inf', 'nan', '1_0.5', '0x1p3',
              '999999999.9999', '1234567890.1', '--1', '80,5']
    kmers = ['AAAAAA', 'ACGTAC', 'NNNNNN', 'acgtac', 'AAAAA', 'AAAAAAA']
    contigs = ['c1', 'c2', 'chr_long_name.1', 'nope', 'C1', 'c1x']
    names = ['r%d' % i for i in range(6)] + ['read-with-a-much-longer-name_0123456789abcdef', 'r0']
    n_same = n_declined = n_raise = 0
    for case in range(300):
        n_lines = int(rng.integers(1, 400))
        mode = case % 3                                 # 0: only forms the fast path takes; 1: anything; 2: odd but valid forms
        weird = mode > 0
        ints_m = ints if mode == 1 else ['0', '7', '+5', '-3', '2147483647', '00012']
        floats_m = floats if mode == 1 else ['.5', '5.', '+81.0', '-3.75', '100', '80.12345', '8.05e1', '1E2', '0080.50']
        lines, name = [], names[0]
        for i in range(n_lines):
            if rng.random() < 0.1:
                name = names[int(rng.integers(len(names)))]
            u = rng.random()
            if weird and u < 0.03:
                lines.append('')
                continue
            if weird and u < 0.06:
                lines.append(seps[int(rng.integers(len(seps)))].join(['c1', '5', 'AAAAAA'][:int(rng.integers(1, 4))]))
                continue
            pick_i = (lambda: ints_m[int(rng.integers(len(ints_m)))]) if weird and rng.random() < 0.05 else (lambda: str(int(rng.integers(0, 5000))))
            pick_f = (lambda: floats_m[int(rng.integers(len(floats_m)))]) if weird and rng.random() < 0.05 else (lambda: '%.2f' % rng.uniform(50, 120))
            k1, k2 = kmers[int(rng.integers(len(kmers)))] if weird else 'ACGTAC', kmers[int(rng.integers(3))]
            contig = contigs[int(rng.integers(len(contigs)))] if rng.random() < (0.15 if weird else 0.02) else 'c1'
            tok = [contig, pick_i(), k1, name, 't', pick_i(), pick_f(), '1.0', '0.001', k2, pick_f(), '1.0', '0.1']
            if mode == 1 and rng.random() < 0.02:
                tok = tok[:int(rng.integers(10, 13))]
            if weird and rng.random() < 0.02:
                tok += ['extra', 'tokens']
            sep = (lambda: seps[int(rng.integers(len(seps)))]) if weird else (lambda: '\t')
            line = ''.join(t + sep() for t in tok[:-1]) + tok[-1]
            if weird and rng.random() < 0.05:
                line = ' ' + line + ' \r'
            lines.append(line)
        body = '\n'.join(lines) + ('\n' if rng.random() < 0.8 else '')
        p = str(tmp_path / ('r%d.tsv' % case))
        open(p, 'w', encoding='utf-8').write(body)
        size = os.path.getsize(p)
        text = _lib.TextBlock(p, 0, size)
        slot = dev.parse_begin(text, ['c1', 'c2', 'chr_long_name.1', 'c1'], 4096)
        t = dev.parse_end(slot, text)
        try:
            host = _lib.parse_eventalign(p, 0, size, ['c1', 'c2', 'chr_long_name.1', 'c1'], exact_range=True)
        except _lib.McError:
            assert t is None, 'case %d: the host parser raises, the device parser did not decline' % case
            n_raise += 1
            continue
        if t is None:
            n_declined += 1
            continue
        assert_same_table(dev, host, t, slot)
        dev.parse_abandon(slot)
        n_same += 1
    print('random text: %d files column for column, %d declined, %d malformed' % (n_same, n_declined, n_raise))
    assert n_same >= 100 and n_raise >= 10 and n_declined >= 10


def test_lines_too_long_for_the_staging_buffer(dev, tmp_path):
    """A workgroup's 256 lines that do not fit the 48 KB LDS stage are read in place (a 13th token of 60 KB: ignored by both
    parsers); a read name longer than 65535 bytes is left to the host parser."""
    row = 'c1\t{pos}\tAAAAAA\tread{r}\tt\t{pos}\t80.5\t1.0\t0.001\tAAAAAA\t81.25\t1.0\t0.1'
    lines = [row.format(pos=i, r=i // 40) for i in range(700)]
    lines[300] += '\t' + 'x' * 60000
    lines[301] += ' ' + 'y' * 70000 + ' z'
    p = str(tmp_path / 'long.tsv')
    open(p, 'w').write('\n'.join(lines) + '\n')
    host, t, slot, text = both_parsers(dev, p, ['c1'])
    assert_same_table(dev, host, t, slot)
    assert t.n_rows == 700
    dev.parse_abandon(slot)
    lines[10] = row.format(pos=10, r=0).replace('read0', 'r' * 70000)
    open(p, 'w').write('\n'.join(lines) + '\n')
    from mcaller_amd import _lib
    text = _lib.TextBlock(p, 0, os.path.getsize(p))
    slot = dev.parse_begin(text, ['c1'], 4096)
    assert dev.parse_end(slot, text) is None
    host = _lib.parse_eventalign(p, 0, os.path.getsize(p), ['c1'], exact_range=True)       # (the host parser takes it)
    assert host.n_rows == 700 and max(len(n) for n in host.read_names) == 70000
