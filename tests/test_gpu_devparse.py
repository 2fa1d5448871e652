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
