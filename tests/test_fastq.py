"""Native FASTQ quality reader (mc_fastq_read_quality) against the plain-Python statement of read_qual.py:6-19."""
import gzip
import os

import numpy as np
import pytest

from mcaller_amd import _lib
from mcaller_amd.read_qual import extract_read_quality, extract_read_quality_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _same(path, n_threads=0):
    a = extract_read_quality(path, n_threads)
    b = extract_read_quality_py(path)
    assert list(a.keys()) == list(b.keys())
    for key in a:
        x, y = a[key], b[key]
        assert (np.isnan(x) and np.isnan(y)) or (x == y and type(x) is type(y)), key
    return a


def test_reference_fastq():
    path = os.path.join(HERE, 'golden', 'testdata', 'masonread1.fastq')
    if not os.path.exists(path):
        pytest.skip('fixture not present')
    got = _same(path)
    import json
    ref = json.load(open(os.path.join(HERE, 'golden', 'ref_outputs', 'read2qual.json')))['read2qual']   # the reference's own dict
    assert {k: repr(float(v)) for k, v in got.items()} == ref


def _random_fastq(rng, n, crlf=False, at_quality=False, blank_every=0, dup=False):
    nl = '\r\n' if crlf else '\n'
    out = []
    for i in range(n):
        length = int(rng.integers(0 if i % 97 == 5 else 1, 400))
        name = 'read%d' % (i // 2 if dup else i)
        title = '@%s_Basecall_2D_template:extra ch=%d' % (name, i) if i % 3 else '@%s:x runid=7' % name
        seq = ''.join('ACGT'[c] for c in rng.integers(0, 4, length))
        qual = ''.join(chr(33 + int(q)) for q in rng.integers(0, 60, length))
        if at_quality and length:
            qual = '@' + qual[1:]                             # phred 31: a quality line that looks like a title
        if blank_every and i % blank_every == 0:
            out.append(nl)
        out.append(title + nl + seq + nl + '+' + nl + qual + nl)
    return ''.join(out)


@pytest.mark.parametrize('kind', ['plain', 'crlf', 'at_quality', 'blank', 'dup'])
def test_random_files(tmp_path, kind):
    rng = np.random.default_rng(7)
    text = _random_fastq(rng, 3000, crlf=kind == 'crlf', at_quality=kind == 'at_quality',
                         blank_every=7 if kind == 'blank' else 0, dup=kind == 'dup')
    path = str(tmp_path / 'reads.fastq')
    with open(path, 'w', newline='') as f:
        f.write(text)
    one = _same(path, 1)
    for n_threads in (2, 5):
        keys, means = _lib.fastq_read_quality(path, n_threads)
        assert dict(zip(keys, means)).keys() == one.keys()
        assert len(keys) == 3000


def test_many_pieces(tmp_path):
    """Large enough that the file is really cut (pieces of at least 4 MB), every quality line starting with '@'."""
    rng = np.random.default_rng(11)
    text = _random_fastq(rng, 60000, at_quality=True)
    assert len(text) > 5 * (1 << 22)
    path = str(tmp_path / 'big.fastq')
    with open(path, 'w') as f:
        f.write(text)
    k1, m1 = _lib.fastq_read_quality(path, 1)
    k8, m8 = _lib.fastq_read_quality(path, 5)
    assert k1 == k8 and np.array_equal(m1, m8, equal_nan=True)
    assert len(k1) == 60000
    _same(path, 5)


def test_gzip_and_no_final_newline(tmp_path):
    rng = np.random.default_rng(3)
    text = _random_fastq(rng, 500).rstrip('\n')
    gz = str(tmp_path / 'reads.fastq.gz')
    with gzip.open(gz, 'wt') as f:
        f.write(text)
    plain = str(tmp_path / 'reads.fastq')
    with open(plain, 'w') as f:
        f.write(text)
    assert _same(gz) == _same(plain) or all(np.isnan(v) for v in _same(gz).values() if v != v)


def test_empty_file(tmp_path):
    path = str(tmp_path / 'empty.fastq')
    open(path, 'w').close()
    assert extract_read_quality(path) == {} == extract_read_quality_py(path)


@pytest.mark.parametrize('text,needle', [
    ('read1\nACGT\n+\nIIII\n', "should start with '@'"),
    ('@read1\nACGT\nACGT\n+\nIIIIIIII\n', 'multi-line'),
    ('@read1\nACGT\n+\nIII\n', 'Lengths of sequence and quality'),
    ('@read1\nACGT\n', 'multi-line'),
])
def test_errors(tmp_path, text, needle):
    path = str(tmp_path / 'bad.fastq')
    with open(path, 'w') as f:
        f.write('@ok\nAC\n+\nII\n' + text)
    for fn in (extract_read_quality, extract_read_quality_py):
        with pytest.raises(ValueError) as e:
            fn(path)
        assert needle in str(e.value)
