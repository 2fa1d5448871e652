"""FASTQ -> {read id: mean phred}; same contract as the reference's read_qual.py:6-19.

Keys are `id.split(':')[0].split('_')[0]` (read_qual.py:11-12), values np.float64 means of the
per-base phred scores (exact integer sum / n).  `.gz` anywhere in the file name selects gzip, like
the reference's `fastqfi.find(".gz")` test.

`extract_read_quality` reads the file with the native multi-threaded reader (mc_fastq_read_quality, csrc/mc_fastq.cpp);
`extract_read_quality_py` is the same contract in plain Python, kept as the statement the native reader is tested against.
"""
import gzip

import numpy as np


def _records(handle):
    while True:
        title = handle.readline()
        if not title:
            return
        if not title.strip():
            continue
        if not title.startswith('@'):
            raise ValueError("Records in Fastq files should start with '@' character")
        seq = handle.readline()
        plus = handle.readline()
        if not plus.startswith('+'):
            raise ValueError('multi-line FASTQ records are not supported')
        qual = handle.readline().rstrip('\n').rstrip('\r')
        if len(qual) != len(seq.strip()):
            raise ValueError('Lengths of sequence and quality values differs for %s' % title.strip())
        yield title[1:].split(None, 1)[0], qual


def extract_read_quality(fastqfi, n_threads=0):
    from . import _lib
    keys, means = _lib.fastq_read_quality(fastqfi, n_threads)
    return dict(zip(keys, means))                 # a later record of the same key replaces the earlier one (:12)


def extract_read_quality_py(fastqfi):
    read2qual = {}
    opener = (lambda: gzip.open(fastqfi, 'rt')) if fastqfi.find('.gz') != -1 else (lambda: open(fastqfi, 'r'))
    with opener() as handle:
        for rid, qual in _records(handle):
            rid = rid.split(':')[0].split('_')[0]
            phred = np.frombuffer(qual.encode('latin1'), dtype=np.uint8).astype(np.int64) - 33
            read2qual[rid] = np.float64(int(phred.sum())) / np.float64(len(phred)) if len(phred) else np.float64('nan')
    return read2qual
