"""Reference marking on the host: the pre-pass that builds what the kernels read.

Mirrors extract_contexts.py:33-81 (methylate_motifs / methylate_positions / methylate_references /
find_and_methylate): per contig, two copies of the upper-cased sequence in which the target bases are
replaced by 'M' -- `meth_fwd` (motif, base) and `meth_rev` (revcomp(motif), complement base), or the
positions listed for '+' / '-' in the positions file.  The strings are kept (the 2k-1 context of every
call is sliced from them, :194) and turned into the per-strand bitmasks of `mc_ref_view`.
"""
import os
import sys

import numpy as np

base_comps = {'A': 'T', 'C': 'G', 'T': 'A', 'G': 'C', 'N': 'N', 'M': 'M'}     # extract_contexts.py:11


def comp(seq):
    return ''.join([base_comps[nt] for nt in seq])


def revcomp(seq, rev=True):
    if not rev:
        return seq
    return comp(seq)[::-1]


def strand(rev):
    return '-' if rev else '+'


_WS = bytes(bytearray([9, 10, 11, 12, 13, 32]))
_fasta_cache = {}


def read_fasta(path):
    """[(id, sequence)] in file order; id = first token of the title line (what Bio.SeqIO yields).  Whole-file
    bytes operations (a 4.6 MB genome in a few ms); the last file read is cached by (path, size, mtime): the CLI
    reads the reference twice, like the reference does (mCaller.py:176, extract_contexts.py:77)."""
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    if key in _fasta_cache:
        return _fasta_cache[key]
    with open(path, 'rb') as fh:
        data = fh.read()
    records = []
    # records start at a '>' in column 0; text before the first one is ignored
    starts = [0] if data[:1] == b'>' else []
    i = data.find(b'\n>')
    while i >= 0:
        starts.append(i + 1)
        i = data.find(b'\n>', i + 1)
    for j, a in enumerate(starts):
        b = starts[j + 1] if j + 1 < len(starts) else len(data)
        nl = data.find(b'\n', a, b)
        if nl < 0:
            nl = b
        title = data[a + 1:nl].decode('latin1').rstrip()
        parts = title.split(None, 1)
        seq = data[nl:b].translate(None, _WS).decode('latin1')
        records.append((parts[0] if parts else '', seq))
    _fasta_cache.clear()
    _fasta_cache[key] = records
    return records


def methylate_motifs(ref_seq, motif, meth_base):
    """extract_contexts.py:33-41 (meth_position=None): left-to-right, non-overlapping."""
    return ref_seq.replace(motif, 'M'.join(motif.split(meth_base)))


def methylate_positions(ref_seq, positions, meth_base, quiet=False):
    """extract_contexts.py:45-56; prints and exits like the reference on a wrong base (quiet: exits only -- a caller that
    will go through the same contig again on another path lets that path do the printing)."""
    buf = bytearray(ref_seq, 'latin1')
    b, m = ord(meth_base), ord('M')
    for pos in positions:
        if pos < 0:
            raise NotImplementedError('negative position %d in the positions file' % pos)
        if buf[pos] == b or buf[pos] == m:          # IndexError past the contig end, like the reference
            buf[pos] = m
        else:
            if not quiet:
                print('Base {} does not correspond to methylated base - check reference positions are 0-based'
                      ' - quitting thread now'.format(pos))
            sys.exit(0)
    return buf.decode('latin1')


def _positions_for(positions, contig, strand_char):
    out = []
    for line in open(positions, 'r').read().split('\n'):
        t = line.split()
        if len(t) > 1 and t[2] == strand_char and t[0] == contig:
            out.append(int(t[1]))
    return out


def methylate_references(ref_seq, base, motif=None, positions=None, train=False, contig=None, quiet=False):
    """extract_contexts.py:60-73 -> (meth_fwd, meth_rev)."""
    if not positions and motif:
        meth_fwd = methylate_motifs(ref_seq, motif, base)
        meth_rev = methylate_motifs(ref_seq, revcomp(motif), base_comps[base])
    elif positions:
        meth_fwd = methylate_positions(ref_seq, _positions_for(positions, contig, '+'), base, quiet)
        meth_rev = methylate_positions(ref_seq, _positions_for(positions, contig, '-'), base_comps[base], quiet)
    else:
        if not quiet:
            print('no motifs or positions specified')
        sys.exit(0)
    return meth_fwd, meth_rev


def m_bitmask(meth, pad_words=2):
    """bit p of the little-endian u32 array is set <=> meth[p] == 'M'; zero padding at the end."""
    raw = np.frombuffer(meth.encode('latin1'), dtype=np.uint8)
    packed = np.packbits(raw == ord('M'), bitorder='little')
    n_words = (len(meth) + 31) // 32 + pad_words
    out = np.zeros(n_words * 4, dtype=np.uint8)
    out[:len(packed)] = packed
    return out.view('<u4')


class MarkedReference(object):
    """Contig table + marked strings for the contigs a table touches (marked on first use, in the
    order the reference would load them: extract_contexts.py:154-157)."""

    def __init__(self, fasta_path, base, motif, positions_list):
        self.records = read_fasta(fasta_path)
        self.names = [r[0] for r in self.records]
        self.base, self.motif, self.positions_list = base, motif, positions_list
        self.meth = {}                                  # contig id -> (meth_fwd, meth_rev)
        self.quiet = False                              # the exit paths of the marking do not print
        self._arrays = (None, None)
        self._upper = {}
        self._upper_bytes = {}
        import threading
        self._lock = threading.RLock()

    def first_index(self):
        idx = {}
        for i, n in enumerate(self.names):
            idx.setdefault(n, i)                        # first record with that id wins (:77-81)
        return idx

    def upper(self, contig_id):
        """The contig's sequence in upper case (:79), made once."""
        if contig_id not in self._upper:
            if contig_id in self._upper_bytes:
                self._upper[contig_id] = str(memoryview(self._upper_bytes[contig_id]), 'ascii')
            else:
                self._upper[contig_id] = self.records[contig_id][1].upper()
        return self._upper[contig_id]

    def _mark_motif_native(self, contig_id):
        """Motif mode on a long ASCII contig: upper-casing and the two replacements in the library, without the interpreter
        lock (mc_mark_motifs; the same strings as `methylate_references(seq.upper(), ...)`, tests/test_host_pipeline.py)."""
        seq = self.records[contig_id][1]
        motif_f, motif_r = self.motif, revcomp(self.motif)
        repl_f, repl_r = 'M'.join(motif_f.split(self.base)), 'M'.join(motif_r.split(base_comps[self.base]))
        if len(seq) < (1 << 16) or not seq.isascii() or not (motif_f + motif_r).isascii() or len(repl_f) != len(motif_f) \
                or len(repl_r) != len(motif_r):
            return
        try:
            import ctypes
            from . import _lib
            L = _lib.lib()
        except (ImportError, OSError):
            return
        raw, n = seq.encode('ascii'), len(seq)
        bufs = [np.empty(n, dtype=np.uint8) for _ in range(3)]        # (not zero-filled: every byte is written)
        _lib.check(L.mc_mark_motifs(raw, n, motif_f.encode('ascii'), repl_f.encode('ascii'), len(motif_f), motif_r.encode('ascii'),
                                    repl_r.encode('ascii'), len(motif_r), bufs[0].ctypes.data, bufs[1].ctypes.data,
                                    bufs[2].ctypes.data))
        self._upper_bytes[contig_id] = bufs[0]                        # (what device_arrays sends; the string is made on demand)
        self.meth[contig_id] = (str(memoryview(bufs[1]), 'ascii'), str(memoryview(bufs[2]), 'ascii'))

    def mark(self, contig_id):
        with self._lock:                                # (a streamed file marks its first contig ahead of time, in a thread)
            if contig_id not in self.meth and self.motif and not self.positions_list:
                self._mark_motif_native(contig_id)
            if contig_id not in self.meth:
                name, seq = self.records[contig_id]
                self.meth[contig_id] = methylate_references(self.upper(contig_id), self.base, motif=self.motif,
                                                            positions=self.positions_list, contig=name, quiet=self.quiet)
            return self.meth[contig_id]

    def motif_for_the_device(self):
        """(motif_fwd, repl_fwd, motif_rev, repl_rev) as bytes if the device can make the site masks itself
        (mc_ctx_set_reference_motif): motif mode, ASCII, at most 16 bases, and neither motif can overlap itself (no proper
        prefix is also a suffix -- str.replace's left-to-right, non-overlapping rule is then "every occurrence"); else None."""
        if not self.motif or self.positions_list:
            return None
        try:
            motif_f, motif_r = self.motif, revcomp(self.motif)
            repl_f, repl_r = 'M'.join(motif_f.split(self.base)), 'M'.join(motif_r.split(base_comps[self.base]))
        except KeyError:                                # (a letter revcomp does not know: the reference's own crash, elsewhere)
            return None
        for m, r in ((motif_f, repl_f), (motif_r, repl_r)):
            if not (1 <= len(m) <= 16) or len(r) != len(m) or not m.isascii() or any(m[:i] == m[-i:] for i in range(1, len(m))):
                return None
        if not all(seq.isascii() for _, seq in self.records):
            return None
        return motif_f.encode('ascii'), repl_f.encode('ascii'), motif_r.encode('ascii'), repl_r.encode('ascii')

    def raw_arrays(self):
        """The arrays of mc_ref_view for mc_ctx_set_reference_motif: the raw bases of EVERY contig, laid out as device_arrays
        would lay them out with every contig marked (mask words: ceil(len / 32) + 2 per contig)."""
        n = len(self.records)
        lens = np.array([len(seq) for _, seq in self.records], dtype=np.int64)
        words = (lens + 31) // 32 + 2
        seq_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
        word_off = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
        raw = np.frombuffer(''.join(seq for _, seq in self.records).encode('ascii') + b'\0' * 8, dtype=np.uint8)
        return dict(contig_len=lens, seq_off=seq_off, word_off=word_off, seq=raw, mbits_fwd=np.zeros(2, np.uint32),
                    mbits_rev=np.zeros(2, np.uint32), n_words=int(words.sum()), n_seq_bytes=int(lens.sum()))

    def device_arrays(self):
        """Concatenated arrays for mc_ref_view (unmarked contigs: empty sequence, all-zero masks)."""
        key = tuple(sorted(self.meth))
        if self._arrays[0] == key:
            return self._arrays[1]
        out = self._device_arrays()
        self._arrays = (key, out)
        return out

    def _device_arrays(self):
        n = len(self.records)
        contig_len = np.zeros(n, dtype=np.int64)
        seq_off = np.zeros(n, dtype=np.int64)
        word_off = np.zeros(n, dtype=np.int64)
        seqs, fw, rv = [], [], []
        so = wo = 0
        for cid in range(n):
            seq_off[cid], word_off[cid] = so, wo
            if cid in self.meth:
                mf, mr = self.meth[cid]
                s = self._upper_bytes[cid] if cid in self._upper_bytes else np.frombuffer(self.upper(cid).encode('latin1'), dtype=np.uint8)
                contig_len[cid] = len(s)
                bf, br = m_bitmask(mf), m_bitmask(mr)
                if len(bf) != len(br):                  # cannot happen: both come from one sequence
                    raise AssertionError('marked strands differ in length')
            else:
                s = np.zeros(0, dtype=np.uint8)
                bf = br = np.zeros(2, dtype='<u4')
            seqs.append(s)
            fw.append(bf)
            rv.append(br)
            so += len(s)
            wo += len(bf)
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0, dt), dtype=dt)
        return dict(contig_len=contig_len, seq_off=seq_off, word_off=word_off,
                    seq=cat(seqs + [np.zeros(8, np.uint8)], np.uint8),
                    mbits_fwd=cat(fw, np.uint32), mbits_rev=cat(rv, np.uint32))
