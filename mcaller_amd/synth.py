"""Seeded synthetic eventalign workloads, generated directly in columnar form.

Shaped on the statistics of the reference's test read: one contig of 4,641,652 random bases (~18k GATC sites);
reads of U(5,20) kb on a random strand; per reference position present w.p. 0.94, events per present position
from the measured pmf (tail >= 8 geometric); 4 % extra NNNNNN rows; model_mean = a seeded 4096-entry table
U(55,117) pA on the strand-specific 6-mer, 2 decimals; event_mean = model_mean + N(-0.17, 2.44^2), 2 decimals;
event_index strictly increasing ('+') / decreasing ('-'); mean read quality U(6,12).
"""
import numpy as np

from . import _lib
from .refmark import methylate_references, m_bitmask

GENOME_LEN = 4641652
GENOME_SEED = 20190101
_EV_PMF = np.array([0.527, 0.243, 0.113, 0.058, 0.027, 0.014, 0.007, 0.011])   # 1..7, >=8


def genome(length=GENOME_LEN, seed=GENOME_SEED):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=length, dtype=np.uint8)
    return codes


def codes_to_str(codes):
    return np.frombuffer(b'ACGT', dtype=np.uint8)[codes].tobytes().decode('ascii')


def kmer_indices(codes, k=6):
    """(fwd, revcomp) base-4 index of the k-mer starting at every position (zero-padded past the end)."""
    n = len(codes)
    c = np.concatenate([codes.astype(np.int32), np.zeros(k, dtype=np.int32)])
    fwd = np.zeros(n, dtype=np.int32)
    rc = np.zeros(n, dtype=np.int32)
    for i in range(k):
        fwd = fwd * 4 + c[i:i + n]
        rc = rc + (3 - c[i:i + n]) * (4 ** i)
    return fwd, rc


class SynthRef(object):
    """The marked reference for a synthetic genome (duck-types MarkedReference for Finisher/compute)."""

    def __init__(self, codes, base='A', motif='GATC', name='ecoli_syn'):
        self.names = [name]
        seq = codes_to_str(codes)
        self.records = [(name, seq)]
        self.meth = {0: methylate_references(seq, base, motif=motif, contig=name)}
        self.base, self.motif = base, motif

    def device_arrays(self):
        mf, mr = self.meth[0]
        seq = np.frombuffer(self.records[0][1].encode('ascii'), dtype=np.uint8)
        return dict(contig_len=np.array([len(seq)], dtype=np.int64), seq_off=np.zeros(1, dtype=np.int64),
                    word_off=np.zeros(1, dtype=np.int64),
                    seq=np.ascontiguousarray(np.concatenate([seq, np.zeros(8, np.uint8)])),
                    mbits_fwd=np.ascontiguousarray(m_bitmask(mf)), mbits_rev=np.ascontiguousarray(m_bitmask(mr)))


def make_table(n_rows, seed=1, codes=None, read_len=(5000, 20000), chunk_reads=64):
    """-> (_lib.Table with n_rows rows, qual float64[n_reads]).  Deterministic in (n_rows, seed)."""
    if codes is None:
        codes = genome()
    G = len(codes)
    fwd, rc = kmer_indices(codes)
    rng = np.random.default_rng(seed)
    model_tbl = (np.round(np.random.default_rng(777).uniform(55.0, 117.0, size=4096), 2) * 100).round().astype(np.int64) * 100
    cols = dict(pos=[], ev=[], mu=[], idx=[], flags=[])
    seg_begin, n_reads, total = [0], 0, 0
    while total < n_rows:
        for _ in range(chunk_reads):
            if total >= n_rows:
                break
            L = int(rng.integers(read_len[0], read_len[1] + 1))
            L = min(L, G - 6)
            s = int(rng.integers(0, G - 5 - L))
            rev = bool(rng.integers(0, 2))
            p = np.arange(s, s + L, dtype=np.int32)
            p = p[rng.random(L) < 0.94]
            cls = rng.choice(8, size=len(p), p=_EV_PMF / _EV_PMF.sum())
            nev = cls + 1
            tail = cls == 7
            nev[tail] = 8 + rng.geometric(0.5, size=int(tail.sum())) - 1
            rows_p = np.repeat(p, nev)
            # extra NNNNNN rows (4 %): inserted after a random subset of rows, same position
            extra = rng.random(len(rows_p)) < 0.04
            rep = 1 + extra.astype(np.int64)
            pos = np.repeat(rows_p, rep)
            is_n = np.zeros(len(pos), dtype=bool)
            ends = np.cumsum(rep) - 1
            is_n[ends[extra]] = True
            n = len(pos)
            kidx = rc[pos] if rev else fwd[pos]
            mu = model_tbl[kidx]
            ev = mu + np.round(rng.normal(-0.17, 2.44, size=n) * 100).astype(np.int64) * 100
            ev_n = (np.round(rng.uniform(60.0, 120.0, size=n), 2) * 100).round().astype(np.int64) * 100
            mu = np.where(is_n, 0, mu)
            ev = np.where(is_n, ev_n, ev)
            i0 = int(rng.integers(0, 50000))
            idx = (i0 + n - np.arange(n)) if rev else (i0 + np.arange(n))
            fl = np.zeros(n, dtype=np.uint8)
            eq = (fwd[pos] == rc[pos]) if rev else np.ones(n, dtype=bool)
            fl[eq & ~is_n] |= _lib.F_KMER_EQ
            fl[is_n] |= _lib.F_MODEL_N
            fl[0] |= _lib.F_SEG_START | _lib.F_NAME_START
            if total + n > n_rows:
                keep = n_rows - total
                pos, ev, mu, idx, fl = pos[:keep], ev[:keep], mu[:keep], idx[:keep], fl[:keep]
                n = keep
            cols['pos'].append(pos.astype(np.int32))
            cols['ev'].append(ev.astype(np.int32))
            cols['mu'].append(mu.astype(np.int32))
            cols['idx'].append(idx.astype(np.int32))
            cols['flags'].append(fl)
            total += n
            seg_begin.append(total)
            n_reads += 1
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    qual = np.random.default_rng(seed + 99991).uniform(6.0, 12.0, size=n_reads)
    names = ['%08x-syn-%06d_Basecall_2D_template' % (seed & 0xffffffff, i) for i in range(n_reads)]
    table = _lib.Table(cat['pos'], cat['ev'], cat['mu'], cat['idx'], cat['flags'], np.array(seg_begin, dtype=np.int64),
                       np.arange(n_reads, dtype=np.int32), np.zeros(n_reads, dtype=np.int32), n_reads,
                       read_names=names)
    return table, qual


def tile_table(table, qual, times):
    """Repeat a table `times` times (fresh read ids per copy): a cheap way to reach 10^8 rows."""
    if times == 1:
        return table, qual
    n, nr = table.n_rows, table.n_reads
    seg_begin = np.concatenate([table.seg_row_begin[:-1] + i * n for i in range(times)] + [[n * times]])
    seg_read = np.concatenate([table.seg_read + i * nr for i in range(times)])
    names = None
    if table.read_names is not None:
        names = ['%s.%d' % (nm, i) for i in range(times) for nm in table.read_names]
    t = _lib.Table(np.tile(table.pos, times), np.tile(table.event_e4, times), np.tile(table.model_e4, times),
                   np.tile(table.event_idx, times), np.tile(table.flags, times), seg_begin, seg_read,
                   np.tile(table.seg_contig, times), nr * times, read_names=names)
    return t, np.tile(qual, times)


def write_tsv_native(table, codes, path, contig='ecoli_syn', n_threads=0):
    """The same text as write_tsv, written by the library on all host cores (mc_synth_write_tsv).  -> bytes written."""
    import ctypes as C
    seq = codes_to_str(codes).encode('ascii')
    names = _lib._cstr_array(table.read_names)
    v = table.view()
    nb = C.c_int64(0)
    _lib.check(_lib.lib().mc_synth_write_tsv(path.encode('utf-8'), C.byref(v), seq, len(seq), contig.encode('ascii'), names,
                                             int(n_threads), C.byref(nb)))
    return nb.value


def write_inputs(table, qual, codes, directory, stem='syn', contig='ecoli_syn'):
    """TSV + FASTA + FASTQ of a synthetic workload in `directory` -> dict of paths (the FASTQ gives every read a constant
    phred equal to its rounded quality)."""
    import os
    paths = dict(tsv=os.path.join(directory, stem + '.eventalign.tsv'), fasta=os.path.join(directory, 'ref.fasta'),
                 fastq=os.path.join(directory, 'reads.fastq'))
    write_tsv_native(table, codes, paths['tsv'], contig=contig)
    with open(paths['fasta'], 'w') as fa:
        s = codes_to_str(codes)
        fa.write('>%s\n' % contig + '\n'.join(s[i:i + 60] for i in range(0, len(s), 60)) + '\n')
    with open(paths['fastq'], 'w') as fq:
        for i, name in enumerate(table.read_names):
            fq.write('@%s\nACGTACGTAC\n+\n%s\n' % (name, chr(33 + int(round(qual[i]))) * 10))
    return paths


def write_tsv(table, codes, path, contig='ecoli_syn'):
    """Write a table as nanopolish-eventalign text (pure Python: slow, use on <= 10^6 rows; write_tsv_native is the fast one)."""
    seq = codes_to_str(codes)
    comp = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}
    with open(path, 'w') as out:
        for seg in range(table.n_seg):
            name = table.read_names[int(table.seg_read[seg])]
            for r in range(int(table.seg_row_begin[seg]), int(table.seg_row_begin[seg + 1])):
                p = int(table.pos[r])
                ref_kmer = seq[p:p + 6]
                fl = int(table.flags[r])
                if fl & _lib.F_MODEL_N:
                    mk = 'NNNNNN'
                elif fl & _lib.F_KMER_EQ:
                    mk = ref_kmer
                else:
                    mk = ''.join(comp[c] for c in reversed(ref_kmer))
                out.write('%s\t%d\t%s\t%s\tt\t%d\t%.2f\t1.500\t0.00200\t%s\t%.2f\t1.50\t0.10\n' % (
                    contig, p, ref_kmer, name, int(table.event_idx[r]), table.event_e4[r] / 10000.0, mk,
                    table.model_e4[r] / 10000.0))
