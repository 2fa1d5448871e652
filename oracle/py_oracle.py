"""CPU oracle (pure Python) for the mCaller hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in our own words, the algorithm of the reference's
``extract_contexts.py::extract_features`` (the per-row window machine) and of the
``MLPClassifier.predict_proba`` call it makes per observation.  It exists so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can
check the HIP path.  Nothing under ``mcaller_amd/`` may import it.

Pinning: this restatement is pinned against the reference itself, run in the
build container by ``tests/golden/make_golden.py`` (testdata cases + several
hundred randomly generated micro-cases, byte-for-byte on the ``.diffs`` text and
the printed counters).  The committed fixtures under ``tests/golden/`` are the
captured reference outputs.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Pure-Python loops: use only on small inputs; the C oracle
(``oracle/mc_oracle.c``) is the one used at 10^6..10^8 events.
"""
import math

COMP = {'A': 'T', 'C': 'G', 'T': 'A', 'G': 'C', 'N': 'N', 'M': 'M'}  # extract_contexts.py:11


class RefExit(Exception):
    """The reference reached one of its ``print(...); sys.exit(0)`` paths."""


# ----------------------------------------------------------------------------------------------
# sequence helpers (extract_contexts.py:14-29)
# ----------------------------------------------------------------------------------------------
def complement(seq):
    return ''.join(COMP[c] for c in seq)            # KeyError on other letters, like :15


def revcomp(seq, rev=True):
    return complement(seq)[::-1] if rev else seq     # :18-22


def strand_char(rev):
    return '-' if rev else '+'                       # :25-29


# ----------------------------------------------------------------------------------------------
# reference marking (extract_contexts.py:33-81)
# ----------------------------------------------------------------------------------------------
def mark_motif(seq, motif, base):
    """:33-41 with meth_position=None: every `base` inside `motif` becomes 'M',
    occurrences replaced left-to-right, non-overlapping (str.replace)."""
    return seq.replace(motif, 'M'.join(motif.split(base)))


def mark_positions(seq, positions, base, log):
    """:45-56.  0-based positions; the base there must be `base` or already 'M'."""
    for p in positions:
        if seq[p] == base or seq[p] == 'M':
            seq = seq[:p] + 'M' + seq[p + 1:]
        else:
            log('Base {} does not correspond to methylated base - check reference positions '
                'are 0-based - quitting thread now'.format(p))
            raise RefExit()
    return seq


def read_positions(path, contig, strand):
    """:66-67 -- tokens: chrom pos strand [label]; lines with < 2 tokens ignored."""
    out = []
    for line in open(path, 'r').read().split('\n'):
        t = line.split()
        if len(t) > 1 and t[2] == strand and t[0] == contig:
            out.append(int(t[1]))
    return out


def mark_reference(seq, base, motif, positions_file, contig, log):
    """:60-73 -> (meth_fwd, meth_rev)."""
    if not positions_file and motif:
        return mark_motif(seq, motif, base), mark_motif(seq, revcomp(motif), COMP[base])
    if positions_file:
        fwd = read_positions(positions_file, contig, '+')
        rev = read_positions(positions_file, contig, '-')
        return (mark_positions(seq, fwd, base, log),
                mark_positions(seq, rev, COMP[base], log))
    log('no motifs or positions specified')
    raise RefExit()


def read_fasta(path):
    """What Bio.SeqIO.parse(path,'fasta') gives the reference (:77-80): id = first token of the
    title, sequence = the lines joined (blanks and CR removed)."""
    recs, name, chunks = [], None, []
    with open(path, 'r') as fh:
        for line in fh:
            if line.startswith('>'):
                if name is not None:
                    recs.append((name, ''.join(chunks)))
                title = line[1:].rstrip()
                name = title.split(None, 1)[0] if title.split() else ''
                chunks = []
            elif name is not None:
                chunks.append(line.strip().replace(' ', '').replace('\r', ''))
    if name is not None:
        recs.append((name, ''.join(chunks)))
    return recs


def read_fastq_quality(path):
    """read_qual.py:6-19: {id.split(':')[0].split('_')[0]: mean phred} (4-line records)."""
    import gzip
    opener = (lambda p: gzip.open(p, 'rt')) if path.find('.gz') != -1 else (lambda p: open(p, 'r'))
    out = {}
    with opener(path) as fh:
        while True:
            title = fh.readline()
            if not title:
                break
            if not title.strip():
                continue
            fh.readline()
            fh.readline()
            qual = fh.readline().rstrip('\n').rstrip('\r')
            rid = title[1:].split(None, 1)[0].split(':')[0].split('_')[0]
            phred = [ord(c) - 33 for c in qual]
            out[rid] = sum(phred) / len(phred)       # exact integer sum / n == np.mean of ints
    return out


def submodel_table(base, twobase):
    """base_models, :99-106."""
    if base == 'A' and twobase:
        return {'MG': 'MG', 'MC': 'MH', 'MA': 'MH', 'MT': 'MH', 'MM': 'MH', 'MH': 'MH',
                'AT': 'MH', 'AC': 'MH', 'AG': 'MG', 'AA': 'MH', 'AM': 'MH'}
    t = {}
    for first in ('M', 'A', 'T'):
        for nxt in ('A', 'C', 'G', 'T', 'M'):
            t[first + nxt] = 'general'
    return t


# ----------------------------------------------------------------------------------------------
# numerics (NumPy semantics restated)
# ----------------------------------------------------------------------------------------------
def round_dec(x, decimals):
    """np.round(x, decimals) for a float64 scalar: rint(x*10^d)/10^d (half-to-even)."""
    f = float(10 ** decimals)
    y = x * f
    if math.isinf(y) or math.isnan(y) or abs(y) >= 2.0 ** 52:
        r = y
    else:
        r = float(round(y))
        if r == 0.0 and (y < 0 or (y == 0 and math.copysign(1.0, y) < 0)):
            r = -0.0
    return r / f


def pairwise_sum(a, lo, n):
    """NumPy's DOUBLE pairwise_sum (umath loops_utils): n<8 sequential from -0.0; n<=128 eight
    strided accumulators, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added in order;
    else split at n/2 rounded down to a multiple of 8."""
    if n < 8:
        res = -0.0
        for i in range(n):
            res += a[lo + i]
        return res
    if n <= 128:
        r = [a[lo + j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += a[lo + i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[lo + i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum(a, lo, n2) + pairwise_sum(a, lo + n2, n - n2)


def np_mean(values):
    """np.mean(list of float64): (0.0 + pairwise_sum) / n  (extract_contexts.py:186)."""
    return (0.0 + pairwise_sum(values, 0, len(values))) / len(values)


def fmt_float(x):
    """str(np.float64(x)): shortest round-trip repr, same text as Python's repr(float)."""
    return repr(float(x))


def mlp_forward(weights, x):
    """sklearn MLPClassifier.predict_proba for the shipped 7-100-1 tanh/logistic nets
    (call site extract_contexts.py:199): p1 = expit(tanh(x.W1+b1).W2+b2)."""
    import numpy as np
    W1, b1, W2, b2 = weights
    h = np.tanh(np.asarray(x, dtype=np.float64) @ W1 + b1)
    z = h @ W2 + b2
    return float(1.0 / (1.0 + np.exp(-z[0])))


# ----------------------------------------------------------------------------------------------
# the window machine (extract_contexts.py:110-303)
# ----------------------------------------------------------------------------------------------
class Counters:
    def __init__(self):
        self.observations = 0
        self.positions = set()
        self.multi = set()
        self.with_skips = set()
        self.too_many = set()

    def lines(self):
        return ['thread finished processing...:',
                '%d observations' % self.observations,
                '%d positions' % len(self.positions),
                '%d regions with multiple methylated bases' % len(self.multi),
                '%d observations with skips included' % len(self.with_skips),
                '%d observations with too many skips' % len(self.too_many)]


def consumed_lines(tsv_path, startline, endline):
    """Row ingest, :140-148: seek to max(start-500,0); read batches of lines whose total size
    first exceeds 8,000,000 (io readlines(hint) rule) while linepos <= endline-500."""
    with open(tsv_path, 'r') as fh:
        fh.seek(max(startline - 500, 0))
        linepos = max(startline - 500, 0)
        while linepos <= endline - 500:
            batch = fh.readlines(8000000)
            if not batch:
                # the reference would spin forever here; callers never create this case
                raise RuntimeError('reference would loop forever (file ended before endline-500)')
            for line in batch:
                linepos += len(line)
                yield line


def extract_features_oracle(tsv_path, fasta_path, read2qual, k, skip_thresh, qual_thresh,
                            models, startline, endline, train=False, pos_label=None,
                            base=None, motif=None, positions_list=None):
    """Literal restatement of extract_features (:110-303).

    `models`: None in train mode, else {'general': w} or {'MG': w, 'MH': w, ...} with
    w = (W1, b1, W2, b2), plus key '__twobase__' -> bool (was the pickle a dict? :126-130).
    Returns dict(rows=[list of output rows (list of str)], stdout=[lines], signals, contexts,
    exit=bool, written=[rows that reached the tmp file]).
    """
    out_rows, written, stdout = [], [], []
    log = stdout.append
    cnt = Counters()

    if not train:
        twobase = models['__twobase__']
        table = submodel_table(base, twobase)                                    # :131
        signals = contexts = None
    else:
        table = submodel_table(base, False)                                      # :133
        signals = {v: {} for v in table.values()}
        contexts = {v: {} for v in table.values()}

    # machine state (:113-119)
    last_read = ''
    last_contig = None
    mpos = None
    slots = [[] for _ in range(k)]
    last_rev = last_ref = None
    first_idx = None
    meth_fwd = meth_rev = None
    pending = []                                     # 'towrite' (:137)

    def finish(exited):
        if not exited:
            written.extend(pending)                                              # :293
            stdout.extend(cnt.lines())                                           # :295-301
        return dict(rows=out_rows, written=written, stdout=stdout, signals=signals,
                    contexts=contexts, exit=exited)

    try:
        for line in consumed_lines(tsv_path, startline, endline):
            tok = line.split()[:12]
            if len(tok) < 12:                                                    # :149-152
                continue
            chrom, pos_s, ref_kmer, name, _x, idx_s, ev_s, _sd, _y, model_kmer, mu_s, _msd = tok

            if chrom != last_contig:                                             # :154-160
                found = None
                for cid, seq in read_fasta(fasta_path):
                    if cid == chrom:
                        found = mark_reference(seq.upper(), base, motif, positions_list, chrom, log)
                        break
                if found is None:
                    log('Error: could not find sequence for reference contig ' + chrom)
                    continue
                meth_fwd, meth_rev = found
                last_contig = chrom
            if name != last_read:                                                # :161-162
                first_idx = int(idx_s)
            try:                                                                 # :163-166
                qual = read2qual[name]
            except KeyError:
                qual = read2qual[name.split(':')[0].split('_')[0]]
            if qual < qual_thresh or model_kmer == 'NNNNNN':                     # :167-168
                continue
            if (name != last_read and ref_kmer == model_kmer) or \
               (name == last_read and int(idx_s) > first_idx):                   # :169-174
                rev, meth = False, meth_fwd
            else:
                rev, meth = True, meth_rev
            pos = int(pos_s)
            kmer = meth[pos:pos + k]                                             # :176

            # ---- flush the finished window (:179-239) ----
            if mpos and ((pos >= mpos + 1 and name == last_read) or name != last_read):
                nskip = sum(1 for s in slots if s == [])
                if nskip <= skip_thresh:
                    if nskip > 0:
                        cnt.with_skips.add((last_read, mpos))
                    feats = [np_mean(s) if s != [] else 0 for s in slots]        # :186
                    if not last_rev:
                        feats = feats[::-1]                                      # :187-188
                    try:
                        q = read2qual[last_read]
                    except KeyError:
                        q = read2qual[last_read.split(':')[0].split('_')[0]]
                    feats = feats + [q]                                          # :193
                    context = revcomp(last_ref[mpos - k + 1:mpos + k], last_rev)  # :194
                    feat_txt = ','.join('0' if (isinstance(f, int)) else fmt_float(f) for f in feats)
                    if context[int(len(context) / 2)] == 'M':                    # IndexError propagates
                        try:
                            key = table[context[int(len(context) / 2):int(len(context) / 2) + 2]]
                            if not train:
                                p1 = mlp_forward(models[key], [float(f) for f in feats])
                                if p1 >= 0.5:
                                    label = 'm6A' if base == 'A' else 'm' + base
                                else:
                                    label = base
                                label = label + '\t' + fmt_float(round_dec(p1, 2))   # :207
                            else:
                                label = pos_label[(chrom, mpos, strand_char(last_rev))]  # :210
                                signals[key].setdefault(label, []).append(feats)
                                contexts[key].setdefault(label, []).append(context)
                            row = [chrom, last_read, str(mpos), context, feat_txt,
                                   strand_char(last_rev), label]                 # :216
                            out_rows.append(row)
                            pending.append(row)
                        except (IndexError, KeyError):                           # :218-223
                            log(last_read + '\t' + str(mpos) + '\t' + context + '\t' + feat_txt +
                                '\t' + strand_char(last_rev) + ' - Index or Key Error')
                            raise RefExit()
                    else:                                                        # :224-228
                        log(last_read + '\t' + str(mpos) + '\t' + context + '\t' + feat_txt +
                            '\t' + strand_char(last_rev))
                        raise RefExit()
                    cnt.observations += 1
                    if cnt.observations % 5000 == 0:                             # :230-232
                        written.extend(pending)
                        del pending[:]
                    cnt.positions.add(mpos)
                else:
                    cnt.too_many.add((last_read, mpos))                          # :239

                # ---- reset or shift (:242-266) ----
                if ('M' not in kmer) or name != last_read or pos > mpos + skip_thresh + 1:
                    slots = [[] for _ in range(k)]
                    mpos = None
                else:
                    if kmer[0] != 'M':
                        cnt.multi.add((last_read, mpos))                         # :247-248
                    old = mpos
                    mpos = pos + kmer.index('M')
                    s = min(k, mpos - old)
                    slots = [[] for _ in range(s)] + slots[:-s]                  # :255
                    if len(slots) != k:                                          # :257-266
                        log('n diffs off')
                        raise RefExit()

            # ---- accumulate (:269-291) ----
            if 'M' in kmer:
                off = kmer.index('M')
                if mpos:
                    if name != last_read:
                        mpos = None
                        slots = [[] for _ in range(k)]
                    elif rev != last_rev:
                        mpos = None                                              # slots kept (:276-277)
                if not mpos:
                    mpos = pos + off
                last_read, last_rev, last_ref = name, rev, meth
                slots[off].append(round_dec(float(ev_s) - float(mu_s), 4))       # :286
            elif mpos:
                mpos = None                                                      # :289-291
                slots = [[] for _ in range(k)]
    except RefExit:
        return finish(True)
    return finish(False)
