"""`extract_features` on MI355X: the drop-in for the reference's extract_contexts.py:110-303.

Same signature, same side effects (appends `<tsv minus ext>.diffs.<k>[.train].tmp<startline>`, prints the
five counter lines, returns `(signals, contexts)` in train mode), same error behaviour (messages +
`sys.exit(0)`), but the per-row window machine and the per-observation `predict_proba` run as HIP kernels
behind the C ABI of include/mcaller_hip.h.  The host keeps what is text: FASTA/positions marking
(refmark.py), the model file (model_io.py), context strings and number formatting.

There is no CPU path here: without libmcaller_hip.so or without a GPU every call raises.
"""
import sys

import numpy as np

from . import _lib
from .device import get_device
from .model_io import load_model_file
from .refmark import MarkedReference, revcomp, strand, base_comps, comp  # noqa: F401  (reference names)

_I = _lib


def base_models(base, twobase=False):
    """Sub-model key for a context's two centre characters (extract_contexts.py:99-106)."""
    if base == 'A' and twobase:
        return {'MG': 'MG', 'MC': 'MH', 'MA': 'MH', 'MT': 'MH', 'MM': 'MH', 'MH': 'MH', 'AT': 'MH', 'AC': 'MH',
                'AG': 'MG', 'AA': 'MH', 'AM': 'MH'}
    base_model = {'M' + nextb: 'general' for nextb in ['A', 'C', 'G', 'T', 'M']}
    base_model.update({'A' + nextb: 'general' for nextb in ['A', 'C', 'G', 'T', 'M']})
    base_model.update({'T' + nextb: 'general' for nextb in ['A', 'C', 'G', 'T', 'M']})
    return base_model


def writefi(data, fi):
    """Append rows to the tmp file (extract_contexts.py:83-86)."""
    with open(fi, 'a') as outfi:
        for entry in data:
            outfi.write('\t'.join(entry) + '\n')


def write_text(blob, fi):
    """writefi for rows that are already text."""
    with open(fi, 'ab') as outfi:
        outfi.write(blob)


def fmt_float(x):
    """str(np.float64): shortest round-trip repr (what the reference's str(diff) prints)."""
    return repr(float(x))


def round2(p):
    """np.round(p, 2) (extract_contexts.py:207)."""
    return float(np.round(np.float64(p), 2))


class Prepared(object):
    """Everything the kernels and the formatter need for one (tsv byte range, reference, marking)."""
    pass


def _lookup_quality(read2qual, name):
    try:
        return read2qual[name]                                     # extract_contexts.py:163-166
    except KeyError:
        return read2qual[name.split(':')[0].split('_')[0]]


def prepare(tsv_input, fasta_input, read2qual, startline, endline, base, motif, positions_list, n_threads=0,
            exact_range=False, ref=None, quiet=False):
    """Parse + mark: the host-side pre-pass.  Returns a Prepared; `fatal` holds the exception the
    reference would hit at table row `len(table)` (the table is cut there).  `ref`: a MarkedReference to go on
    with (the shards of one file share it); quiet: the "could not find sequence" lines are kept in P.messages
    instead of being printed."""
    P = Prepared()
    if ref is None:
        ref = MarkedReference(fasta_input, base, motif, positions_list)
    table = _lib.parse_eventalign(tsv_input, startline, endline, ref.names, n_threads, exact_range=exact_range)
    return prepare_table(P, table, ref, read2qual, quiet)


def prepare_table(P, table, ref, read2qual, quiet=False):
    """The part of `prepare` behind the parser: contigs marked as they first appear, read qualities looked up.  `table`: from
    the host parser, or made on the device (Device.parse_end)."""
    P.messages = ['Error: could not find sequence for reference contig ' + name for name in table.unknown]   # :159
    if not quiet:
        for line in P.messages:
            print(line)
    P.fatal = None
    qual_obj = [None] * table.n_reads
    cut_seg = None
    for seg in range(table.n_seg):
        cid, rid = int(table.seg_contig[seg]), int(table.seg_read[seg])
        try:
            if cid not in ref.meth:
                ref.mark(cid)                                                    # :154-157 (may print + exit)
            if qual_obj[rid] is None:
                qual_obj[rid] = _lookup_quality(read2qual, table.read_names[rid])
        except (SystemExit, Exception) as e:                                     # noqa
            P.fatal = e
            cut_seg = seg
            break
    if cut_seg is not None and table.pos is not None:      # (a device-parsed table is only ever streamed: fatal sends the file to the one-table path)
        table = table.slice_segments(0, cut_seg)
    P.ref, P.table, P.qual_obj = ref, table, qual_obj
    P.qual = np.array([float(q) if q is not None else np.nan for q in qual_obj], dtype=np.float64)
    return P


def submodel_setup(modelset, base):
    """-> (base_model table, [weights...], key -> index, uint8[256] context[k] char -> index or 255)."""
    table = base_models(base, modelset.twobase)
    keys = modelset.keys()
    index = {key: i for i, key in enumerate(keys)}
    soc = np.full(256, 255, dtype=np.uint8)
    for c in range(256):
        two = 'M' + chr(c)
        if two in table and table[two] in index:
            soc[c] = index[table[two]]
    return table, [modelset.models[key] for key in keys], index, soc


def compute(P, k, skip_thresh, qual_thresh, modelset, base, train, device=None, tail_contig=-1):
    """Upload + run the HIP path.  Returns (records, info dict)."""
    dev = device if device is not None else get_device()
    dev.set_reference(P.ref.device_arrays())
    dev.upload_table(P.table)
    dev.set_read_quality(P.qual)
    if not train:
        _, weights, _, soc = submodel_setup(modelset, base)
        dev.set_classifier(weights, soc)
    rec = dev.extract(k, skip_thresh, qual_thresh, tail_contig=tail_contig, score=not train)
    return rec


class Finisher(object):
    """Flush records -> the reference's rows, counters and train dicts, in record (= file) order.

    Predict mode: the rows come with the records when the device wrote them (streamed shards: mc_rowtext.hip), else they are
    written by the native formatter (mc_format_diffs, all host cores); a record it hands
    back (context leaving the contig, unscored, unknown sub-model key: the reference's exit/crash paths) goes through
    `_one`, the literal per-record transcription of extract_contexts.py:179-239, which train mode uses throughout
    (it has to build the Python lists the caller trains on)."""

    def __init__(self, P, k, base, train, modelset=None, pos_label=None, device=None, tail_chrom=None):
        self.P, self.k, self.base, self.train = P, k, base, train
        self.pos_label = pos_label
        self.device = device
        self.tail_chrom = tail_chrom
        if not train:
            self.table, _, self.model_index, self.soc = submodel_setup(modelset, base)
            self.model_keys = modelset.keys()
        else:
            self.table = base_models(base, False)                                 # :133
            self.model_keys = None
        self.signals = {bm: {} for bm in self.table.values()} if train else None
        self.contexts = {bm: {} for bm in self.table.values()} if train else None
        self.stdout = None      # where the exit paths' lines go (None: sys.stdout; a stream that will be replayed by the one-table path: a sink)
        self.host_scored = {}   # record -> probability, for the records the host had to score itself (edge contexts)
        self.blobs = []         # emitted rows as text (bytes), in order
        self.num_observations = 0
        self.pos_set, self.multi, self.w_skips, self.skipped = set(), set(), set(), set()
        self._n_pos = self._n_multi = self._n_wskips = self._n_skipped = self._kept_pos = None    # set by the vectorised counters

    # ---- output ----
    def write_to(self, sink):
        """The rows to sink(bytes-like), piece by piece as they were made (no joined copy) -> bytes written."""
        n = 0
        for b in self.blobs:
            if len(b):
                n += len(b)
                sink(b.view if isinstance(b, (_lib.LibBuffer, _lib.RowText)) else b)
            if isinstance(b, _lib.RowText):
                b.release()                                    # (the pinned block goes back to the context)
        return n

    def text(self, max_rows=None):
        """The rows as bytes; max_rows: only the first that many (the reference's 5000-row batches on an exit)."""
        blob = b''.join(b.view if isinstance(b, (_lib.LibBuffer, _lib.RowText)) else b for b in self.blobs)
        if max_rows is None:
            return blob
        return b''.join(blob.splitlines(True)[:max_rows])

    @property
    def rows(self):
        return [line.split('\t') for line in self.text().decode('utf-8', 'surrogateescape').splitlines()]

    def counters(self):
        if self._n_pos is None and self._kept_pos is None and getattr(self, '_counted_natively', False):
            n = self._rec.n
            self._kept_pos = self._site_pos[:n][(self._info[:n] & _I.I_TOO_MANY) == 0]
        if self._n_pos is None and self._kept_pos is not None:
            self._n_pos = len(distinct_positions(self._kept_pos))
        n_pos = len(self.pos_set) if self._n_pos is None else self._n_pos
        n_multi = len(self.multi) if self._n_multi is None else self._n_multi
        n_wskips = len(self.w_skips) if self._n_wskips is None else self._n_wskips
        n_skipped = len(self.skipped) if self._n_skipped is None else self._n_skipped
        return ['thread finished processing...:', '%d observations' % self.num_observations,
                '%d positions' % n_pos, '%d regions with multiple methylated bases' % n_multi,
                '%d observations with skips included' % n_wskips,
                '%d observations with too many skips' % n_skipped]

    def _bind(self, rec):
        n = rec.n
        self._rec = rec
        self._info = rec.info[:n]
        self._site_pos = rec.site_pos[:n]
        self._seg_of = rec.site_seg[:n]
        self._lazy = None

    def _per_record(self):
        """(slot means [calls, k], row of every record in them or None, segment of every record's closing row): only the
        per-record transcription (_one) needs these -- a streamed predict-mode shard whose rows all come from the native
        formatter leaves the packed slot means (mc_calls_view.feats_lo32) as they arrived."""
        if self._lazy is None:
            rec, k, t, n = self._rec, self.k, self.P.table, self._rec.n
            m = rec.n_calls                   # (a compacted view, mc_wait_records: rows of the calls only, see Records.call_row)
            feats = rec.feats[:m * k].reshape(m, k)
            row = rec.call_row[:n] if rec.call_row is not None else None
            close_seg = np.searchsorted(t.seg_row_begin, rec.close_row[:n], side='right') - 1
            self._lazy = (feats, row, close_seg)
        return self._lazy

    def run(self, rec):
        """Returns None, or the exception (SystemExit / error) the reference would raise at that record."""
        self._bind(rec)
        n = rec.n
        if self.train or n == 0:
            for j in range(n):
                stop = self._one(j)
                if stop is not None:
                    return stop
            return None
        P, t = self.P, self.P.table
        text = getattr(rec, 'row_text', None)
        if text is not None:
            # the rows came with the records, written on the device (mc_rowtext.hip: every record was one the native formatter
            # would have printed -- anything else and the pass comes without text): the counters are all that is left to do
            self.blobs.append(text)
            self.num_observations += text.n_rows
            self._count(n)
            return None
        label_meth = 'm6A' if self.base == 'A' else 'm' + self.base                # :200-204
        fmt = _lib.DiffsFormatter(rec, t, P.ref.device_arrays(), P.ref.names, [str(q) for q in P.qual_obj], self.k,
                                  label_meth, self.base, self.soc, tail_chrom=self.tail_chrom)
        first, done_to, stop_exc = 0, n, None
        while first < n:
            blob, n_rows, stop = fmt.rows(first, n_threads=FORMAT_THREADS[0])
            self.blobs.append(blob)
            self.num_observations += n_rows
            if stop >= n:
                break
            stop_exc = self._one(stop)                   # the record the formatter handed back
            if stop_exc is not None:
                done_to = stop
                break
            first = stop + 1
        self._count(done_to)
        return stop_exc

    def host_prob(self, rec):
        """Probability per record (one row per record), the host-scored ones filled in."""
        r = rec.by_record()
        p = np.array(r.prob[:r.n], dtype=np.float64)
        for j, v in self.host_scored.items():
            p[j] = v
        return p

    def _count(self, n):
        """The four sets of :184-185,:234-239,:247-248 over records [0, n), vectorised (set sizes only).  The sets hold (read,
        site) pairs, and records come in file order: a table whose read names do not repeat has them in strictly ascending
        order of (read id, site) -- every pair is then a new one and a set's size is a count, no sort (a one-base motif: 150 000
        records per shard, four sorts of them were most of what a shard's rows cost)."""
        if n == self._rec.n and n > 0:
            # (one pass in the library, without the interpreter lock -- mc_count_records; the pairs ascend unless read names repeat)
            counts, ascending, _, _ = self._rec.count(n, seg_read=self.P.table.seg_read)
            if ascending:
                self._n_skipped, self._n_wskips, self._n_multi = counts
                self._kept_pos = None
                self._n_pos = None
                self._counted_natively = True
                return
        info = self._info[:n]
        rid = self.P.table.seg_read[self._seg_of[:n]].astype(np.int64)
        key = (rid << 32) | (self._site_pos[:n].astype(np.int64) & 0xFFFFFFFF)
        too = (info & _I.I_TOO_MANY) != 0
        kept = ~too
        if n < 2 or bool((key[1:] > key[:-1]).all()):
            size = np.count_nonzero
        else:
            size = lambda mask: len(np.unique(key[mask]))            # noqa: E731
        self._n_skipped = int(size(too))
        self._n_wskips = int(size(kept & ((info & _I.I_EMPTY_MASK) != 0)))
        self._n_multi = int(size((info & _I.I_MULTI) != 0))
        self._kept_pos = self._site_pos[:n][kept]           # (the distinct positions: counted when somebody asks, counters())
        self._n_pos = None

    def _one(self, j):
        P, k, t = self.P, self.k, self.P.table
        rec = self._rec
        feats, rows_of, close_seg = self._per_record()
        names = t.read_names
        half = int((2 * k - 1) / 2)
        inf = int(self._info[j])
        seg = int(self._seg_of[j])
        rid = int(t.seg_read[seg])
        read, mpos = names[rid], int(self._site_pos[j])
        rev = bool(inf & _I.I_REV)
        if inf & _I.I_TOO_MANY:
            self.skipped.add((read, mpos))                                    # :239
        else:
            empty = inf & _I.I_EMPTY_MASK
            if empty:
                self.w_skips.add((read, mpos))                                # :184-185
            row = j if rows_of is None else int(rows_of[j])
            diffs = [0 if (empty >> i) & 1 else float(feats[row, i]) for i in range(k)]
            qual = P.qual_obj[rid]
            diffs_txt = ','.join(['0' if (empty >> i) & 1 else fmt_float(feats[row, i]) for i in range(k)]
                                 + [str(qual)])
            cseg = int(close_seg[j])
            chrom = self.tail_chrom if cseg >= t.n_seg else P.ref.names[int(t.seg_contig[cseg])]
            last_ref = P.ref.meth[int(t.seg_contig[seg])][1 if rev else 0]
            context = revcomp(last_ref[mpos - k + 1:mpos + k], rev)           # :194 (Python slicing rules)
            line = read + '\t' + str(mpos) + '\t' + context + '\t' + diffs_txt + '\t' + strand(rev)
            centre = int(len(context) / 2)
            if context[centre] == 'M':                                        # IndexError propagates, as there
                try:
                    twobase_model = self.table[context[centre:centre + 2]]
                    if not self.train:
                        mi = self.model_index[twobase_model]                  # KeyError: model[...] :199
                        p1 = rec.prob[row]
                        want = (inf >> _I.I_NEXT_SHIFT) & 0xFF
                        if (inf & _I.I_EDGE) or np.isnan(p1):
                            dev = self.device if self.device is not None else get_device()
                            p1 = dev.mlp_forward(np.array([diffs + [float(qual)]], dtype=np.float64),
                                                 np.array([mi], dtype=np.uint8))[0]
                            self.host_scored[j] = float(p1)
                        elif len(context) > half + 1 and ord(context[half + 1]) != want:
                            raise AssertionError('device and host disagree on the sub-model of %s' % line)
                        if p1 >= 0.5:
                            label = 'm6A' if self.base == 'A' else 'm' + self.base
                        else:
                            label = self.base
                        label = label + '\t' + fmt_float(round2(p1))          # :207
                    else:
                        label = self.pos_label[(chrom, mpos, strand(rev))]    # :210
                        self.signals[twobase_model].setdefault(label, []).append(diffs + [qual])
                        self.contexts[twobase_model].setdefault(label, []).append(context)
                    row = [chrom, read, str(mpos), context, diffs_txt, strand(rev), label]
                    self.blobs.append(('\t'.join(row) + '\n').encode('utf-8', 'surrogateescape'))
                except (IndexError, KeyError) as e:                           # :218-223
                    print(line, '- Index or Key Error', file=self.stdout)
                    print(list(self.model_keys or []), list(self.table.keys()), context[centre:centre + 2], file=self.stdout)
                    print(e, file=self.stdout)
                    return SystemExit(0)
            else:                                                             # :224-228
                print(line, file=self.stdout)
                return SystemExit(0)
            self.num_observations += 1
            self.pos_set.add(mpos)
        if inf & _I.I_MULTI:
            self.multi.add((read, mpos))                                      # :247-248
        return None


def distinct_positions(pos):
    """np.unique of an array of site positions (small non-negative integers: the contig's length bounds them) without
    sorting it: a mark per position."""
    pos = np.asarray(pos)
    if len(pos) == 0:
        return np.zeros(0, dtype=np.int32)
    lo, hi = int(pos.min()), int(pos.max())
    if lo < 0 or hi - lo > (1 << 28):
        return np.unique(pos)
    seen = np.zeros(hi - lo + 1, dtype=bool)
    seen[pos - lo] = True
    return (np.flatnonzero(seen) + lo).astype(pos.dtype, copy=False)


def cut_names(table, rec):
    """What a cut of the file at a read start can change is `last_read` (extract_contexts.py:161-174): a read whose name equals
    the name of the LAST READ THAT HAD A SITE ROW is tested on `event_idx > first_read_ind` instead of on its k-mers.  A table
    knows that for its own reads (name blocks that repeat: the literal path); across a cut it can only matter for the reads of
    the piece behind the cut up to and including its first read with a site row, against the reads of the piece in front of it
    from its last read with a site row on.  A read with a flush record has a site row, so the reads up to the first one with
    a record (`head`) and the reads from the last one with a record on (`tail`) cover both -- a name that comes back anywhere
    else, gigabytes later, changes nothing.  -> (head names, tail names, the table has records)"""
    names, seg_read = table.read_names, table.seg_read
    n = int(rec.n)
    if n == 0:
        every = set(names[int(r)] for r in seg_read)
        return every, every, False
    first_seg, last_seg = int(rec.site_seg[0]), int(rec.site_seg[n - 1])
    return (set(names[int(r)] for r in seg_read[:first_seg + 1]), set(names[int(r)] for r in seg_read[last_seg:]), True)


FORMAT_THREADS = [int(__import__('os').environ.get('MCALLER_FORMAT_THREADS', '0'))]     # threads of the native row formatter (0: every core this process may use)
STREAM_SHARD_BYTES = 128 << 20      # eventalign text per shard of a streamed file (~10^6 rows)
STREAM_SHARD_MIN_BYTES = 8 << 20    # ... of a short range, at least (a shard costs the main thread half a millisecond whatever its size)
STREAM_MIN_SHARDS = 24              # ... which is cut into at least this many (shard_schedule)
STREAM_SHARD_MAX_BYTES = 2 << 30    # a shard beyond this (the cuts are at read starts: one giant read) sends the file to the one-table path


def shard_schedule(lo, hi):
    """Where a streamed byte range is cut into shards (offsets, to be moved to read starts): shards of STREAM_SHARD_BYTES, but a
    short range -- the piece of one GPU of a sharded run -- in at least STREAM_MIN_SHARDS of them (ten shards fill and drain a
    six-deep pipeline for a third of their time), and the first three shards an eighth, a quarter, half of that: nothing happens on
    the GPU before the first shard's text has been read and sent, and the last two half and a quarter: what is left to do when
    the last text has arrived is one shard's parse, pass, copy-out and rows."""
    total = hi - lo
    full = int(min(STREAM_SHARD_BYTES, max(STREAM_SHARD_MIN_BYTES, total // STREAM_MIN_SHARDS)))
    if total < 2 * full:
        return [lo + total // 2] if total >= 2 * STREAM_SHARD_MIN_BYTES else []
    head, tail = [full // 8, full // 4, full // 2], [full // 2, full // 4]
    if total < 4 * full:
        head, tail = [], []
    body = total - sum(head) - sum(tail)
    n_body = max(1, int(round(body / float(full))))
    sizes = head + [body // n_body] * n_body + tail
    offs, at = [], lo
    for sz in sizes[:-1]:
        at += sz
        offs.append(at)
    return offs


def head_contig(P, qual_thresh):
    """Contig id of the first row of P.table that passes the filters (:167-168), or None: the row that closes the last
    window of the table before it (R6) and supplies that record's chrom column (R8)."""
    t = P.table
    for seg in range(t.n_seg):
        if P.qual[t.seg_read[seg]] < qual_thresh:
            continue
        r0, r1 = int(t.seg_row_begin[seg]), int(t.seg_row_begin[seg + 1])
        if ((t.flags[r0:r1] & _lib.F_MODEL_N) == 0).any():
            return int(t.seg_contig[seg])
    return None


class _Unstreamable(Exception):
    """The file needs the one-table path (an exit path of the reference, a read name in two shards, ...)."""


class StreamResult(object):
    """What stream_features hands back (the rows themselves went to the sink, shard after shard)."""

    def __init__(self):
        self.counters, self.messages, self.names = [], [], set()
        self.n_rows = self.n_bytes = self.n_obs = self.n_multi = self.n_wskips = self.n_skipped = 0
        self.positions = np.zeros(0, dtype=np.int32)
        self.signals = self.contexts = None
        # what crosses a cut of the file in front of / behind this stream (see cut_names): names up to the first read with a
        # flush record, names from the last such read on, whether any read had one
        self.head_names, self.tail_names, self.had_records = set(), set(), False


def stream_features(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelset, endline, base, motif,
                    positions_list, n_shards=None, device=None, sink=None, byte_range=None, tail_of_last=None, on_head=None,
                    on_shard=None, mark_all=False, min_shards=2, train=False, pos_label=None):
    """A whole file (or the byte range of one GPU of a sharded run), as the reference's batch loop (:140-148) streams it -- here
    in shards cut at read starts (a window never spans two reads, :179,:242): two threads read the shards' text into pinned
    memory, the main thread keeps the text of up to six shards on its way to the GPU, where it is parsed (mc_ctx_parse_begin /
    _end / _finish; a shard the device parser declines, or every shard with MCALLER_HOST_PARSER, goes through the host parser
    and mc_ctx_upload_table_async), two passes in flight (mc_extract_features_async), and formats the rows of the shards that
    come back; reading, H2D, parsing, kernels, D2H and formatting overlap.  The rows of a shard go to `sink(bytes)` as soon as
    they exist, in file order (the reference appends every 5000 observations, :230-232): memory is bounded by the shards in
    flight, whatever the file's size.

    byte_range: (lo, hi), both at first lines of reads, instead of what the reference's loop consumes of (0, endline);
    tail_of_last(): called when the last shard is about to be enqueued -> name of the contig of the first unfiltered row
    BEHIND the range (it closes the range's last window, R6/R8), None: end of file; on_head(name | None): called once, as soon as
    the contig of the range's own first unfiltered row is known (what closes the range in front of it);
    on_shard(P, rec, fin, tail name, rows of the shards before): every shard's records when they have been handed out (the
    per-site reduction of a --bed run); mark_all: every contig is marked before the first pass (one site numbering for all the
    GPUs of a run); train: features only, the reference's train dicts are collected (pos_label) and returned.
    -> StreamResult, or raises _Unstreamable (an exit path of the reference, a read name on both sides of a cut -- cut_names():
    whoever called decides what becomes of the rows the sink has seen).  sink() is handed bytes-like objects that are valid
    during the call only."""
    import os
    import time
    t_enter = time.perf_counter()
    dev = device if device is not None else get_device()
    if byte_range is None:
        lo, hi = _lib.eventalign_consumed_range(tsv_input, 0, endline)
    else:
        lo, hi = byte_range
    if n_shards is None:
        n_shards = int(os.environ.get('MCALLER_STREAM_SHARDS', '0')) or None
    if n_shards is None:
        want = shard_schedule(lo, hi)
        if len(want) + 1 < min_shards:
            raise _Unstreamable('one shard')
        cuts = _lib.eventalign_read_cuts_at(tsv_input, want, lo, hi)
    else:
        if n_shards < min_shards:
            raise _Unstreamable('one shard')
        cuts = _lib.eventalign_read_cuts(tsv_input, n_shards, lo, hi)
    pieces = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
    if byte_range is not None and not pieces:
        # the range of a GPU of a sharded run that holds no read (fewer reads than GPUs): a finished piece, not a reason to
        # send the whole file to one GPU
        if on_head is not None:
            on_head(None)
        stream_features.last_clock = dict(shards=0)
        return StreamResult()
    if len(pieces) < min_shards:
        raise _Unstreamable('one shard')
    import contextlib
    import io
    import threading
    ref = MarkedReference(fasta_input, base, motif, positions_list)
    ref.quiet = True                   # (an exit path sends the file to the one-table path, which prints)
    if not train:
        _, weights, _, soc = submodel_setup(modelset, base)
        dev.set_classifier(weights, soc)               # (MLP or forest: either runs behind the emit of a pipelined pass)
    out = StreamResult()
    if train:
        bm = base_models(base, False)                                             # :133
        out.signals = {key: {} for key in bm.values()}
        out.contexts = {key: {} for key in bm.values()}

    L = _lib.lib()
    L.mc_host_pool_config(1, -1)                           # the parser's tables live in pinned memory, recycled
    masks_on_device = False            # (decided below, once the first shards are being read)
    clock = dict(wait_parser=0.0, hand_out=0.0, enqueue=0.0, parse=0.0, shards=len(pieces),     # MCALLER_TIMING
                 wait_records=0.0, format=0.0, write=0.0, out_bytes=0, records=0)        # (hand_out, split: GPU + copy-out waited for | rows formatted | sink)
    # two reader / parser threads take the shards in turn (the native calls spread a shard over all cores, but opening,
    # cutting and stitching are serial: two shards in the works hide that); at most three shards ahead of the GPU
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=2)

    # The text is parsed on the GPU (mc_ctx_parse_*: the host threads only move the bytes into pinned memory -- on a box whose
    # CPU time is rationed the parse is what a file costs) unless MCALLER_HOST_PARSER is set; a shard the device parser
    # declines (a number form that needs strtod, ...) goes through the host parser.
    on_device = not os.environ.get('MCALLER_HOST_PARSER')
    if on_device:
        biggest = max(b - a for a, b in pieces)
        if biggest > STREAM_SHARD_MAX_BYTES:                   # (one giant read: twelve slots of that size are not worth reserving)
            L.mc_host_pool_config(0, -1)
            pool.shutdown(wait=False)
            raise _Unstreamable('a shard of %d bytes' % biggest)
        rows_cap = biggest // 48 + 65536
        try:
            dev.reserve_tables(rows_cap, rows_cap // 16, rows_cap // 16)
        except _lib.McError as e:
            L.mc_host_pool_config(0, -1)
            pool.shutdown(wait=False)
            raise _Unstreamable('the table slots cannot be reserved: %s' % e)
    clock['device_parsed'] = 0
    clock['events'] = []            # (MCALLER_TIMING=2: when the main thread did what)
    t_zero = time.perf_counter()

    def mark(what):
        clock['events'].append((time.perf_counter() - t_zero, what))

    def parse_shard(lo_i, hi_i):
        t_p = time.perf_counter()
        if on_device and hi_i - lo_i < (1 << 32) - 64:         # (mc_ctx_parse_begin: at most 4 GB of text per shard)
            res = _lib.TextBlock(tsv_input, lo_i, hi_i)
        else:
            res = prepare(tsv_input, None, read2qual, lo_i, hi_i, base, motif, positions_list, exact_range=True, ref=ref, quiet=True)
        clock['parse'] += time.perf_counter() - t_p
        return res

    def mark_ahead():
        """The first contig of the file is marked while the first shards are read and sent (marking E. coli takes 14 ms; the main
        thread would do it when the first table comes back, with the GPU waiting).  An exit path of the marking is left to
        the main thread: it marks again and meets it there."""
        try:
            with open(tsv_input, 'rb') as fh:
                fh.seek(pieces[0][0])
                for line in fh.read(1 << 16).splitlines():
                    tok = line.split()
                    if len(tok) >= 12 and tok[0].decode('utf-8', 'surrogateescape') in ref.names:
                        cid = ref.names.index(tok[0].decode('utf-8', 'surrogateescape'))
                        ref.mark(cid)                          # (ref.quiet: nothing is printed from here)
                        ref.device_arrays()                    # (cached: the main thread's set_reference finds them made)
                        return
        except BaseException:                                  # noqa
            pass

    # (started when three shards of text are on their way: what is left of the marking under the interpreter lock -- making
    # Python strings of 2 x 4.6 MB -- would hold up the reader threads at the very start)
    mark_thread = []
    ahead = []                      # futures of the shards being read / parsed by the host threads, in file order
    parsing = []                    # (slot, text, piece) of the shards the device is parsing, in file order
    next_piece = [0]

    def next_shard():
        """The next shard in file order (None behind the last); keeps the parser threads (and the device parser) busy."""
        while next_piece[0] < len(pieces) and len(ahead) < 3:
            ahead.append((pool.submit(parse_shard, *pieces[next_piece[0]]), pieces[next_piece[0]]))
            next_piece[0] += 1
        if not on_device:
            return ahead.pop(0)[0].result() if ahead else None
        held_back = top_up()
        if held_back is not None:
            return held_back
        if not parsing:
            return None
        slot, text, piece = parsing.pop(0)
        mark('parse_end ...')
        try:
            table = dev.parse_end(slot, text)
        except _lib.McError as e:
            raise _Unstreamable('the device parser failed: %s' % e)
        mark('parse_end done')
        if table is None:                                      # declined: the host parser takes the shard
            return prepare(tsv_input, None, read2qual, piece[0], piece[1], base, motif, positions_list, exact_range=True, ref=ref,
                           quiet=True)
        clock['device_parsed'] += 1
        top_up_quietly()                                       # (the marking of the first contig may be waited for next)
        P_new = prepare_table(Prepared(), table, ref, read2qual, quiet=True)
        mark('prepared')
        return P_new

    def top_up():
        """Text of the shards ahead on its way to the device (back to back over the link; 12 table slots), up to six shards --
        four until the reference masks are there: they travel over the same link and the first pass waits for them.  Called
        wherever the main thread is about to wait.  -> a shard the host parser had to take (too long), when it is its turn."""
        while ahead and len(parsing) < (6 if marked[0] >= 0 else 4):
            fut, piece = ahead.pop(0)
            text = fut.result()
            if not isinstance(text, _lib.TextBlock):           # a shard too long for the device parser: parsed by the host already
                if parsing:                                    # (file order: the shards in front of it come first)
                    ahead.insert(0, (fut, piece))
                    break
                return text
            mark('text ready')
            parsing.append((needs_a_slot(dev.parse_begin, text, ref.names, rows_cap), text, piece))
            mark('parse_begin done')
            if not mark_thread and not mark_all and (len(parsing) >= 3 or next_piece[0] >= len(pieces)):
                import threading
                mark_thread.append(threading.Thread(target=mark_ahead, daemon=True))
                mark_thread[0].start()
            while next_piece[0] < len(pieces) and len(ahead) < 3:
                ahead.append((pool.submit(parse_shard, *pieces[next_piece[0]]), pieces[next_piece[0]]))
                next_piece[0] += 1
        return None

    def needs_a_slot(fn, *a):
        """A call that takes a table slot (mc_ctx_parse_begin, mc_ctx_upload_table_async): with every slot taken the oldest pass is
        handed out first; whatever else the streaming machinery declines sends the file to the one-table path."""
        while True:
            try:
                return fn(*a)
            except _lib.McError as e:
                if e.code == _lib.E_NO_FREE_SLOT and in_flight:
                    hand_out()
                    continue
                raise _Unstreamable('the streaming machinery declined: %s' % e)

    positions = [np.zeros(1 << 16, dtype=bool)]        # positions[0][p]: a call at site position p has been seen
    in_flight = []                  # (P, tail name, rows of the shards before it) of the passes enqueued, oldest first
    marked = [-1]                                          # (>= 0: the reference masks are on the device)

    # The rows of a shard are formatted by helper threads and appended by another while the main thread goes on to the next shard (its
    # table, its passes, the wait for its records): a one-base motif writes 1.3 GB of rows per 10^8 events, and formatter + write were
    # two thirds of what the main thread did.  TWO formatting helpers take the shards in turn: what a shard costs there is the native
    # formatter on all host cores (2.5 ms per 10^6 rows of a one-base motif, one call at a time) and 1.5 ms of interpreter around it
    # (the counters, the names at the cuts, the marks) -- the one's interpreter part runs beside the other's native part.  What depends
    # on the order of the shards (names across the cuts, the rows handed to the writer, the totals) is done by every shard in its turn.
    # Not in train mode (the per-record transcription holds the interpreter lock) and not when every shard's records are reduced on
    # the device (on_shard needs what the formatter found).  The helpers are at most two shards behind: the records they read stay
    # where they are until six more passes have been enqueued.
    overlap = not train and on_shard is None and not os.environ.get('MCALLER_NO_OVERLAP')
    # ... and the rows themselves are written on the GPU, behind the records they are made from (mc_rowtext.hip), when the shard's
    # table is one the device parser made: what is left for the helpers is the counters and the write.  MCALLER_DEVICE_ROWS=0: the
    # host formatter throughout.
    device_rows = on_device and not train and os.environ.get('MCALLER_DEVICE_ROWS', '1') != '0'
    clock['device_rows'] = 0
    if device_rows:
        dev.row_text(True, 'm6A' if base == 'A' else 'm' + base, base, first=True)     # (:200-204; the blocks of a stream that failed are free again)
    fmt_pool = ThreadPoolExecutor(max_workers=2) if overlap else None
    write_pool = ThreadPoolExecutor(max_workers=1) if overlap else None     # (... and one more appends them: in order, one shard behind)
    pending, writes = [], []        # the helpers' jobs in flight (futures), if any
    turn = [None]                   # the event the shard handed out last sets when its part in order is done
    failed = [False]                # a shard met an exit path or a name on both sides of a cut: the shards behind it write nothing
    clock['overlapped'] = bool(overlap)
    clock['format_threads'] = 2 if overlap else 0           # (0: the main thread formats)

    def finish_pending(leave=0):
        while len(pending) > leave:
            pending.pop(0).result()                            # (its exception, if it met an exit path, is raised here)
        if not leave:
            while writes:
                writes.pop(0).result()

    def hand_out():
        t_h = time.perf_counter()
        try:
            _hand_out()
        finally:
            clock['hand_out'] += time.perf_counter() - t_h

    def _hand_out():
        P, tail, rows_before = in_flight.pop(0)
        mark('wait ...')
        t_w = time.perf_counter()
        rec = dev.wait()
        t_f = time.perf_counter()
        clock['wait_records'] += t_f - t_w
        mark('records here')
        if overlap:
            finish_pending(leave=1)                            # (the shard before the last: done, or its exit path raised)
            clock['wait_formatter'] = clock.get('wait_formatter', 0.0) + time.perf_counter() - t_f
            before, mine = turn[0], threading.Event()
            turn[0] = mine
            pending.append(fmt_pool.submit(_finish_in_turn, P, tail, rows_before, rec, before, mine))
        else:
            _finish(P, tail, rows_before, rec, None)

    def _finish_in_turn(P, tail, rows_before, rec, before, mine):
        try:
            _finish(P, tail, rows_before, rec, before)
        except BaseException:
            failed[0] = True
            raise
        finally:
            mine.set()

    def _finish(P, tail, rows_before, rec, before):
        t_f = time.perf_counter()
        fin = Finisher(P, k, base, train, modelset=modelset, pos_label=pos_label, device=dev, tail_chrom=tail)
        fin.stdout = io.StringIO()                             # (its exit paths print; the one-table path will)
        stop = fin.run(rec)
        t_r = time.perf_counter()
        if before is not None:
            before.wait()                                      # ---- from here on: in the order of the shards ----
            if failed[0]:
                return
        t_f += time.perf_counter() - t_r                       # (the wait for the turn is not formatting)
        if stop is not None:
            raise _Unstreamable('an exit path of the reference')
        # `last_read` across the cut in front of this shard (cut_names): a name on both sides of it sends the file to the one-table path
        head_n, tail_n, has_rec = cut_names(P.table, rec)
        if head_n & out.tail_names:
            raise _Unstreamable('a read name on both sides of a cut between two shards')
        if not out.had_records:
            out.head_names |= head_n
        out.tail_names = tail_n if has_rec else (out.tail_names | tail_n)
        out.had_records = out.had_records or has_rec
        t_s = time.perf_counter()
        clock['format'] += t_s - t_f

        def write_rows():
            t_w = time.perf_counter()
            n_out = fin.write_to(sink)
            clock['write'] += time.perf_counter() - t_w
            clock['out_bytes'] += n_out
            out.n_bytes += n_out
        if overlap:
            while len(writes) > 1:                             # (at most two shards' rows wait to be written)
                writes.pop(0).result()
            writes.append(write_pool.submit(write_rows))
        else:
            write_rows()
        clock['records'] += int(rec.n)
        clock['device_rows'] += 1 if getattr(rec, 'row_text', None) is not None else 0
        n = rec.n
        if n:
            # the distinct positions of the file: a mark per position, counted at the end (mc_count_records: one pass in the library)
            _, _, lo_pos, top = rec.count(n, pos_marks=positions[0])
            if lo_pos < 0:
                raise _Unstreamable('a negative site position')
            if top > len(positions[0]):
                positions[0] = np.concatenate([positions[0], np.zeros(max(top, 2 * len(positions[0])) - len(positions[0]), dtype=bool)])
                rec.count(n, pos_marks=positions[0])
        if train:                                              # (train mode goes record by record: its sets count)
            out.n_obs += fin.num_observations
            out.n_multi += len(fin.multi)
            out.n_wskips += len(fin.w_skips)
            out.n_skipped += len(fin.skipped)
            for key, by_label in fin.signals.items():
                for label, rows in by_label.items():
                    out.signals[key].setdefault(label, []).extend(rows)
            for key, by_label in fin.contexts.items():
                for label, rows in by_label.items():
                    out.contexts[key].setdefault(label, []).extend(rows)
        else:
            out.n_obs += fin.num_observations
            out.n_multi += len(fin.multi) if fin._n_multi is None else fin._n_multi
            out.n_wskips += len(fin.w_skips) if fin._n_wskips is None else fin._n_wskips
            out.n_skipped += len(fin.skipped) if fin._n_skipped is None else fin._n_skipped
        if on_shard is not None:
            on_shard(P, rec, fin, tail, rows_before)

    def give_back(P_dropped):
        """A table the device parser has put into a slot and that no pass will scan: the slot is free again."""
        slot = getattr(P_dropped.table, 'device_slot', None)
        if slot is not None:
            dev.parse_abandon(slot)
            P_dropped.table.device_slot = None

    def top_up_quietly():
        held = top_up()
        if held is not None:                                   # (a host-parsed shard whose turn has come: back in line)
            from concurrent.futures import Future
            fut_done = Future()
            fut_done.set_result(held)
            ahead.insert(0, (fut_done, (0, 0)))

    def enqueue(P, tail_id):
        n_marked = len(ref.meth)                               # (the parser thread marks contigs as they first appear)
        if not masks_on_device and n_marked != marked[0]:      # a contig marked since the last upload: new masks
            while in_flight:
                hand_out()
            if on_device:
                top_up_quietly()                               # (the link stays busy while the masks are made ready)
            dev.set_reference(ref.device_arrays())
            marked[0] = n_marked
        needs_a_slot(dev.upload_table_async, P.table, P.qual)
        if device_rows:
            # (str(quality) on the device is repr of the double: for what read_qual / the FASTQ reader return, floats)
            dev.row_text(all(isinstance(q, float) for q in P.qual_obj), 'm6A' if base == 'A' else 'm' + base, base)     # :200-204
        dev.run_async(k, skip_thresh, qual_thresh, tail_contig=tail_id, score=not train)

    while next_piece[0] < len(pieces) and len(ahead) < 3:      # (the first shards are read while the masks are made)
        ahead.append((pool.submit(parse_shard, *pieces[next_piece[0]]), pieces[next_piece[0]]))
        next_piece[0] += 1
    # Motif mode: the site masks of every contig are made on the GPU, from the raw bases, before the first text is on its way
    # (mc_ctx_set_reference_motif) -- the marked strings the rows' contexts are sliced from are made by a thread meanwhile and
    # are not waited for by the passes.  (Not for motifs that can overlap themselves, positions mode, very long references:
    # then the masks come from the host's marking, contig by contig as they appear.)
    dev_motif = ref.motif_for_the_device() if not os.environ.get('MCALLER_HOST_PARSER') else None
    if dev_motif is not None and sum(len(seq) for _, seq in ref.records) <= (256 << 20):
        dev.set_reference_motif(ref.raw_arrays(), *dev_motif)
        masks_on_device = True
    marked[0] = 0 if masks_on_device else -1
    P = prev = None
    try:
        if mark_all:                                           # (one site numbering for every GPU of the run: all contigs, now)
            for cid in range(len(ref.names)):
                ref.mark(cid)
        head_told = False
        rows_seen, prev_rows_before = 0, 0
        clock['setup'] = time.perf_counter() - t_enter        # (cuts, FASTA, classifier, table slots, masks: before the first shard is asked for)
        t_loop = time.perf_counter()
        while True:
            t_q = time.perf_counter()
            P = next_shard()
            clock['wait_parser'] += time.perf_counter() - t_q
            if P is not None:
                if P.fatal is not None:
                    raise _Unstreamable('an exit path of the reference')
                out.names.update(P.table.read_names)                       # (what crosses a cut is looked at when the shard's records are here)
                out.messages.extend(P.messages)
                rows_before = rows_seen
                rows_seen += P.table.n_rows
                if P.table.n_rows == 0:
                    give_back(P)
                    continue
                head = head_contig(P, qual_thresh)
                if head is None:
                    give_back(P)
                    continue                                   # no row passes the filters (:167-168): the loop never sees this shard
                if not head_told and on_head is not None:
                    on_head(ref.names[head])
                head_told = True
            if prev is not None:
                if P is not None:
                    tail_id = head
                else:                                          # the range's last shard: what follows the range closes its last window
                    tail_name = tail_of_last() if tail_of_last is not None else None
                    tail_id = ref.names.index(tail_name) if tail_name is not None else -1
                while len(in_flight) >= 2:
                    hand_out()
                t_e = time.perf_counter()
                mark('enqueue ...')
                enqueue(prev, tail_id)
                mark('enqueued')
                clock['enqueue'] += time.perf_counter() - t_e
                in_flight.append((prev, ref.names[tail_id] if tail_id >= 0 else None, prev_rows_before))
                if device_rows and len(in_flight) >= 2:
                    dev.wait_begin()                           # (the oldest pass's copy-out and row writer start now, not when it is waited for)
            prev = P
            if P is None:
                break
            prev_rows_before = rows_before
        if not head_told and on_head is not None:
            on_head(None)
        while in_flight:
            hand_out()
        finish_pending()
    except BaseException:
        next_piece[0] = len(pieces)
        for f, _ in ahead:
            f.cancel()
        try:
            for fut in pending + writes:                       # (the helpers must be done with the records before anything is torn down)
                try:
                    fut.result()
                except BaseException:                          # noqa
                    pass
            del pending[:], writes[:]
            dev.sync()
            for slot, _, _ in parsing:                         # tables the device parser was filling: their slots go back
                dev.parse_abandon(slot)
            for P_left in (P, prev):
                if P_left is not None and getattr(P_left.table, 'device_slot', None) is not None:
                    dev.parse_abandon(P_left.table.device_slot)
                    P_left.table.device_slot = None
            while in_flight:                                   # nothing may stay in flight on the shared device
                in_flight.pop(0)
                dev.wait()
        except Exception:                                      # noqa
            pass
        raise
    finally:
        if device_rows:
            try:
                dev.row_text(False)
            except Exception:                                  # noqa
                pass
        pool.shutdown(wait=True)
        if fmt_pool is not None:
            fmt_pool.shutdown(wait=True)
            write_pool.shutdown(wait=True)
        L.mc_host_pool_config(0, -1)
    clock['loop'] = time.perf_counter() - t_loop
    out.n_rows = rows_seen
    out.positions = np.flatnonzero(positions[0]).astype(np.int32)
    out.counters = ['thread finished processing...:', '%d observations' % out.n_obs, '%d positions' % len(out.positions),
                    '%d regions with multiple methylated bases' % out.n_multi,
                    '%d observations with skips included' % out.n_wskips,
                    '%d observations with too many skips' % out.n_skipped]
    out.ref = ref
    stream_features.last_clock = clock
    return out


def extract_features(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile, classifier,
                     startline, endline=None, train=False, pos_label=None, base=None, motif=None,
                     positions_list=None):
    """Drop-in for extract_contexts.py:110 (see module docstring)."""
    import os
    import time
    timing = os.environ.get('MCALLER_TIMING')
    t_start = time.perf_counter()
    suffix = '.diffs.' + str(k) + ('.train' if train else '') + '.tmp' + str(startline)
    tsv_output = '.'.join(tsv_input.split('.')[:-1]) + suffix                     # :122 / :134
    modelset = None
    if not train:
        modelset = load_model_file(modelfile)                                     # :123-130

    if startline == 0 and endline is not None and os.environ.get('MCALLER_NO_STREAM') is None:
        # a whole file: streamed through the GPU in shards, every shard's rows appended as they come back (:230-232); whatever
        # the shards cannot reproduce (the reference's exit paths, a read name that comes back later) takes the one-table path
        # below from scratch -- the rows appended so far are taken back first, nothing is written or printed twice
        size_before = os.path.getsize(tsv_output) if os.path.exists(tsv_output) else None
        try:
            with open(tsv_output, 'ab') as out_fh:
                res = stream_features(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelset, endline, base,
                                      motif, positions_list, sink=out_fh.write, train=train, pos_label=pos_label)
        except BaseException as e:
            # the rows appended so far are taken back whatever stopped the stream (a device error, MemoryError, ^C: a re-run
            # must not find half a file to append to); only _Unstreamable goes on to the one-table path
            try:                           # (the cleanup must not replace what stopped the stream: the file may never have been opened)
                if size_before is None:
                    os.remove(tsv_output)
                else:
                    os.truncate(tsv_output, size_before)
            except OSError:
                pass
            if not isinstance(e, _Unstreamable):
                raise
        else:
            for line in res.messages:
                print(line)
            counters = res.counters
            if timing:
                ck = getattr(stream_features, 'last_clock', {})
                print('[mcaller_amd timing] streamed in %s shards (%s parsed on the device): total %.3f s | reader / parser threads '
                      '%.3f s | main thread: waiting for the next table %.3f, upload + enqueue %.3f, wait + format %.3f (records '
                      'waited for %.3f, %d records formatted %.3f, %d bytes written %.3f)' % (
                          ck.get('shards'), ck.get('device_parsed'), time.perf_counter() - t_start, ck.get('parse', 0),
                          ck.get('wait_parser', 0), ck.get('enqueue', 0), ck.get('hand_out', 0), ck.get('wait_records', 0),
                          ck.get('records', 0), ck.get('format', 0), ck.get('out_bytes', 0), ck.get('write', 0)), file=sys.stderr)
            if timing == '2':
                for t_ev, what in getattr(stream_features, 'last_clock', {}).get('events', []):
                    print('[mcaller_amd timing] %8.2f ms %s' % (t_ev * 1e3, what), file=sys.stderr)
            for line in counters:                                                 # :295-301
                print(line)
            return (res.signals, res.contexts) if train else None

    P = prepare(tsv_input, fasta_input, read2qual, startline, endline, base, motif, positions_list)
    t_prep = time.perf_counter()
    rec = compute(P, k, skip_thresh, qual_thresh, modelset, base, train)
    t_gpu = time.perf_counter()
    fin = Finisher(P, k, base, train, modelset=modelset, pos_label=pos_label)
    stop = fin.run(rec)
    if stop is None and P.fatal is not None:
        stop = P.fatal
    if stop is not None:
        # the reference dies mid-file: only the 5000-row batches already flushed are on disk (:230-232)
        n_written = (fin.num_observations // 5000) * 5000
        write_text(fin.text(n_written), tsv_output)
        raise stop
    write_text(fin.text(), tsv_output)                                            # :293
    if timing:
        t_end = time.perf_counter()
        print('[mcaller_amd timing] rows=%d records=%d  parse+mark %.3f s | upload+kernels+fetch %.3f s (kernels %s ms) | '
              'format+write %.3f s | total %.3f s' % (P.table.n_rows, rec.n, t_prep - t_start, t_gpu - t_prep,
                                                     get_device().times_ms(), t_end - t_gpu, t_end - t_start),
              file=sys.stderr)

    for line in fin.counters():                                                   # :295-301
        print(line)
    if train:
        return fin.signals, fin.contexts
