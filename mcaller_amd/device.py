"""One MI355X: the resident event table, marked reference, read qualities, classifier, and the hot path."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Params, Records, check, lib, make_ref_view, _ptr


def default_device_index():
    for var in ('MCALLER_DEVICE', 'LOCAL_RANK'):
        if os.environ.get(var, '') != '':
            return int(os.environ[var])
    return 0


def _serialized(fn):
    """A method that calls the library on the context: one at a time (the C ABI's contract)."""
    import functools

    @functools.wraps(fn)
    def call(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)
    return call


class Device(object):
    def __init__(self, index=None):
        import threading
        self._lock = threading.RLock()      # calls on a context are serialized (include/mcaller_hip.h): the stream's formatter thread scores a stray record while the main thread waits
        self.index = default_device_index() if index is None else int(index)
        self._ctx = C.c_void_p()
        check(lib().mc_ctx_create(self.index, C.byref(self._ctx)))
        self._keep = {}

    @staticmethod
    def bind_host_to_numa_node(device):
        """One process per GPU: bind this process to the cores next to `device` (-> NUMA node, or -1: unknown, unchanged)."""
        return int(lib().mc_bind_to_device_numa_node(int(device)))

    def close(self):
        with self._lock:
            if self._ctx:
                lib().mc_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()

    def _row_text_release(self, block):
        """A RowText gives its pinned block back (any thread; nothing to do once the context is gone)."""
        with self._lock:
            if self._ctx:
                lib().mc_row_text_release(self._ctx, int(block))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- resident inputs ----
    @_serialized
    def set_reference(self, arrays):
        v = make_ref_view(arrays)
        check(lib().mc_ctx_set_reference(self._ctx, C.byref(v)))

    @_serialized
    def set_reference_motif(self, arrays, motif_fwd, repl_fwd, motif_rev, repl_rev):
        """The reference from its raw bases, site masks made on the GPU (arrays: MarkedReference.raw_arrays(); motifs:
        MarkedReference.motif_for_the_device())."""
        v = _lib.make_ref_view(arrays)
        v.n_words = int(arrays['n_words'])
        check(lib().mc_ctx_set_reference_motif(self._ctx, C.byref(v), motif_fwd, repl_fwd, len(motif_fwd), motif_rev, repl_rev,
                                               len(motif_rev)))

    @_serialized
    def fetch_reference(self, n_seq_bytes, n_words, n_contigs):
        """(seq, mbits_fwd, mbits_rev, rank_fwd, rank_rev, site_base, n_sites) as the device holds them (tests)."""
        seq = np.empty(n_seq_bytes, dtype=np.uint8)
        mf, mr = np.empty(n_words, dtype=np.uint32), np.empty(n_words, dtype=np.uint32)
        rf, rr = np.empty(n_words, dtype=np.int32), np.empty(n_words, dtype=np.int32)
        base, n_sites = np.empty(2 * n_contigs, dtype=np.int64), C.c_int64(0)
        check(lib().mc_ctx_fetch_reference(self._ctx, _ptr(seq), int(n_seq_bytes), _ptr(mf), _ptr(mr), _ptr(rf), _ptr(rr), int(n_words),
                                           _ptr(base), C.byref(n_sites)))
        return seq, mf, mr, rf, rr, base, n_sites.value

    @_serialized
    def upload_table(self, table):
        v = table.view()
        check(lib().mc_ctx_upload_table(self._ctx, C.byref(v)))
        self.n_rows = table.n_rows
        return self.current_slot()

    @_serialized
    def current_slot(self):
        return int(lib().mc_ctx_current_slot(self._ctx))

    @_serialized
    def select_table(self, slot, as_new=False):
        """Make the table resident in `slot` the current one again; as_new: the next pass does everything the first pass over a
        table does (every row validated), whatever earlier passes learned about it."""
        check(lib().mc_ctx_select_table(self._ctx, int(slot), 1 if as_new else 0))

    @_serialized
    def reserve_tables(self, max_rows, max_segs, max_reads):
        """Size the table slots, the per-pass scratch and the record sets once for a stream of tables up to these sizes."""
        check(lib().mc_ctx_reserve_tables(self._ctx, int(max_rows), int(max_segs), int(max_reads)))

    @_serialized
    def upload_table_async(self, table, qual=None):
        """Enqueue the upload of `table` (+ its read qualities) into a free slot and make it the current table; returns the
        slot.  The table's arrays must stay alive and untouched until wait_upload(slot) (or until the records of a pass over
        it have been handed out); they should be pinned (Table.pinned(), or parsed with the pool switched on)."""
        q = None if qual is None else np.ascontiguousarray(qual, dtype=np.float64)
        if getattr(table, 'device_slot', None) is not None:          # parsed on the device: the columns are in the slot already
            sr = np.ascontiguousarray(table.seg_read, dtype=np.int32)
            check(lib().mc_ctx_parse_finish(self._ctx, int(table.device_slot), _ptr(sr), int(table.n_reads),
                                            None if q is None else _ptr(q)))
            self.n_rows = table.n_rows
            slot, table.device_slot = table.device_slot, None
            return slot
        v = table.view()
        slot = C.c_int32(-1)
        check(lib().mc_ctx_upload_table_async(self._ctx, C.byref(v), None if q is None else _ptr(q), C.byref(slot)))
        self.n_rows = table.n_rows
        return slot.value

    # ---- the eventalign text parsed on the device (mc_ctx_parse_*) ----
    @_serialized
    def parse_begin(self, text, contig_names, max_rows):
        """Send a TextBlock and enqueue the parse into a free table slot -> slot."""
        arr = (C.c_char_p * max(1, len(contig_names)))()
        for i, n in enumerate(contig_names):
            arr[i] = n.encode('utf-8')
        slot = C.c_int32(-1)
        check(lib().mc_ctx_parse_begin(self._ctx, text.ptr, int(text.n_bytes), arr, len(contig_names), int(max_rows), C.byref(slot)))
        return slot.value

    @_serialized
    def parse_end(self, slot, text):
        """-> the Table (columns on the device, in `slot`; upload_table_async finishes it), or None: the shard needs the host
        parser (the slot has been given back)."""
        res = _lib.DevParseResult()
        check(lib().mc_ctx_parse_end(self._ctx, int(slot), C.byref(res)))
        if res.status != 0:
            self.parse_fallback_reason = lib().mc_last_error().decode('utf-8', 'replace')
            check(lib().mc_ctx_parse_abandon(self._ctx, int(slot)))
            return None
        t = _lib.device_table(res, text)
        t.device_slot = int(slot)
        return t

    @_serialized
    def parse_abandon(self, slot):
        check(lib().mc_ctx_parse_abandon(self._ctx, int(slot)))

    @_serialized
    def fetch_columns(self, slot, n_rows):
        """(pos, evmu [n, 2], event_idx, flags) of the table in `slot`, copied back (tests)."""
        pos, evmu = np.empty(n_rows, dtype=np.int32), np.empty((n_rows, 2), dtype=np.int32)
        idx, fl = np.empty(n_rows, dtype=np.int32), np.empty(n_rows, dtype=np.uint8)
        check(lib().mc_ctx_fetch_columns(self._ctx, int(slot), int(n_rows), _ptr(pos), _ptr(evmu), _ptr(idx), _ptr(fl)))
        return pos, evmu, idx, fl

    @_serialized
    def wait_upload(self, slot):
        check(lib().mc_ctx_wait_upload(self._ctx, int(slot)))

    @_serialized
    def upload_times_ms(self, slot):
        """(H2D ms, 0.0) of the last upload into `slot`; waits for it.  (Nothing runs at upload: the first pass validates.)"""
        a, b = C.c_float(0), C.c_float(0)
        check(lib().mc_ctx_upload_times_ms(self._ctx, int(slot), C.byref(a), C.byref(b)))
        return a.value, b.value

    @_serialized
    def parse_times_ms(self, slot):
        """(text H2D ms, device parser ms) of the parse begun into `slot` (between parse_begin and parse_end / parse_abandon)."""
        a, b = C.c_float(0), C.c_float(0)
        check(lib().mc_ctx_parse_times_ms(self._ctx, int(slot), C.byref(a), C.byref(b)))
        return a.value, b.value

    @_serialized
    def set_read_quality(self, qual):
        q = np.ascontiguousarray(qual, dtype=np.float64)
        check(lib().mc_ctx_set_read_quality(self._ctx, _ptr(q), len(q)))

    @_serialized
    def set_mlp(self, weights, submodel_of_char):
        """weights: list of MLPWeights (same shapes); submodel_of_char: uint8[256]."""
        n_in, n_hidden = weights[0].n_in, weights[0].n_hidden
        for w in weights:
            if (w.n_in, w.n_hidden) != (n_in, n_hidden):
                raise NotImplementedError('sub-models with different shapes')
        W1 = np.ascontiguousarray(np.stack([w.W1 for w in weights]), dtype=np.float64)
        b1 = np.ascontiguousarray(np.stack([w.b1 for w in weights]), dtype=np.float64)
        W2 = np.ascontiguousarray(np.stack([w.W2 for w in weights]), dtype=np.float64)
        b2 = np.ascontiguousarray(np.concatenate([w.b2 for w in weights]), dtype=np.float64)
        soc = np.ascontiguousarray(submodel_of_char, dtype=np.uint8)
        assert soc.shape == (256,)
        check(lib().mc_ctx_set_mlp(self._ctx, len(weights), n_in, n_hidden, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
                                   _ptr(soc)))

    @_serialized
    def set_forest(self, forests, submodel_of_char):
        """forests: list of ForestWeights (one per sub-model)."""
        arr = forest_arrays(forests)
        soc = np.ascontiguousarray(submodel_of_char, dtype=np.uint8)
        check(lib().mc_ctx_set_forest(self._ctx, len(forests), forests[0].n_in, _ptr(arr['model_tree_off']),
                                      _ptr(arr['tree_node_off']), _ptr(arr['left']), _ptr(arr['right']),
                                      _ptr(arr['feature']), _ptr(arr['threshold']), _ptr(arr['value']), _ptr(soc)))
        self._clf = 'forest'

    @_serialized
    def set_simple(self, models, submodel_of_char):
        """models: list of LogisticWeights or of GaussianNBWeights (one per sub-model) -- `-c LR` / `-c NBC`."""
        kind = {'logistic': 1, 'gnb': 2}[models[0].kind]
        if any(m.kind != models[0].kind or m.n_in != models[0].n_in for m in models):
            raise NotImplementedError('sub-models of different kinds or shapes')
        params = np.ascontiguousarray(np.stack([m.params() for m in models]), dtype=np.float64)
        soc = np.ascontiguousarray(submodel_of_char, dtype=np.uint8)
        check(lib().mc_ctx_set_simple_classifier(self._ctx, kind, len(models), models[0].n_in, _ptr(params), params.shape[1], _ptr(soc)))
        self._clf = 'simple'

    def set_classifier(self, weights, submodel_of_char):
        """MLP, forest, logistic regression or naive Bayes, whatever the model file held (extract_contexts.py:199 calls any
        of them the same way)."""
        if weights[0].kind == 'forest':
            self.set_forest(weights, submodel_of_char)
        elif weights[0].kind in ('logistic', 'gnb'):
            self.set_simple(weights, submodel_of_char)
        else:
            self.set_mlp(weights, submodel_of_char)
            self._clf = 'mlp'

    @_serialized
    def classifier_forward(self, X, submodel):
        X = np.ascontiguousarray(X, dtype=np.float64)
        sm = np.ascontiguousarray(submodel, dtype=np.uint8)
        p = np.empty(len(X), dtype=np.float64)
        fn = {'forest': lib().mc_forest_forward, 'simple': lib().mc_simple_forward}.get(getattr(self, '_clf', 'mlp'), lib().mc_mlp_forward)
        check(fn(self._ctx, _ptr(X), _ptr(sm), len(X), _ptr(p)))
        return p

    # ---- the hot path ----
    @_serialized
    def run(self, k, skip_thresh, qual_thresh, tail_contig=-1, score=True, entry_read=-1, entry_first_idx=0):
        """K0+K1+K2 on the resident table; records stay on the device.  Returns their number."""
        p = Params(int(k), int(skip_thresh), float(qual_thresh), int(tail_contig), 1 if score else 0,
                   int(entry_read), int(entry_first_idx))
        n = C.c_int64(0)
        check(lib().mc_extract_features(self._ctx, C.byref(p), C.byref(n)))
        self._last = (n.value, int(k))
        return n.value

    @_serialized
    def run_async(self, k, skip_thresh, qual_thresh, tail_contig=-1, score=True, entry_read=-1, entry_first_idx=0):
        """Enqueue one pass (K0 + K1 on the ctx stream, the emit, K2 + packing on a side stream); at most six in flight."""
        p = Params(int(k), int(skip_thresh), float(qual_thresh), int(tail_contig), 1 if score else 0,
                   int(entry_read), int(entry_first_idx))
        check(lib().mc_extract_features_async(self._ctx, C.byref(p)))
        self._async_k = getattr(self, '_async_k', []) + [int(k)]

    @_serialized
    def wait_begin(self):
        """Start the copy-out of the oldest pass whose copy-out has not been started, without waiting for it."""
        check(lib().mc_wait_records_begin(self._ctx))

    @_serialized
    def wait(self):
        """Records of the oldest pass in flight: views of pinned buffers (slot means / probabilities of the calls only, see
        Records.call_row), valid until six more passes have been enqueued."""
        n, v = C.c_int64(0), _lib.CallsView()
        check(lib().mc_wait_records(self._ctx, C.byref(n), C.byref(v)))
        k = self._async_k.pop(0)
        self._last = (n.value, k)
        rerun = C.c_int32(0)
        check(lib().mc_last_pass_info(self._ctx, None, C.byref(rerun)))
        if rerun.value:
            # a pass the library repeated synchronously (irregular reads, record buffers too small) hands out the context's ONE set
            # of buffers for synchronous runs, which the next such pass overwrites -- and the stream's formatter thread reads a
            # shard's records while the main thread waits for the next pass: such records are copied (they are rare)
            rec = Records(n.value, k)
            vc = rec.view()
            check(lib().mc_fetch_records(self._ctx, C.byref(vc)))
            rec.n = n.value
            return rec
        rec = Records.from_view(v, n.value, k, self)
        text, nb, nr, block = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int32(-1)
        check(lib().mc_last_row_text(self._ctx, C.byref(text), C.byref(nb), C.byref(nr), C.byref(block)))
        if block.value >= 0:                 # the rows as text, made on the device (row_text): the host formatter has nothing to do
            rec.row_text = _lib.RowText(self, text.value, nb.value, nr.value, block.value)
        return rec

    @_serialized
    def row_text(self, on, label_meth=None, label_unmeth=None, first=False):
        """The passes enqueued from now on also write their rows as text on the device (mc_ctx_row_text): wait() hands them out as
        Records.row_text when the pass had them.  first: the start of a stream -- blocks an earlier stream never gave back are taken
        back (nobody reads them any more)."""
        check(lib().mc_ctx_row_text(self._ctx, (2 if first else 1) if on else 0, label_meth.encode() if label_meth else None,
                                    label_unmeth.encode() if label_unmeth else None))

    @_serialized
    def fetch(self, copy=True):
        """Records of the last run.  copy=False: views of the context's pinned buffers (overwritten by the next run)."""
        n, k = self._last
        if not copy:
            v = _lib.CallsView()
            check(lib().mc_fetch_records_view(self._ctx, C.byref(v)))
            return Records.from_view(v, n, k, self)
        rec = Records(n, k)
        v = rec.view()
        check(lib().mc_fetch_records(self._ctx, C.byref(v)))
        rec.n = n
        return rec

    def extract(self, k, skip_thresh, qual_thresh, **kw):
        self.run(k, skip_thresh, qual_thresh, **kw)
        return self.fetch()

    @_serialized
    def sync(self):
        """hipDeviceSynchronize on this context's device (all of its streams)."""
        check(lib().mc_ctx_sync(self._ctx))

    @_serialized
    def set_pass_timing(self, every_n):
        """Pipelined passes: record the timing events with every n-th pass only (each costs the queue ~9 us); 0 = never."""
        check(lib().mc_ctx_set_pass_timing(self._ctx, int(every_n)))

    @_serialized
    def last_pass_timed(self):
        return bool(lib().mc_last_pass_timed(self._ctx))

    @_serialized
    def last_pass_info(self):
        """(record slots per piece if the pass handed out last ran as the fused dense kernel, else 0; whether it was repeated
        synchronously inside wait())."""
        a, b = C.c_int32(0), C.c_int32(0)
        check(lib().mc_last_pass_info(self._ctx, C.byref(a), C.byref(b)))
        return a.value, bool(b.value)

    @_serialized
    def times_ms(self):
        t = np.zeros(5, dtype=np.float32)
        check(lib().mc_last_times_ms(self._ctx, _ptr(t)))
        return dict(strand_resolve=float(t[0]), window_scan=float(t[1]), emit=float(t[2]), classifier=float(t[3]),
                    total=float(t[4]))

    # ---- per-site reduction feeding make_bed (the one exchange step of a multi-GPU job) ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId (call on rank 0, ship the 128 bytes to every rank)."""
        buf = np.zeros(128, dtype=np.uint8)
        check(lib().mc_comm_unique_id(_ptr(buf)))
        return buf.tobytes()

    @staticmethod
    def comm_probe():
        """Can librccl.so be loaded in this process?  Raises if not (nothing else is touched)."""
        check(lib().mc_comm_available())

    @_serialized
    def site_counts_fetch(self):
        """This rank's own per-site counts as they stand (no collective) -> (n_meth, n_total, first)."""
        n = lib().mc_site_count(self._ctx)
        n_meth, n_total = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
        check(lib().mc_site_counts_fetch(self._ctx, _ptr(n_meth), _ptr(n_total), _ptr(first)))
        return n_meth, n_total, first

    @_serialized
    def comm_init(self, world, rank, unique_id):
        """ncclCommInitRank on this GPU (RCCL over xGMI); collective over all ranks."""
        uid = np.frombuffer(unique_id, dtype=np.uint8).copy()
        assert len(uid) == 128
        check(lib().mc_comm_init(self._ctx, int(world), int(rank), _ptr(uid)))

    @_serialized
    def comm_destroy(self):
        lib().mc_comm_destroy(self._ctx)

    @_serialized
    def site_counts(self, row_offset=0, tail_contig=-1):
        """Per-site counts of the last run's records, on the device.  Returns how many records the host scored itself
        (NaN probability on the device): add those with site_counts_add.  Records closed by a row of another contig than
        their site's are left out too (self.n_cross_contig; make_bed.cross_contig_records lists them)."""
        pending, cross = C.c_int64(0), C.c_int64(0)
        check(lib().mc_site_counts(self._ctx, int(row_offset), int(tail_contig), C.byref(pending), C.byref(cross)))
        self.n_cross_contig = cross.value
        return pending.value

    @_serialized
    def site_counts_reset(self):
        """Zero the device-side per-site counts (before the first shard of a streamed file)."""
        check(lib().mc_site_counts_reset(self._ctx))

    @_serialized
    def site_counts_accumulate(self, row_offset=0, tail_contig=-1):
        """Add the records of the pass handed out last to the per-site counts (a streamed file: shard after shard); returns
        like site_counts."""
        pending, cross = C.c_int64(0), C.c_int64(0)
        check(lib().mc_site_counts_accumulate(self._ctx, int(row_offset), int(tail_contig), C.byref(pending), C.byref(cross)))
        self.n_cross_contig = cross.value
        return pending.value

    @_serialized
    def site_counts_add(self, site, is_meth, first_row):
        site = np.ascontiguousarray(site, dtype=np.int64)
        meth = np.ascontiguousarray(is_meth, dtype=np.uint8)
        first = np.ascontiguousarray(first_row, dtype=np.int64)
        check(lib().mc_site_counts_add(self._ctx, _ptr(site), _ptr(meth), _ptr(first), len(site)))

    @_serialized
    def site_allreduce(self):
        """Sum / min over the ranks of the communicator (none: this rank alone) -> (n_meth, n_total, first, ms)."""
        n = lib().mc_site_count(self._ctx)
        n_meth, n_total = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
        ms = C.c_float(0)
        check(lib().mc_site_allreduce(self._ctx, _ptr(n_meth), _ptr(n_total), _ptr(first), C.byref(ms)))
        return n_meth, n_total, first, ms.value

    # ---- the classifier fit behind --train ----
    @_serialized
    def mlp_fit(self, X, y, jobs, hidden=100, alpha=0.001, lr_init=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8,
                batch_size=200, max_iter=200, tol=1e-4, n_iter_no_change=10, shuffle=True, seed=1, seeds=None, init=None):
        """Fit one 7-H-1 tanh/logistic perceptron per job on the GPU (mc_mlp_fit; all jobs side by side).
        jobs: [(train_rows, validation_rows)] index arrays into X / y.  -> list of dicts W1, b1, W2, b2, loss_curve,
        n_iter, val_correct, n_val."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.uint8)
        n, d = X.shape
        nj = len(jobs)
        tr = [np.ascontiguousarray(j[0], dtype=np.int32) for j in jobs]
        va = [np.ascontiguousarray(j[1], dtype=np.int32) for j in jobs]
        tr_off = np.concatenate([[0], np.cumsum([len(a) for a in tr])]).astype(np.int64)
        va_off = np.concatenate([[0], np.cumsum([len(a) for a in va])]).astype(np.int64)
        tr_idx = np.ascontiguousarray(np.concatenate(tr + [np.zeros(1, np.int32)]))
        va_idx = np.ascontiguousarray(np.concatenate(va + [np.zeros(1, np.int32)]))
        prm = _lib.FitParams(d, int(hidden), int(batch_size), int(max_iter), int(n_iter_no_change), 1 if shuffle else 0,
                             float(alpha), float(lr_init), float(beta1), float(beta2), float(epsilon), float(tol), int(seed))
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        ini = None
        if init is not None:      # [(W1, b1, W2, b2)] per job
            ini = np.ascontiguousarray(np.concatenate([np.concatenate([np.ravel(w[0]), np.ravel(w[1]), np.ravel(w[2]),
                                                                       np.ravel([w[3]])]) for w in init]), dtype=np.float64)
            assert len(ini) == nj * (d * hidden + 2 * hidden + 1)
        W1 = np.zeros((nj, d, hidden)); b1 = np.zeros((nj, hidden)); W2 = np.zeros((nj, hidden)); b2 = np.zeros(nj)
        curve = np.zeros((nj, max_iter)); n_iter = np.zeros(nj, dtype=np.int32); correct = np.zeros(nj, dtype=np.int64)
        check(lib().mc_mlp_fit(self._ctx, C.byref(prm), _ptr(X), _ptr(y), n, nj, _ptr(tr_off), _ptr(tr_idx), _ptr(va_off),
                               _ptr(va_idx), None if sd is None else _ptr(sd), None if ini is None else _ptr(ini),
                               _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(curve), _ptr(n_iter), _ptr(correct)))
        return [dict(W1=W1[j], b1=b1[j], W2=W2[j], b2=float(b2[j]), loss_curve=curve[j, :n_iter[j]].copy(),
                     n_iter=int(n_iter[j]), val_correct=int(correct[j]), n_val=len(va[j])) for j in range(nj)]

    @_serialized
    def mlp_forward(self, X, submodel):
        if getattr(self, '_clf', 'mlp') != 'mlp':
            return self.classifier_forward(X, submodel)
        X = np.ascontiguousarray(X, dtype=np.float64)
        sm = np.ascontiguousarray(submodel, dtype=np.uint8)
        p = np.empty(len(X), dtype=np.float64)
        check(lib().mc_mlp_forward(self._ctx, _ptr(X), _ptr(sm), len(X), _ptr(p)))
        return p


def forest_arrays(forests):
    """Concatenate the sub-models' trees into the flat arrays of mc_ctx_set_forest."""
    model_tree_off, tree_node_off = [0], [0]
    left, right, feature, threshold, value = [], [], [], [], []
    for f in forests:
        base = tree_node_off[-1]
        left.append(np.where(f.left >= 0, f.left + base, -1))
        right.append(np.where(f.right >= 0, f.right + base, -1))
        feature.append(f.feature)
        threshold.append(f.threshold)
        value.append(f.value)
        tree_node_off.extend((f.tree_off[1:] + base).tolist())
        model_tree_off.append(model_tree_off[-1] + f.n_trees)
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs), dtype=dt)
    return dict(model_tree_off=np.asarray(model_tree_off, dtype=np.int32), tree_node_off=np.asarray(tree_node_off, dtype=np.int32),
                left=cat(left, np.int32), right=cat(right, np.int32), feature=cat(feature, np.int32),
                threshold=cat(threshold, np.float64), value=cat([v.reshape(-1) for v in value], np.float64))


_devices = {}


def get_device(index=None):
    index = default_device_index() if index is None else int(index)
    if index not in _devices:
        _devices[index] = Device(index)
    return _devices[index]
