"""ctypes binding of libmcaller_hip.so (C ABI: include/mcaller_hip.h).

The library holds the native eventalign parser and the HIP kernels.  There is no fallback: if the
shared object is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MCALLER_LIB') or os.path.join(_HERE, 'libmcaller_hip.so')      # (MCALLER_LIB: a variant build, tools/variants.sh)

MC_MAX_K = 8
F_KMER_EQ, F_MODEL_N, F_SEG_START, F_NAME_START = 1, 2, 4, 8
I_EMPTY_MASK, I_REV, I_TOO_MANY, I_MULTI, I_EDGE, I_NEXT_SHIFT = 0xFF, 0x100, 0x200, 0x400, 0x800, 16


E_NO_FREE_SLOT = -16      # include/mcaller_hip.h: MC_E_NO_FREE_SLOT


class McError(RuntimeError):
    """An error of the library; `code`: what the call returned."""
    code = None


class TableView(C.Structure):
    _fields_ = [('n_rows', C.c_int64),
                ('pos', C.c_void_p), ('event_model_e4', C.c_void_p),
                ('event_idx', C.c_void_p), ('flags', C.c_void_p),
                ('n_seg', C.c_int32),
                ('seg_row_begin', C.c_void_p), ('seg_read', C.c_void_p), ('seg_contig', C.c_void_p),
                ('n_reads', C.c_int32)]


class RefView(C.Structure):
    _fields_ = [('n_contigs', C.c_int32),
                ('contig_len', C.c_void_p), ('seq_off', C.c_void_p), ('seq', C.c_void_p),
                ('word_off', C.c_void_p), ('mbits_fwd', C.c_void_p), ('mbits_rev', C.c_void_p),
                ('n_seq_bytes', C.c_int64), ('n_words', C.c_int64)]


class DevParseResult(C.Structure):
    """mc_devparse_result (include/mcaller_hip.h)."""
    _fields_ = [('status', C.c_int32), ('n_lines', C.c_int64), ('n_rows', C.c_int64), ('n_seg', C.c_int32), ('n_unknown', C.c_int32),
                ('seg_row_begin', C.c_void_p), ('seg_contig', C.c_void_p), ('seg_name_off', C.c_void_p), ('seg_name_len', C.c_void_p),
                ('seg_name_start', C.c_void_p), ('unknown_off', C.c_void_p), ('unknown_len', C.c_void_p), ('flags', C.c_void_p)]


class CallsView(C.Structure):
    _fields_ = [('capacity', C.c_int64),
                ('feats', C.c_void_p), ('site_pos', C.c_void_p), ('site_seg', C.c_void_p),
                ('close_row', C.c_void_p), ('info', C.c_void_p), ('prob', C.c_void_p),
                ('call_row', C.c_void_p), ('n_call_rows', C.c_int64), ('close_row32', C.c_void_p), ('compacted', C.c_int32),
                ('feats_lo32', C.c_void_p), ('feats_hi32', C.c_void_p), ('feats_wide', C.c_void_p), ('n_wide', C.c_int64)]


class FormatArgs(C.Structure):
    _fields_ = [('rec', C.POINTER(CallsView)), ('n_records', C.c_int64), ('k', C.c_int32),
                ('table', C.POINTER(TableView)), ('ref', C.POINTER(RefView)),
                ('contig_names', C.POINTER(C.c_char_p)), ('read_names', C.POINTER(C.c_char_p)),
                ('read_qual_txt', C.POINTER(C.c_char_p)), ('tail_chrom', C.c_char_p),
                ('label_meth', C.c_char_p), ('label_unmeth', C.c_char_p), ('submodel_of_char', C.c_void_p)]


class FitParams(C.Structure):
    _fields_ = [('n_in', C.c_int32), ('n_hidden', C.c_int32), ('batch_size', C.c_int32), ('max_iter', C.c_int32),
                ('n_iter_no_change', C.c_int32), ('shuffle', C.c_int32),
                ('alpha', C.c_double), ('lr_init', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double),
                ('epsilon', C.c_double), ('tol', C.c_double), ('seed', C.c_uint64)]


class Params(C.Structure):
    _fields_ = [('k', C.c_int32), ('skip_thresh', C.c_int32), ('qual_thresh', C.c_double),
                ('tail_contig', C.c_int32), ('score', C.c_int32),
                ('entry_read', C.c_int32), ('entry_first_idx', C.c_int32)]


_lib = None


def lib():
    """Load libmcaller_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.mc_last_error.restype = C.c_char_p
        L.mc_version.restype = C.c_char_p
        L.mc_parse_eventalign.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_char_p), C.c_int32,
                                          C.c_int32, C.POINTER(C.c_void_p)]
        L.mc_parse_eventalign_range.argtypes = L.mc_parse_eventalign.argtypes
        L.mc_eventalign_read_cuts.argtypes = [C.c_char_p, C.c_int32, C.c_void_p]
        L.mc_eventalign_read_cuts_range.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
        L.mc_eventalign_read_cuts_at.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.mc_eventalign_consumed_range.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mc_fastq_read_quality.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.mc_fastq_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.mc_fastq_view.restype = C.c_int64
        L.mc_fastq_free.argtypes = [C.c_void_p]
        L.mc_parsed_view.argtypes = [C.c_void_p, C.POINTER(TableView)]
        L.mc_parsed_read_name.argtypes = [C.c_void_p, C.c_int32]
        L.mc_parsed_read_name.restype = C.c_char_p
        L.mc_parsed_n_unknown.argtypes = [C.c_void_p]
        L.mc_parsed_n_unknown.restype = C.c_int64
        L.mc_parsed_unknown_name.argtypes = [C.c_void_p, C.c_int64]
        L.mc_parsed_unknown_name.restype = C.c_char_p
        L.mc_parsed_free.argtypes = [C.c_void_p]
        L.mc_parsed_free.restype = None
        L.mc_parsed_n_pieces.argtypes = [C.c_void_p]
        L.mc_synth_write_tsv.argtypes = [C.c_char_p, C.POINTER(TableView), C.c_char_p, C.c_int64, C.c_char_p,
                                         C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_int64)]
        L.mc_host_alloc.argtypes = [C.c_int64]
        L.mc_host_alloc.restype = C.c_void_p
        L.mc_host_free.argtypes = [C.c_void_p]
        L.mc_host_free.restype = None
        L.mc_host_is_pinned.argtypes = [C.c_void_p]
        L.mc_host_pool_config.argtypes = [C.c_int32, C.c_int64]
        L.mc_ctx_reserve_tables.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32]
        L.mc_mark_motifs.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_ctx_set_reference_motif.argtypes = [C.c_void_p, C.POINTER(RefView), C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p,
                                                 C.c_int32]
        L.mc_ctx_fetch_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_void_p, C.POINTER(C.c_int64)]
        L.mc_read_file_range.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32]
        L.mc_map_file_range.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.mc_unmap_file_range.argtypes = [C.c_void_p]
        L.mc_unmap_file_range.restype = None
        L.mc_ctx_parse_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.c_int32, C.c_int64,
                                         C.POINTER(C.c_int32)]
        L.mc_ctx_parse_end.argtypes = [C.c_void_p, C.c_int32, C.POINTER(DevParseResult)]
        L.mc_ctx_parse_finish.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.mc_ctx_parse_abandon.argtypes = [C.c_void_p, C.c_int32]
        L.mc_ctx_fetch_columns.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_ctx_upload_table_async.argtypes = [C.c_void_p, C.POINTER(TableView), C.c_void_p, C.POINTER(C.c_int32)]
        L.mc_ctx_wait_upload.argtypes = [C.c_void_p, C.c_int32]
        L.mc_ctx_current_slot.argtypes = [C.c_void_p]
        L.mc_ctx_select_table.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.mc_ctx_upload_times_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mc_last_pass_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mc_ctx_parse_times_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mc_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mc_ctx_destroy.argtypes = [C.c_void_p]
        L.mc_ctx_destroy.restype = None
        L.mc_ctx_set_reference.argtypes = [C.c_void_p, C.POINTER(RefView)]
        L.mc_ctx_upload_table.argtypes = [C.c_void_p, C.POINTER(TableView)]
        L.mc_ctx_set_read_quality.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.mc_ctx_set_mlp.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_extract_features.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_int64)]
        L.mc_fetch_records.argtypes = [C.c_void_p, C.POINTER(CallsView)]
        L.mc_fetch_records_view.argtypes = [C.c_void_p, C.POINTER(CallsView)]
        L.mc_extract_features_async.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.mc_wait_records.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(CallsView)]
        L.mc_wait_records_begin.argtypes = [C.c_void_p]
        L.mc_last_times_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_ctx_sync.argtypes = [C.c_void_p]
        L.mc_ctx_set_pass_timing.argtypes = [C.c_void_p, C.c_int]
        L.mc_last_pass_timed.argtypes = [C.c_void_p]
        L.mc_bind_to_device_numa_node.argtypes = [C.c_int]
        L.mc_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mc_forest_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mc_ctx_set_forest.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 8
        L.mc_simple_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mc_ctx_set_simple_classifier.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.mc_site_count.argtypes = [C.c_void_p]
        L.mc_site_count.restype = C.c_int64
        L.mc_site_counts.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mc_site_counts_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.mc_site_counts_reset.argtypes = [C.c_void_p]
        L.mc_site_counts_accumulate.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mc_comm_unique_id.argtypes = [C.c_void_p]
        L.mc_comm_available.argtypes = []
        L.mc_site_counts_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.mc_comm_destroy.argtypes = [C.c_void_p]
        L.mc_site_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.mc_mlp_fit.argtypes = [C.c_void_p, C.POINTER(FitParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 13
        L.mc_calls_expand.argtypes = [C.POINTER(CallsView), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_count_records.argtypes = [C.POINTER(CallsView), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p,
                                       C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mc_format_diffs.argtypes = [C.POINTER(FormatArgs), C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mc_free.argtypes = [C.c_void_p]
        L.mc_free.restype = None
        L.mc_repr_double.argtypes = [C.c_double, C.c_char_p]
        L.mc_repr_double_rowtext.argtypes = [C.c_double, C.c_char_p]
        L.mc_ctx_row_text.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_char_p]
        L.mc_last_row_text.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.mc_row_text_release.argtypes = [C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        e = McError(lib().mc_last_error().decode('utf-8', 'replace') or 'libmcaller_hip error %d' % rc)
        e.code = rc
        raise e


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _from_ptr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class PinnedArray(object):
    """A numpy array over a block of the library's pinned host pool (mc_host_alloc); the block goes back to the pool
    when the object dies."""

    def __init__(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().mc_host_alloc(max(n, 1) + 64)
        if not self.ptr:
            raise MemoryError('mc_host_alloc(%d)' % n)
        buf = (C.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                lib().mc_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class Table(object):
    """Columnar event table on the host: numpy arrays + a TableView over them.  The event and model currents are one
    column of (event, model) pairs, `evmu` [n, 2]; `event_e4` / `model_e4` are its two halves (strided views)."""

    def __init__(self, pos, event_e4, model_e4, event_idx, flags, seg_row_begin, seg_read, seg_contig, n_reads,
                 read_names=None, unknown=(), owner=None, evmu=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        if evmu is None:
            evmu = np.empty((len(self.pos), 2), dtype=np.int32)
            evmu[:, 0] = event_e4
            evmu[:, 1] = model_e4
        self.evmu = np.ascontiguousarray(evmu, dtype=np.int32).reshape(-1, 2)
        self.event_idx = np.ascontiguousarray(event_idx, dtype=np.int32)
        self.flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.seg_row_begin = np.ascontiguousarray(seg_row_begin, dtype=np.int64)
        self.seg_read = np.ascontiguousarray(seg_read, dtype=np.int32)
        self.seg_contig = np.ascontiguousarray(seg_contig, dtype=np.int32)
        self.n_reads = int(n_reads)
        self.read_names = read_names
        self.unknown = list(unknown)
        self._owner = owner
        self.n_rows = len(self.pos)
        self.n_seg = len(self.seg_read)

    @property
    def event_e4(self):
        return self.evmu[:, 0]

    @property
    def model_e4(self):
        return self.evmu[:, 1]

    def pinned(self):
        """A copy whose columns sit in the pinned host pool (what mc_ctx_upload_table_async wants to read from)."""
        keep = []

        def pin(a):
            h = PinnedArray(a.shape, a.dtype)
            h.array[...] = a
            keep.append(h)
            return h.array
        t = Table(pin(self.pos), None, None, pin(self.event_idx), pin(self.flags), self.seg_row_begin, self.seg_read,
                  self.seg_contig, self.n_reads, read_names=self.read_names, unknown=self.unknown, evmu=pin(self.evmu))
        t._owner = keep
        return t

    def view(self):
        v = TableView()
        v.n_rows = self.n_rows
        if self.pos is not None:                       # (a device-parsed table has its columns in a table slot only)
            v.pos, v.event_model_e4 = _ptr(self.pos), _ptr(self.evmu)
            v.event_idx = _ptr(self.event_idx)
        v.flags = _ptr(self.flags)
        v.n_seg = self.n_seg
        v.seg_row_begin, v.seg_read, v.seg_contig = _ptr(self.seg_row_begin), _ptr(self.seg_read), _ptr(self.seg_contig)
        v.n_reads = self.n_reads
        return v

    def slice_segments(self, s0, s1):
        """Sub-table of segments [s0, s1) (a shard); row indices restart at 0, read ids are kept."""
        r0, r1 = int(self.seg_row_begin[s0]), int(self.seg_row_begin[s1])
        return Table(self.pos[r0:r1], None, None, self.event_idx[r0:r1],
                     self.flags[r0:r1], self.seg_row_begin[s0:s1 + 1] - r0, self.seg_read[s0:s1],
                     self.seg_contig[s0:s1], self.n_reads, read_names=self.read_names, owner=self._owner,
                     evmu=self.evmu[r0:r1])


class TextBlock(object):
    """A byte range of a file where the DMA engines can read it (what the device parser is given): mapped from the page cache
    and registered (mc_map_file_range: no copy by the CPU; MCALLER_READER=mmap), or -- the default, and where the mapping cannot
    be registered -- read into a block of the pinned host pool."""

    def __init__(self, path, lo, hi, n_threads=0):
        self.n_bytes = int(hi - lo)
        self._map = None
        self._mem = None
        if self.n_bytes and os.environ.get('MCALLER_READER', 'pread').startswith('mmap'):
            handle, ptr = C.c_void_p(), C.c_void_p()
            if lib().mc_map_file_range(path.encode('utf-8'), int(lo), int(hi), C.byref(handle), C.byref(ptr)) == 0:
                self._map = handle
                self.ptr = ptr.value
                self.array = np.frombuffer((C.c_char * self.n_bytes).from_address(self.ptr), dtype=np.uint8)
                return
        self._mem = PinnedArray((max(self.n_bytes, 1),), np.uint8)
        self.array = self._mem.array
        self.ptr = self._mem.ptr
        if self.n_bytes:
            check(lib().mc_read_file_range(path.encode('utf-8'), int(lo), int(hi), self.ptr, int(n_threads)))

    @property
    def mapped(self):
        return self._map is not None

    def token(self, off, n):
        return bytes(self.array[off:off + n]).decode('utf-8', 'surrogateescape')

    def __del__(self):
        try:
            if self._map is not None:
                self.array = None
                if os.environ.get('MCALLER_READER') != 'mmap_keep':      # (experiment: the mappings stay until the process ends)
                    lib().mc_unmap_file_range(self._map)
                self._map = None
        except Exception:       # noqa (interpreter shutdown)
            pass


def device_table(res, text):
    """Table whose columns the device parser has put into a table slot (mc_ctx_parse_end's result `res`, a DevParseResult;
    `text`: the TextBlock it was made from): the host side holds the segments, the flag column and the names only."""
    n, ns = int(res.n_rows), int(res.n_seg)
    t = Table.__new__(Table)
    t.pos = t.evmu = t.event_idx = None                # (on the device only)
    t.flags = _from_ptr(res.flags, n, np.uint8).copy()
    rows = _from_ptr(res.seg_row_begin, ns, np.int64)
    t.seg_row_begin = np.concatenate([rows, np.array([n], dtype=np.int64)])
    t.seg_contig = _from_ptr(res.seg_contig, ns, np.int32).copy()
    off, ln = _from_ptr(res.seg_name_off, ns, np.int64), _from_ptr(res.seg_name_len, ns, np.int32)
    ids, names, seg_read = {}, [], np.empty(ns, dtype=np.int32)
    for i in range(ns):                                # read ids in the order the names first appear, like the host parser
        name = text.token(int(off[i]), int(ln[i]))
        rid = ids.get(name)
        if rid is None:
            rid = ids[name] = len(names)
            names.append(name)
        seg_read[i] = rid
    t.seg_read = seg_read
    t.seg_name_start = _from_ptr(res.seg_name_start, ns, np.uint8).copy()
    t.n_reads, t.read_names = len(names), names
    uo, ul = _from_ptr(res.unknown_off, int(res.n_unknown), np.int64), _from_ptr(res.unknown_len, int(res.n_unknown), np.int32)
    t.unknown = [text.token(int(uo[i]), int(ul[i])) for i in range(int(res.n_unknown))]
    t._owner = None
    t.n_rows, t.n_seg = n, ns
    t.n_pieces = 0
    t.device_slot = None                               # set by Device.parse_end
    return t


def fastq_read_quality(path, n_threads=0):
    """Native FASTQ reader -> (keys: list of str, means: float64 array), one entry per record in file order."""
    L = lib()
    handle = C.c_void_p()
    if L.mc_fastq_read_quality(path.encode('utf-8'), int(n_threads), C.byref(handle)) != 0:
        raise ValueError(L.mc_last_error().decode('utf-8', 'replace'))
    try:
        pool, off, mean = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = L.mc_fastq_view(handle, C.byref(pool), C.byref(off), C.byref(mean))
        if n <= 0:
            return [], np.zeros(0, dtype=np.float64)
        offs = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_int64)), shape=(n + 1,))
        text = C.string_at(pool, int(offs[n])).decode('utf-8')
        means = np.ctypeslib.as_array(C.cast(mean, C.POINTER(C.c_double)), shape=(n,)).copy()
        return text.split('\n')[:n], means
    finally:
        L.mc_fastq_free(handle)


def eventalign_read_cuts(path, n_parts, lo=None, hi=None):
    """Byte offsets (n_parts + 1) cutting an eventalign file (or its byte range [lo, hi)) at the first lines of reads,
    pieces of similar size."""
    cuts = np.zeros(n_parts + 1, dtype=np.int64)
    if lo is None and hi is None:
        check(lib().mc_eventalign_read_cuts(path.encode('utf-8'), int(n_parts), _ptr(cuts)))
    else:
        check(lib().mc_eventalign_read_cuts_range(path.encode('utf-8'), int(lo or 0), int((1 << 62) if hi is None else hi),
                                                  int(n_parts), _ptr(cuts)))
    return [int(c) for c in cuts]


def eventalign_read_cuts_at(path, want, lo, hi):
    """Byte offsets cutting [lo, hi) at the first lines of the reads at or behind the offsets `want` (ascending): [lo, ..., end]."""
    want = np.ascontiguousarray(want, dtype=np.int64)
    cuts = np.zeros(len(want) + 2, dtype=np.int64)
    check(lib().mc_eventalign_read_cuts_at(path.encode('utf-8'), int(lo), int(hi), _ptr(want), len(want), _ptr(cuts)))
    return [int(c) for c in cuts]


def eventalign_consumed_range(path, startline, endline):
    """[lo, hi): the bytes the reference's batch loop reads for (startline, endline) (extract_contexts.py:141-148)."""
    lo, hi = C.c_int64(0), C.c_int64(0)
    check(lib().mc_eventalign_consumed_range(path.encode('utf-8'), int(startline), int(endline), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def parse_eventalign(path, startline, endline, contig_names, n_threads=0, exact_range=False):
    """Native parser -> Table over the library's buffers (no copy; freed with the Table)."""
    L = lib()
    arr = (C.c_char_p * max(1, len(contig_names)))()
    for i, n in enumerate(contig_names):
        arr[i] = n.encode('utf-8')
    handle = C.c_void_p()
    fn = L.mc_parse_eventalign_range if exact_range else L.mc_parse_eventalign
    check(fn(path.encode('utf-8'), int(startline), int(endline), arr, len(contig_names), int(n_threads), C.byref(handle)))
    owner = _Parsed(handle)          # the columns stay in the library's buffers (no copy); freed with the Table
    v = TableView()
    check(L.mc_parsed_view(handle, C.byref(v)))
    n, ns = v.n_rows, v.n_seg
    names = [L.mc_parsed_read_name(handle, i).decode('utf-8', 'surrogateescape') for i in range(v.n_reads)]
    unknown = [L.mc_parsed_unknown_name(handle, i).decode('utf-8', 'surrogateescape')
               for i in range(L.mc_parsed_n_unknown(handle))]
    t = Table(_from_ptr(v.pos, n, np.int32), None, None,
              _from_ptr(v.event_idx, n, np.int32), _from_ptr(v.flags, n, np.uint8),
              _from_ptr(v.seg_row_begin, ns + 1, np.int64), _from_ptr(v.seg_read, ns, np.int32),
              _from_ptr(v.seg_contig, ns, np.int32), v.n_reads, read_names=names, unknown=unknown, owner=owner,
              evmu=_from_ptr(v.event_model_e4, 2 * n, np.int32).reshape(-1, 2))
    t.n_pieces = int(L.mc_parsed_n_pieces(handle))
    return t


class _Parsed(object):
    """Owns a mc_parsed handle: the Table built over it keeps it alive."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                lib().mc_parsed_free(self.handle)
                self.handle = None
        except Exception:
            pass


def make_ref_view(arrays):
    """arrays: dict from MarkedReference.device_arrays() -> (RefView, keepalive)."""
    v = RefView()
    v.n_contigs = len(arrays['contig_len'])
    v.contig_len, v.seq_off, v.seq = _ptr(arrays['contig_len']), _ptr(arrays['seq_off']), _ptr(arrays['seq'])
    v.word_off, v.mbits_fwd, v.mbits_rev = _ptr(arrays['word_off']), _ptr(arrays['mbits_fwd']), _ptr(arrays['mbits_rev'])
    v.n_seq_bytes = len(arrays['seq'])
    v.n_words = len(arrays['mbits_fwd'])
    return v


def _cstr_array(strings):
    arr = (C.c_char_p * max(1, len(strings)))()
    for i, s in enumerate(strings):
        arr[i] = s if isinstance(s, bytes) else str(s).encode('utf-8', 'surrogateescape')
    return arr


class DiffsFormatter(object):
    """mc_format_diffs over one (records, table, reference): rows of records [first, stop) as bytes."""

    def __init__(self, rec, table, ref_arrays, contig_names, read_qual_txt, k, label_meth, label_unmeth,
                 submodel_of_char, tail_chrom=None):
        self._keep = (rec, table, ref_arrays)
        self._rv, self._tv, self._fv = rec.view(), table.view(), make_ref_view(ref_arrays)
        self._rv.capacity = rec.n
        self._contigs, self._reads = _cstr_array(contig_names), _cstr_array(table.read_names)
        self._quals = _cstr_array(read_qual_txt)
        self._soc = np.ascontiguousarray(submodel_of_char, dtype=np.uint8)
        a = FormatArgs()
        a.rec, a.n_records, a.k = C.pointer(self._rv), rec.n, k
        a.table, a.ref = C.pointer(self._tv), C.pointer(self._fv)
        a.contig_names, a.read_names, a.read_qual_txt = self._contigs, self._reads, self._quals
        a.tail_chrom = tail_chrom.encode('utf-8', 'surrogateescape') if tail_chrom is not None else None
        a.label_meth, a.label_unmeth = label_meth.encode(), label_unmeth.encode()
        a.submodel_of_char = _ptr(self._soc)
        self._args = a

    def rows(self, first, n_threads=0):
        """-> (rows as a bytes-like object, number of rows, stop): stop < n means record `stop` needs the host's own handling.
        The rows stay in the buffer the library made (a one-base motif writes a gigabyte of them per 10^8 events: no copy into a
        bytes object here, none when they are joined -- whoever writes them hands the buffer to write())."""
        text, nb, nr, stop = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().mc_format_diffs(C.byref(self._args), int(first), int(n_threads), C.byref(text), C.byref(nb),
                                    C.byref(nr), C.byref(stop)))
        return LibBuffer(text, nb.value), nr.value, stop.value


class LibBuffer(object):
    """Bytes the library allocated (mc_free releases them when this object goes): `view` is a memoryview over them."""

    def __init__(self, ptr, n):
        self._ptr, self.n = ptr, int(n)
        self.view = memoryview((C.c_char * self.n).from_address(ptr.value)).cast('B') if self.n else memoryview(b'')

    def __len__(self):
        return self.n

    def __bytes__(self):
        return self.view.tobytes()

    def __del__(self):
        try:
            if self._ptr is not None and self._ptr.value:
                self.view.release()
                lib().mc_free(self._ptr)
        except Exception:       # noqa (interpreter shutdown)
            pass
        self._ptr = None


class RowText(object):
    """The rows of a pass as the device wrote them (mc_last_row_text): `view` is a memoryview over the pinned block, `n_rows` the
    rows in it; release() (or the end of this object) gives the block back to the context -- the text must have been written by then."""

    def __init__(self, device, ptr, n_bytes, n_rows, block):
        self._dev, self._block = device, int(block)        # (the Device, not its context: a block outlives neither)
        self.n, self.n_rows = int(n_bytes), int(n_rows)
        self.view = memoryview((C.c_char * self.n).from_address(ptr)).cast('B') if self.n else memoryview(b'')

    def __len__(self):
        return self.n

    def __bytes__(self):
        return self.view.tobytes()

    def release(self):
        if self._block >= 0:
            block, self._block = self._block, -1
            self.view.release()
            self.view = memoryview(b'')
            self.n = 0
            self._dev._row_text_release(block)

    def __del__(self):
        try:
            self.release()
        except Exception:       # noqa (interpreter shutdown)
            pass


def repr_double(x):
    buf = C.create_string_buffer(40)
    lib().mc_repr_double(float(x), buf)
    return buf.value.decode()


def repr_double_rowtext(x):
    """repr(x) by the device row writer's digit generation, built for the host (None: a double it does not print)."""
    buf = C.create_string_buffer(40)
    n = lib().mc_repr_double_rowtext(float(x), buf)
    return buf.value.decode() if n >= 0 else None


def repr_fixed4(d):
    buf = C.create_string_buffer(40)
    lib().mc_repr_fixed4(int(d), buf)
    return buf.value.decode()


class Records(object):
    """Host buffers for flush records (mc_calls_view)."""

    def __init__(self, capacity, k):
        self.k = k
        self.capacity = int(capacity)
        c = max(1, self.capacity)
        self.feats = np.zeros(c * k, dtype=np.float64)
        self.site_pos = np.zeros(c, dtype=np.int32)
        self.site_seg = np.zeros(c, dtype=np.int32)
        self.close_row = np.zeros(c, dtype=np.int64)
        self.info = np.zeros(c, dtype=np.uint32)
        self.prob = np.full(c, np.nan, dtype=np.float64)
        self.n = 0

    def view(self):
        v = CallsView()
        v.capacity = self.capacity
        v.site_pos, v.site_seg = _ptr(self.site_pos), _ptr(self.site_seg)
        v.info, v.prob = _ptr(self.info), _ptr(self.prob)
        if self._feats is None and self._packed is not None:           # slot means as mc_wait_records sent them
            lo, hi, wide = self._packed
            v.feats_lo32, v.feats_hi32, v.feats_wide, v.n_wide = _ptr(lo), _ptr(hi), _ptr(wide), len(hi)
        else:
            v.feats = _ptr(self.feats)
        if self._close_row is not None or self._close_row32 is None:
            v.close_row = _ptr(self.close_row)
        else:
            v.close_row32 = _ptr(self._close_row32)
        if self._compacted:
            v.compacted, v.n_call_rows = 1, int(self._n_calls)
            if self._call_row is not None:
                v.call_row = _ptr(self._call_row)
        return v

    @classmethod
    def from_view(cls, v, n, k, owner):
        """Zero-copy wrap of library-owned buffers (valid until the next call on the owning context)."""
        r = cls.__new__(cls)
        r.k, r.capacity, r.n, r._owner = k, n, n, owner
        if v.feats:
            r.feats = _from_ptr(v.feats, n * k, np.float64)
        r.site_pos = _from_ptr(v.site_pos, n, np.int32)
        r.site_seg = _from_ptr(v.site_seg, n, np.int32)
        if v.close_row:
            r._close_row = _from_ptr(v.close_row, n, np.int64)
        else:                                # mc_wait_records on tables below 2^31 - 1 rows: 32-bit closing rows
            r._close_row32 = _from_ptr(v.close_row32, n, np.int32)
        r.info = _from_ptr(v.info, n, np.uint32)
        if v.compacted:                      # mc_wait_records: means / probabilities of the calls only, compacted
            m = int(v.n_call_rows)
            r._compacted = True
            if v.call_row:
                r._call_row = _from_ptr(v.call_row, n, np.int32)
            r._n_calls = m
            if v.feats:
                r.feats = _from_ptr(v.feats, m * k, np.float64)
            else:                            # packed slot means (mc_calls_view.feats_lo32): unpacked on first use
                r._packed = (_from_ptr(v.feats_lo32, m * k, np.int32), _from_ptr(v.feats_hi32, int(v.n_wide), np.uint32),
                             _from_ptr(v.feats_wide, m, np.uint8))
            r.prob = _from_ptr(v.prob, m, np.float64)
        else:
            r.prob = _from_ptr(v.prob, n, np.float64)
        return r

    # closing rows, call rows, slot means: columns mc_wait_records does not send in full (mc_calls_view) are rebuilt on first use
    _close_row = _close_row32 = _call_row = _feats = _packed = None
    _compacted = False

    @property
    def feats(self):
        """Slot means, k per row of feats / prob (row-major, flat)."""
        if self._feats is None and self._packed is not None:
            out = np.empty(max(1, self._n_calls * self.k), dtype=np.float64)
            v = self.view()
            check(lib().mc_calls_expand(C.byref(v), int(self.n), int(self.k), None, None, _ptr(out)))
            self._feats = out[:self._n_calls * self.k]
        return self._feats

    @feats.setter
    def feats(self, a):
        self._feats = a

    @property
    def close_row(self):
        if self._close_row is None and self._close_row32 is not None:
            self._close_row = self._close_row32.astype(np.int64)
        return self._close_row

    @close_row.setter
    def close_row(self, a):
        self._close_row = a

    @property
    def call_row(self):
        """None: feats / prob have one row per record.  Else the row of every record in feats / prob (-1: MC_I_TOO_MANY)."""
        if self._call_row is None and self._compacted:
            keep = (self.info[:self.n] & I_TOO_MANY) == 0
            rows = np.cumsum(keep, dtype=np.int64) - 1
            rows[~keep] = -1
            self._call_row = rows.astype(np.int32)
        return self._call_row

    @call_row.setter
    def call_row(self, a):
        self._call_row = a
        self._compacted = a is not None

    def count(self, n, seg_read=None, pos_marks=None):
        """mc_count_records over records [0, n): -> ((too many skips, skips included, multiple M) or None, ascending, smallest site
        position of a call, largest + 1); pos_marks: uint8 array, the calls' positions below its length are marked."""
        v = self.view()
        counts = np.zeros(3, dtype=np.int64)
        asc, lo, top = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        if seg_read is not None:
            seg_read = np.ascontiguousarray(seg_read, dtype=np.int32)
        check(lib().mc_count_records(C.byref(v), int(n), _ptr(seg_read) if seg_read is not None else None,
                                     len(seg_read) if seg_read is not None else 0, _ptr(counts) if seg_read is not None else None,
                                     C.byref(asc), _ptr(pos_marks) if pos_marks is not None else None,
                                     len(pos_marks) if pos_marks is not None else 0, C.byref(lo), C.byref(top)))
        return (tuple(int(c) for c in counts) if seg_read is not None else None), bool(asc.value), int(lo.value), int(top.value)

    @property
    def n_calls(self):
        """Rows of feats / prob."""
        return self._n_calls if self._compacted else self.n

    def by_record(self):
        """A copy with one feats / prob row per record (zeros / NaN for the MC_I_TOO_MANY records of a compacted view)."""
        if self.call_row is None:
            return self
        n, k = self.n, self.k
        r = Records(n, k)
        r.n = n
        for name in ('site_pos', 'site_seg', 'close_row', 'info'):
            getattr(r, name)[:n] = getattr(self, name)[:n]
        kept = self.call_row[:n] >= 0
        rows = self.call_row[:n][kept]
        r.feats.reshape(-1, k)[:n][kept] = self.feats.reshape(-1, k)[rows]
        r.prob[:n][kept] = self.prob[rows]
        return r
