/*
 * mcaller_hip.h -- C ABI of libmcaller_hip.so, the MI355X-native replacement for the hot path of
 * al-mcintyre/mCaller:
 *
 *   extract_contexts.py::extract_features   (reference extract_contexts.py:110-303)
 *     - the per-row 6-slot window machine over nanopolish eventalign rows   (:147-291)
 *     - model[key].predict_proba([diffs]) per observation                    (:199)
 *
 * The reference has no FFI: the path sits behind the Python call boundary
 * `extract_features(tsv_input, fasta_input, read2qual, k, skip_thresh, qual_thresh, modelfile,
 * classifier, startline, endline, train, pos_label, base, motif, positions_list)`
 * (extract_contexts.py:110, called at mCaller.py:53,58,60).  A maintainer of the reference binds the
 * entry points below with ctypes; mcaller_amd/extract_contexts.py is that
 * binding, with the reference's signature.
 *
 * Conventions: every function returns 0 on success, <0 on error (text via mc_last_error(), thread
 * local; MC_E_NO_FREE_SLOT is the one code a caller acts on: hand out the oldest pass and try again).  Plain pointers and sizes only; no exceptions or callbacks cross the ABI.  Host buffers are
 * caller-allocated and never retained after the call returns.  One mc_ctx per GPU; calls on a ctx
 * are serialised by the caller (one process per GPU).
 */
#ifndef MCALLER_HIP_H
#define MCALLER_HIP_H

/* mc_ctx_upload_table_async / mc_ctx_parse_begin: every table slot is taken by a pass in flight (or by the records handed out
 * last).  Not a failure of the stream: mc_wait_records frees a slot. */
#define MC_E_NO_FREE_SLOT (-16)

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_MAX_K 8            /* -n/--num_variables supported: 1..8 (reference default 6, mCaller.py:131) */

/* ---- columnar event table (one entry per eventalign row the reference's loop would process) ---- */
/* flags column bits */
#define MC_F_KMER_EQ    0x01  /* reference_kmer (col 3) == model_kmer (col 10)      extract_contexts.py:169 */
#define MC_F_MODEL_N    0x02  /* model_kmer == "NNNNNN"                            extract_contexts.py:167 */
#define MC_F_SEG_START  0x04  /* first row of a (read name, contig) segment                          */
#define MC_F_NAME_START 0x08  /* first row of a name block (maximal run of rows with one read name)  */

typedef struct mc_table_view {
    int64_t n_rows;
    const int32_t *pos;        /* col 2: 0-based k-mer start                                  :175 */
    const int32_t *event_model_e4; /* [2*n_rows] per row: col 7 event_level_mean, col 11 model_mean, in units of 1e-4 pA,
                                      interleaved -- the parser writes the pair, and a window's events are one DRAM page :286.
                                      |value| < 10^9 (the kernels take event - model in 32 bits; both parsers refuse a row
                                      beyond that, or hand over the rounded difference and a model of 0) */
    const int32_t *event_idx;  /* col 6                                                   :162,169 */
    const uint8_t *flags;      /* MC_F_*                                                           */
    int32_t n_seg;
    const int64_t *seg_row_begin; /* [n_seg+1] */
    const int32_t *seg_read;      /* [n_seg] read id: equal names <=> equal ids               :161 */
    const int32_t *seg_contig;    /* [n_seg] contig id (index into the reference set)         :154 */
    int32_t n_reads;
} mc_table_view;

/* ---- marked reference: what find_and_methylate builds (extract_contexts.py:60-81), as bitmasks ---- */
typedef struct mc_ref_view {
    int32_t n_contigs;
    const int64_t *contig_len;    /* [n_contigs] */
    const int64_t *seq_off;       /* [n_contigs] byte offset of the contig in seq                    */
    const uint8_t *seq;           /* upper-cased bases, ASCII                                    :80 */
    const int64_t *word_off;      /* [n_contigs] u32-word offset of the contig in mbits_*            */
    const uint32_t *mbits_fwd;    /* bit p set <=> meth_fwd[p]=='M'; >= 2 zero words of padding  :62 */
    const uint32_t *mbits_rev;    /* bit p set <=> meth_rev[p]=='M'                              :63 */
    int64_t n_seq_bytes, n_words;
} mc_ref_view;

/* ---- flush records: one per window the machine closes (extract_contexts.py:179-239) ---- */
/* info bits */
#define MC_I_EMPTY_MASK   0x000000FFu /* bit i: feature i (final order) came from an empty slot -> literal 0 :186 */
#define MC_I_REV          0x00000100u /* strand '-'                                               :216 */
#define MC_I_TOO_MANY     0x00000200u /* num_skips > skip_thresh: counted, not emitted            :239 */
#define MC_I_MULTI        0x00000400u /* the closing row shifted the window with kmer[0]!='M'     :247 */
#define MC_I_EDGE         0x00000800u /* context slice leaves the contig: host decides (Python slicing) */
#define MC_I_NEXT_SHIFT   16          /* bits 16..23: context[k] (ASCII) -> sub-model key   :197 */

typedef struct mc_calls_view {
    int64_t capacity;
    double  *feats;      /* [capacity*k] slot means in the order the reference prints them   :186-188 */
    int32_t *site_pos;   /* mpos                                                                 :216 */
    int32_t *site_seg;   /* segment of the window's last site row -> read name, site contig      :216 */
    int64_t *close_row;  /* row that closed the window (its contig is the chrom column, R8); n_rows if
                            the closing row lies beyond this table (see tail_contig)             :216 */
    uint32_t *info;      /* MC_I_* */
    double  *prob;       /* p(m6A) from the MLP/forest; NaN where not scored                     :199 */
    /* compacted == 0: feats and prob hold one row per record (row j = record j).  compacted != 0 (mc_wait_records): they
     * hold n_call_rows rows, one per record WITHOUT MC_I_TOO_MANY, in record order (the reference only counts the others,
     * :239, nothing is stored for them): the row of record j is the number of records before j without MC_I_TOO_MANY.
     * call_row, if not NULL, holds that number for every record (-1 for a MC_I_TOO_MANY record); mc_wait_records leaves it
     * NULL -- the copy-out over PCIe bounds a pass, a column the host can derive is not sent (mc_calls_expand fills one). */
    int32_t *call_row;
    int64_t n_call_rows;
    /* mc_wait_records on a table of fewer than 2^31 - 1 rows: close_row is NULL and the closing rows are here. */
    int32_t *close_row32;
    int32_t compacted;
    /* mc_wait_records: feats is NULL and the slot means of call row r, slot s, are packed -- a slot mean that is
     * fl(d / 10^4) for a 32-bit integer d (every slot that holds one event is; the host divides again and gets the same
     * double) travels as d in feats_lo32[r*k + s]; one that is not has bit s of feats_wide[r] set, its low 32 bits in
     * feats_lo32[r*k + s] and its high 32 bits in feats_hi32[j], j = the number of wide slots before it in (row, slot)
     * order.  mc_format_diffs reads that layout; mc_calls_expand unpacks it. */
    const int32_t *feats_lo32;
    const uint32_t *feats_hi32;
    const uint8_t *feats_wide;
    int64_t n_wide;
} mc_calls_view;

const char *mc_last_error(void);
const char *mc_version(void);
/* Cores this process may run on (sched_getaffinity) -- what the parser, the FASTQ reader and the row formatter size their
 * thread counts by when n_threads <= 0: a worker bound to its GPU's NUMA node starts one thread per core of that node. */
int mc_host_cores(void);
/* The marking of a contig for a motif (extract_contexts.py:33-41,:60-73) without the interpreter lock: upper_out = seq
 * upper-cased (ASCII), fwd_out / rev_out = upper_out with every occurrence of motif_fwd / motif_rev (left to right,
 * non-overlapping, like str.replace) replaced by repl_fwd / repl_rev of the same length; all buffers n bytes. */
int mc_mark_motifs(const char *seq, int64_t n, const char *motif_fwd, const char *repl_fwd, int32_t m_fwd,
                   const char *motif_rev, const char *repl_rev, int32_t m_rev, char *upper_out, char *fwd_out, char *rev_out);

/* ===== pinned host memory, recycled =====
 * The DMA engines read a table at PCIe speed, beside running kernels, only from pinned memory.  Blocks handed back with
 * mc_host_free are kept (up to keep_bytes of idle memory) and handed out again: pinning costs page-table work per page, a
 * stream of shards allocates during its first two or three shards only.  With parser_uses_pool != 0 the columns of every
 * table parsed from then on (mc_parse_eventalign*) live in such blocks.  Without a GPU the blocks are plain memory. */
void *mc_host_alloc(int64_t bytes);
void mc_host_free(void *p);
int mc_host_is_pinned(const void *p);
int mc_host_pool_config(int32_t parser_uses_pool, int64_t keep_bytes);   /* keep_bytes < 0: unchanged (default 8 GiB) */

/* ===== native eventalign parser (host), replaces the line.split() ingest extract_contexts.py:140-152 ===== */
typedef struct mc_parsed mc_parsed;
/* Parses the byte range the reference's loop would consume for (startline, endline):
 * seek(max(startline-500,0)), readlines(8000000) batches while linepos <= endline-500 (:141-146).
 * Rows with < 12 tokens are dropped (:149-152); rows whose contig is not in contig_names are dropped
 * and recorded (the "could not find sequence" path :156-160).  The range is cut at line starts into one piece per
 * thread (n_threads > 0: exactly that many pieces; <= 0: one per core, pieces of at least 4 MB) and stitched in file order. */
int mc_parse_eventalign(const char *path, int64_t startline, int64_t endline,
                        const char *const *contig_names, int32_t n_contigs, int32_t n_threads,
                        mc_parsed **out);
/* The same for an exact byte range [byte_begin, byte_end) that starts at a line start (a piece of a file cut for several
 * GPUs), and the cutter: n_parts+1 offsets, every cut at the first line of a read (column 4 changes), pieces of similar
 * size -- a window never spans two reads (:179,:242), so pieces are scanned independently. */
int mc_parse_eventalign_range(const char *path, int64_t byte_begin, int64_t byte_end,
                              const char *const *contig_names, int32_t n_contigs, int32_t n_threads, mc_parsed **out);
int mc_eventalign_read_cuts(const char *path, int32_t n_parts, int64_t *cuts);
/* ... the same inside the byte range [lo, hi) (lo at a line start): cuts[0] = lo, cuts[n_parts] = min(hi, file size); and
 * the byte range the reference's batch loop consumes for (startline, endline) -- what mc_parse_eventalign parses (:141-148:
 * the last < 500 bytes of a file can stay unread) -- for callers that cut that range into shards. */
int mc_eventalign_read_cuts_range(const char *path, int64_t lo, int64_t hi, int32_t n_parts, int64_t *cuts);
/* ... at offsets of the caller's choosing (ascending, inside [lo, hi)): cuts[0] = lo, cuts[i + 1] = the first line of a read at or
 * behind want[i], cuts[n_want + 1] = min(hi, file size) -- a stream whose first shards are small (its pipeline fills sooner). */
int mc_eventalign_read_cuts_at(const char *path, int64_t lo, int64_t hi, const int64_t *want, int32_t n_want, int64_t *cuts);
int mc_eventalign_consumed_range(const char *path, int64_t startline, int64_t endline, int64_t *lo, int64_t *hi);
int mc_parsed_view(const mc_parsed *p, mc_table_view *out);
const char *mc_parsed_read_name(const mc_parsed *p, int32_t read_id);
int64_t mc_parsed_n_unknown(const mc_parsed *p);                 /* rows dropped for an unknown contig */
const char *mc_parsed_unknown_name(const mc_parsed *p, int64_t i); /* contig text of the i-th such row */
int32_t mc_parsed_n_pieces(const mc_parsed *p);                  /* pieces the range was cut into, one thread each */
void mc_parsed_free(mc_parsed *p);

/* ===== FASTQ read quality (replaces read_qual.py:6-19) =====
 * One (key, mean phred) pair per record, in file order: key = id.split(':')[0].split('_')[0] (:11-12), mean = exact
 * integer sum of (ASCII - 33) / count in float64 (np.mean of the phred list, :11).  A path containing ".gz" is inflated
 * (:7).  n_threads <= 0: one piece per core (pieces of at least 4 MB).  Errors (a title not starting with '@', a third
 * line not starting with '+', sequence and quality of different lengths) are the ValueErrors Biopython raises there. */
typedef struct mc_fastq mc_fastq;
int mc_fastq_read_quality(const char *path, int32_t n_threads, mc_fastq **out);
/* -> number of records; key_pool holds the keys, each followed by '\n'; key_off has n+1 offsets into it. */
int64_t mc_fastq_view(const mc_fastq *f, const char **key_pool, const int64_t **key_off, const double **mean);
void mc_fastq_free(mc_fastq *f);

/* ===== device context ===== */
typedef struct mc_ctx mc_ctx;
int mc_ctx_create(int device, mc_ctx **out);
/* One process per GPU on a multi-socket host: binds the calling thread (and the threads it starts later) to the cores of the
 * NUMA node the GPU is attached to, so that the pinned buffers of its contexts are allocated there.  Call before
 * mc_ctx_create.  -> node, or -1 if the topology is unknown (nothing changed).  Not an error code. */
int mc_bind_to_device_numa_node(int device);
void mc_ctx_destroy(mc_ctx *ctx);
int mc_ctx_set_reference(mc_ctx *ctx, const mc_ref_view *host_ref);          /* H2D, replaces :154-160 */
/* The same from the raw bases, the site masks made on the GPU (the FASTA site-LUT builder, extract_contexts.py:33-41,:60-81,
 * for a motif): host_ref->seq holds the FASTA bytes of EVERY contig (any case; ASCII), mbits_* are not read; motif_fwd /
 * repl_fwd: the motif and what str.replace puts in its place ('M' for the base), *_rev: for the reverse complement and the
 * complement base; 1..16 bases; the motifs must not be able to overlap themselves (no proper prefix is also a suffix -- then
 * "left to right, non-overlapping" is "every occurrence").  Leaves the device reference exactly as mc_ctx_set_reference
 * would with every contig marked (mc_ctx_fetch_reference copies it back: the parity tests). */
int mc_ctx_set_reference_motif(mc_ctx *ctx, const mc_ref_view *host_ref, const char *motif_fwd, const char *repl_fwd, int32_t m_fwd,
                               const char *motif_rev, const char *repl_rev, int32_t m_rev);
int mc_ctx_fetch_reference(mc_ctx *ctx, uint8_t *seq, int64_t n_seq_bytes, uint32_t *mbits_fwd, uint32_t *mbits_rev,
                           int32_t *rank_fwd, int32_t *rank_rev, int64_t n_words, int64_t *site_base, int64_t *n_sites);
int mc_ctx_upload_table(mc_ctx *ctx, const mc_table_view *host_table);        /* H2D of the columns; returns when done */
int mc_ctx_set_read_quality(mc_ctx *ctx, const double *qual, int32_t n_reads);/* read2qual, :163-166   */

/* ---- streaming distinct tables: a file as a sequence of shards (the reference's batch loop, :140-148) ----
 * A ctx holds MC_TABLE_SLOTS resident tables.  mc_ctx_upload_table_async ENQUEUES the upload of a table into a free slot
 * and makes it the current table (the one the passes enqueued afterwards scan): the columns travel on an upload stream of
 * their own (DMA, beside the kernels of earlier passes; the source should be pinned: mc_host_alloc / a parser table with
 * mc_host_pool_config(1, ..)); no kernel runs at upload -- the FIRST pass over a table validates it while it scans (are the
 * positions of every read non-decreasing, its event indices strictly monotone: what lets the window rule stand in for the
 * reference's sequential machine, :161-176), later passes over the same table use what it found.  read_qual
 * ([n_reads], read ids of THIS table; may be NULL: mc_ctx_set_read_quality applies) travels with the table.  No
 * hipMalloc / hipFree happens per table once the slots are big enough: mc_ctx_reserve_tables sizes them (and the per-pass
 * scratch and record sets) once for tables of up to max_rows rows, max_segs segments, max_reads reads; without it the
 * first table that needs more re-allocates (synchronising).  The host buffers must stay untouched until
 * mc_ctx_wait_upload(slot) returns.  (A table can also arrive as TEXT and be parsed on the device: mc_ctx_parse_begin below.)
 * A slot is free again when every pass that scanned its table has been handed out by
 * mc_wait_records (and the next pass after it: the last records handed out may still be reduced by mc_site_counts); with
 * no free slot the call fails with MC_E_NO_FREE_SLOT: wait for a pass first. */
#define MC_TABLE_SLOTS 12
int mc_ctx_reserve_tables(mc_ctx *ctx, int64_t max_rows, int32_t max_segs, int32_t max_reads);
int mc_ctx_upload_table_async(mc_ctx *ctx, const mc_table_view *host_table, const double *read_qual, int32_t *slot);
int mc_ctx_wait_upload(mc_ctx *ctx, int32_t slot);
int mc_ctx_current_slot(mc_ctx *ctx);                       /* slot of the current table, -1: none */
/* Makes the table resident in `slot` the current one again.  as_new != 0: the next pass treats it like a table it has never
 * seen -- it does everything the first pass over a table does (positions and event indices streamed, every row validated),
 * whatever earlier passes learned (measurement: the cost of a table that is scanned once, without the upload).  Fails for a
 * slot that holds no COMPLETE table (never filled; a parse abandoned or never finished).  Passes over the slot that are in
 * flight keep the plan they were enqueued with: declare a table new beside them only if they are first passes too. */
int mc_ctx_select_table(mc_ctx *ctx, int32_t slot, int32_t as_new);
/* hipEvent time of the last upload into `slot` (waits for it): the H2D transfers, in ms.  *validate_ms is 0: nothing runs at
 * upload any more (the validation is part of the first pass's scan). */
int mc_ctx_upload_times_ms(mc_ctx *ctx, int32_t slot, float *h2d_ms, float *validate_ms);
/* hipEvent times of the last mc_ctx_parse_begin into `slot` (waits for it; call between mc_ctx_parse_begin and
 * mc_ctx_parse_finish / _abandon): *text_h2d_ms the transfer of the shard's text, *parse_ms the device parser's kernels behind
 * it (kp_count .. kp_place and the small copies that hand the result out) -- what replaces the reference's `line.split()` loop
 * (extract_contexts.py:146-152).  Meaningful when nothing else was in flight on the parse stream (bench.py's roofline_parser). */
int mc_ctx_parse_times_ms(mc_ctx *ctx, int32_t slot, float *text_h2d_ms, float *parse_ms);
/* MLP weights, row-major float64: W1[n_in*n_hidden], b1[n_hidden], W2[n_hidden], b2[1] per sub-model (at most 8 sub-models:
 * the reference's models have two, 'MG' and 'MH', or one); submodel_of_char[256]: context[k] (ASCII) -> sub-model index,
 * 255 = KeyError path (:197,:218). */
int mc_ctx_set_mlp(mc_ctx *ctx, int32_t n_models, int32_t n_in, int32_t n_hidden,
                   const double *W1, const double *b1, const double *W2, const double *b2,
                   const uint8_t *submodel_of_char);

/* Random forest (classifier RF, train_model.py:39-45; same call site :199): trees flattened, node arrays concatenated.
 * model_tree_off[n_models+1]: first tree of each sub-model; tree_node_off[n_trees+1]: root of each tree; left/right: child
 * node (absolute index) or -1 at a leaf; value[2*node .. 2*node+1]: the node's two class values.  Inputs are cast to
 * float32 before the comparisons `x[feature] <= threshold`, as scikit-learn does; p = mean over trees of v1/(v0+v1). */
int mc_ctx_set_forest(mc_ctx *ctx, int32_t n_models, int32_t n_in, const int32_t *model_tree_off,
                      const int32_t *tree_node_off, const int32_t *left, const int32_t *right, const int32_t *feature,
                      const double *threshold, const double *value, const uint8_t *submodel_of_char);

/* The closed-form classifiers of `-c LR` / `-c NBC` (train_model.py:55-60: LogisticRegression, GaussianNB; scored at the same
 * call site, :199).  params: n_models * stride doubles --
 *   MC_CLF_LOGISTIC, stride n_in + 1:   coef_[0][0..n_in), intercept_[0]              p = expit(x . coef + intercept)
 *   MC_CLF_GNB,      stride 4 n_in + 2: theta_[0], var_[0], theta_[1], var_[1] (n_in each), log(class_prior_[0]), log(class_prior_[1])
 *                                       p = exp(jll_1 - logsumexp(jll)), jll_c = log prior_c - 0.5 sum log(2 pi var_c) - 0.5 sum (x - theta_c)^2 / var_c
 * fp64, sums in index order.  Replaces the MLP / forest of the context. */
#define MC_CLF_LOGISTIC 1
#define MC_CLF_GNB 2
int mc_ctx_set_simple_classifier(mc_ctx *ctx, int32_t kind, int32_t n_models, int32_t n_in, const double *params, int32_t stride,
                                 const uint8_t *submodel_of_char);

typedef struct mc_params {
    int32_t k;             /* -n   (:110 `k`)            */
    int32_t skip_thresh;   /* -s   (:183,242)            */
    double  qual_thresh;   /* -q   (:167)                */
    int32_t tail_contig;   /* contig id of the first unfiltered row AFTER this table (next shard), or -1
                              if none: the last window is then lost, as at EOF (R6)                  */
    int32_t score;         /* 1: run the classifier (predict mode); 0: features only (--train)     */
    int32_t entry_read;    /* read id `last_read` holds when the table starts (-1: none), and      */
    int32_t entry_first_idx; /* the matching first_read_ind (:161-162); shards > 0 of one file      */
} mc_params;

/* How the pass handed out last by mc_wait_records ran: *fused_room > 0 -- scan, ordering and emit as ONE kernel with that many
 * record slots per 1024-row piece (dense references: k1_fused; the slots a piece does not fill never reach the host);
 * *rerun != 0 -- the pipelined pass could not be finished as enqueued (record room too small, an irregular read, a row that
 * contradicts what a block was classified on) and was repeated synchronously inside mc_wait_records. */
int mc_last_pass_info(mc_ctx *ctx, int32_t *fused_room, int32_t *rerun);
/* The rows of a pipelined pass as TEXT, made on the device (mc_rowtext.hip; replaces mc_format_diffs on the host, the row writer of
 * extract_contexts.py:207-216).  mc_ctx_row_text(on = 1, the two labels of :200-206; on = 2 at the start of a stream: also takes back the
 * blocks an earlier stream never released -- nobody may be reading them any more): the passes enqueued from now on also print their
 * records where they are -- the packed records, the read names in the shard's text (tables the device parser made:
 * mc_ctx_parse_begin .. _finish), the contig names, the marked reference, shortest round-trip digits in integer arithmetic
 * (mc_rowtext.h) -- and send the text to pinned host memory behind the records.  After mc_wait_records, mc_last_row_text says
 * where: *text / *n_bytes / *n_rows, and *block >= 0 -- a handle (the block and the ticket it was taken with: one that is given back twice, or
 * after a later stream began, frees nothing) --, which the caller gives back with mc_row_text_release (any thread) once it
 * has written the rows; *block < 0: this pass has none (not asked for, another kind of table, no free block, or a record the
 * device does not print -- a context that leaves the contig, an unknown sub-model key, an unscored record, a number outside
 * [1e-29, 1e9): the reference's exit paths and the host formatter's general cases) and mc_format_diffs makes the rows as before. */
int mc_ctx_row_text(mc_ctx *ctx, int32_t on, const char *label_meth, const char *label_unmeth);
int mc_last_row_text(mc_ctx *ctx, const char **text, int64_t *n_bytes, int64_t *n_rows, int32_t *block);
int mc_row_text_release(mc_ctx *ctx, int32_t block);
/* The hot path on the GPU: strand resolve + window scan + classifier.  Leaves the flush records on the
 * device, in file order; *n_records = how many. */
int mc_extract_features(mc_ctx *ctx, const mc_params *prm, int64_t *n_records);
/* The records of the last mc_extract_features, copied into caller buffers (capacity >= n_records) ... */
int mc_fetch_records(mc_ctx *ctx, const mc_calls_view *host_out);
/* ... or as a view of the context's own pinned host buffers (no copy; valid until the next call on this ctx;
 * capacity is set to the record count).  mc_extract_features already moved them there, overlapped with the classifier. */
int mc_fetch_records_view(mc_ctx *ctx, mc_calls_view *out);
/* Pipelined passes, for callers that stream many tables / shards (or the same table, as bench.py does): a pass is only
 * ENQUEUED -- strand resolve, scan + emit, classifier, packing of what is copied out, all on the ctx stream, each
 * pass with its own counters, strand-resolve output and record set -- and copied out (one DMA transfer: the narrow columns
 * of the n records, then slot means and probabilities of the records that are calls, see mc_calls_view.call_row) when it
 * is waited for, beside the kernels of the passes behind it; no host round trip sits inside a pass.  At
 * most six passes are in flight (one being copied out, one in the side stream's kernels, one computing, the others queued).  mc_wait_records hands out the
 * OLDEST pass and returns a view of the context's pinned buffers (valid until six more passes have been enqueued).  A
 * pass that needs more than the fast path (irregular reads, record buffers too small) is re-run synchronously inside
 * mc_wait_records -- results are the same, only slower.  Either classifier (MLP: k2_mlp, forest: k3_forest) runs on the side
 * stream behind the pass's emit. */
int mc_extract_features_async(mc_ctx *ctx, const mc_params *prm);
int mc_wait_records(mc_ctx *ctx, int64_t *n_records, mc_calls_view *out);
/* Optional first half of mc_wait_records, for the oldest pass in flight whose copy-out has not been started: waits for
 * its kernels, reads its counters and ENQUEUES its copy-out without waiting for it.  Calling it for pass i+1 before
 * mc_wait_records(pass i) keeps the DMA engine busy back to back (bench.py does).  No-op if every pass is being copied. */
int mc_wait_records_begin(mc_ctx *ctx);
/* Kernel times of the last mc_extract_features, from hipEvents on the ctx stream, in ms:
 * [0] strand resolve (K0), [1] window scan (k1_scan), [2] window emit (k1_group_scan + k1_list + k1_emit),
 * [3] classifier (K2), [4] total. */
int mc_last_times_ms(mc_ctx *ctx, float *out5);
/* Pipelined passes: the two hipEvents that only TIME a pass (start of K0, start of the scan) are recorded with every n-th
 * pass (default 1 = every pass, 0 = never).  An event between two kernels costs the queue about 9 us on this GPU, 6 % of a
 * pass of the headline workload.  mc_last_pass_timed: 1 if the pass mc_wait_records handed out last carried them, i.e.
 * mc_last_times_ms is about that pass; for an untimed pass mc_last_times_ms keeps the values of the last timed one. */
int mc_ctx_set_pass_timing(mc_ctx *ctx, int every_n);
int mc_last_pass_timed(mc_ctx *ctx);
int mc_ctx_sync(mc_ctx *ctx);

/* Batched classifier alone (B2, extract_contexts.py:199): X[n*n_in] -> p[n]; host buffers. */
int mc_mlp_forward(mc_ctx *ctx, const double *X, const uint8_t *submodel, int64_t n, double *p);
int mc_forest_forward(mc_ctx *ctx, const double *X, const uint8_t *submodel, int64_t n, double *p);
int mc_simple_forward(mc_ctx *ctx, const double *X, const uint8_t *submodel, int64_t n, double *p);     /* LR / NBC */

/* ===== per-site reduction feeding make_bed (make_bed.py:86-96,:134,:143,:154), the one exchange step of a multi-GPU job =====
 * Sites = every 'M' of the marked strands, numbered per contig: '+' sites by position, then '-' sites by position
 * (mcaller_amd.make_bed.SiteIndex uses the same order).  Each rank reduces the records of its own last
 * mc_extract_features call on the device; mc_site_allreduce sums the counts and takes the minimum first-seen row over
 * the ranks with ncclAllReduce (RCCL over xGMI; librccl.so is loaded on first use) and copies the result out. */
#define MC_UNIQUE_ID_BYTES 128
int64_t mc_site_count(mc_ctx *ctx);                       /* number of marked sites of the reference in the ctx */
/* counts of this rank: n_meth / n_total per site (label 'm..' <=> p >= 0.5, :200) and the smallest close_row +
 * row_offset (global row of the first occurrence, :134).  Two kinds of records are left out and counted for the caller:
 * *n_pending -- probability NaN on the device (the host scored them): add them with mc_site_counts_add;
 * *n_cross_contig -- closed by a row of ANOTHER contig than the site's (tail_contig: the contig of the row after this
 * table, as in mc_params): make_bed keys such a row on the closing row's contig (R8, :216), which is no site of the
 * numbering -- the caller adds them to the BED as rows of their own. */
int mc_site_counts(mc_ctx *ctx, int64_t row_offset, int32_t tail_contig, int64_t *n_pending, int64_t *n_cross_contig);
/* The same over the passes of a streamed file: mc_site_counts_reset zeroes the counts, mc_site_counts_accumulate adds the
 * records of the pass handed out last (mc_wait_records) -- row_offset: the rows of the shards before it, so that the first-seen
 * rows stay in file order.  mc_site_counts == reset + accumulate. */
int mc_site_counts_reset(mc_ctx *ctx);
int mc_site_counts_accumulate(mc_ctx *ctx, int64_t row_offset, int32_t tail_contig, int64_t *n_pending, int64_t *n_cross_contig);
int mc_site_counts_add(mc_ctx *ctx, const int64_t *site, const uint8_t *is_meth, const int64_t *first_row, int64_t n);
int mc_comm_available(void);                              /* 0: librccl.so can be loaded in this process (nothing else is touched) */
int mc_comm_unique_id(uint8_t *out128);                   /* rank 0: ncclGetUniqueId; ship the bytes to every rank */
int mc_comm_init(mc_ctx *ctx, int32_t world, int32_t rank, const uint8_t *unique_id128);   /* ncclCommInitRank */
int mc_comm_destroy(mc_ctx *ctx);
/* all-reduce (if a communicator with world > 1 is set) + D2H: n_meth[n], n_total[n] int32, first_row[n] int64
 * (INT64_MAX: site not seen); *ms = time of the two collectives (hipEvents). */
int mc_site_allreduce(mc_ctx *ctx, int32_t *n_meth, int32_t *n_total, int64_t *first_row, float *ms);
/* This rank's own counts as they stand (D2H only, no collective): what a job adds up on the host when a rank could not
 * take part in the all-reduce. */
int mc_site_counts_fetch(mc_ctx *ctx, int32_t *n_meth, int32_t *n_total, int64_t *first_row);

/* ===== the classifier fit behind --train (train_model.py:47,:62-65,:81-100): one-hidden-layer tanh/logistic perceptron, =====
 * Adam, binary log-loss + L2, scikit-learn's MLPClassifier recipe (batches of min(200, n) rows, tol / n_iter_no_change
 * stopping).  n_jobs independent fits run side by side, one workgroup each (the 5 GroupKFold fits + the final fit of a
 * sub-model); job j trains on rows train_idx[train_off[j] .. train_off[j+1]) of X and is scored (accuracy, p > 0.5) on
 * rows val_idx[val_off[j] .. val_off[j+1]).  Start weights: `init` ([n_jobs][n_in*n_hidden + 2*n_hidden + 1]: W1
 * row-major, b1, W2, b2) or, if NULL, Glorot-uniform from seeds[j] (NULL: seed + j).  The epoch order is a keyed
 * permutation of the row index (shuffle != 0) or the given order. */
typedef struct mc_fit_params {
    int32_t n_in, n_hidden;        /* k+1 inputs (<= MC_MAX_K+1), hidden units (<= 128; reference: 100)  */
    int32_t batch_size;            /* scikit-learn 'auto': min(200, n); the kernel clamps to each job's n */
    int32_t max_iter;              /* 200 */
    int32_t n_iter_no_change;      /* 10  */
    int32_t shuffle;               /* 1   */
    double alpha, lr_init, beta1, beta2, epsilon, tol;   /* 0.001 (train_model.py:47), 0.001, 0.9, 0.999, 1e-8, 1e-4 */
    uint64_t seed;
} mc_fit_params;
int mc_mlp_fit(mc_ctx *ctx, const mc_fit_params *prm, const double *X, const uint8_t *y, int64_t n_samples, int32_t n_jobs,
               const int64_t *train_off, const int32_t *train_idx, const int64_t *val_off, const int32_t *val_idx,
               const uint64_t *seeds, const double *init,
               double *W1, double *b1, double *W2, double *b2,      /* per job: [n_in*n_hidden], [n_hidden], [n_hidden], [1] */
               double *loss_curve /* [n_jobs*max_iter] */, int32_t *n_iter /* epochs run */, int64_t *val_correct);

/* ===== the eventalign text parsed on the GPU (replaces the row ingest, extract_contexts.py:140-152, for streamed shards) =====
 * The host only moves bytes: mc_read_file_range preads a byte range into (pinned) memory with all cores;
 * mc_ctx_parse_begin sends it and enqueues the kernels that split it into lines, tokenise (str.split()'s ASCII whitespace,
 * first 12 tokens), convert (int(); floats in plain decimal form as units of 1e-4) and write the columns of a free table
 * slot; mc_ctx_parse_end waits and hands out what the host needs for names: segments (first row, contig id, where the read
 * name stands in the text, whether it starts a name block), the contig tokens the reference does not hold (file order) and
 * the flag column; mc_ctx_parse_finish takes read ids and qualities: the slot is now what mc_ctx_upload_table_async would
 * have left (current table).  status 1: the shard holds something this path does not reproduce bit for bit (a number form
 * that needs strtod, a value out of range, more rows or segments than the slot holds; mc_last_error says which) -- call
 * mc_ctx_parse_abandon and parse it with mc_parse_eventalign_range.  begin for the next shard may be called before end. */
typedef struct mc_devparse_result {
    int32_t status;
    int64_t n_lines, n_rows;
    int32_t n_seg, n_unknown;
    const int64_t *seg_row_begin;     /* [n_seg] ascending                                              */
    const int32_t *seg_contig;        /* [n_seg]                                                        */
    const int64_t *seg_name_off;      /* [n_seg] offset of the segment's read name in the text          */
    const int32_t *seg_name_len;
    const uint8_t *seg_name_start;    /* [n_seg] 1: first segment of a name block (MC_F_NAME_START)     */
    const int64_t *unknown_off;       /* [n_unknown] contig tokens of skipped lines (:156-160)          */
    const int32_t *unknown_len;
    const uint8_t *flags;             /* [n_rows] host copy of the flag column (pinned, owned by ctx)   */
} mc_devparse_result;
int mc_read_file_range(const char *path, int64_t byte_begin, int64_t byte_end, char *dst, int32_t n_threads);
/* ... or WITHOUT the copy: the range mapped from the page cache and registered for the DMA engines (a streamed shard's text goes
 * to the GPU from where the kernel keeps it).  *ptr points at byte `lo`; valid until mc_unmap_file_range(*handle).  Returns
 * non-zero where the runtime cannot register the mapping (no GPU ...): read the range with mc_read_file_range then. */
int mc_map_file_range(const char *path, int64_t lo, int64_t hi, void **handle, const char **ptr);
void mc_unmap_file_range(void *handle);
int mc_ctx_parse_begin(mc_ctx *ctx, const char *text, int64_t n_bytes, const char *const *contig_names, int32_t n_contigs,
                       int64_t max_rows, int32_t *slot);
int mc_ctx_parse_end(mc_ctx *ctx, int32_t slot, mc_devparse_result *out);
int mc_ctx_parse_finish(mc_ctx *ctx, int32_t slot, const int32_t *seg_read, int32_t n_reads, const double *read_qual);
int mc_ctx_parse_abandon(mc_ctx *ctx, int32_t slot);
/* The columns of the table in `slot` copied back to the host (any may be NULL) -- what the parity tests compare. */
int mc_ctx_fetch_columns(mc_ctx *ctx, int32_t slot, int64_t n_rows, int32_t *pos, int32_t *event_model_e4, int32_t *event_idx,
                         uint8_t *flags);

/* ===== native `.diffs.<k>` row formatter (host), replaces the text assembly of the flush, extract_contexts.py:186-216 ===== */
typedef struct mc_format_args {
    const mc_calls_view *rec;          /* flush records in host memory (mc_fetch_records / _view)                  */
    int64_t n_records;
    int32_t k;
    const mc_table_view *table;        /* seg_row_begin / seg_read / seg_contig are read                           */
    const mc_ref_view *ref;            /* bases + strand masks: the marked strings the context is sliced from :194 */
    const char *const *contig_names;   /* [ref->n_contigs]                                                         */
    const char *const *read_names;     /* [table->n_reads] column 2 of the row                                :216 */
    const char *const *read_qual_txt;  /* [table->n_reads] str(read2qual[...]), the k+1-th feature        :189-193 */
    const char *tail_chrom;            /* chrom of records closed by the next shard's first row (or NULL)          */
    const char *label_meth;            /* 'm6A' / 'm'+base                                                :200-204 */
    const char *label_unmeth;          /* base                                                                :206 */
    const uint8_t *submodel_of_char;   /* [256] as in mc_ctx_set_mlp; 255 = the KeyError path             :218-223 */
} mc_format_args;
/* Rows of records [first, *stop_at) as one malloc'ed text block (release with mc_free); records flagged
 * MC_I_TOO_MANY produce no row.  *stop_at < n_records: that record needs the host's own handling (context leaving the
 * contig, NaN probability, unknown sub-model key or complement, centre not 'M': the reference's exit/crash paths). */
int mc_format_diffs(const mc_format_args *args, int64_t first, int32_t n_threads, char **text, int64_t *n_bytes,
                    int64_t *n_rows, int64_t *stop_at);
/* What records [0, n) add to the reference's counters, in one pass: counts3 = records with MC_I_TOO_MANY (:239), calls with an empty
 * slot (:234-238), records with MC_I_MULTI (:247-248) -- the SIZES of the reference's sets of (read, site) pairs if *ascending comes
 * back 1 (every pair new: a table whose read names do not repeat; seg_read[n_seg]: the read of every segment), otherwise the caller
 * counts distinct pairs itself -- and the positions of the calls (:235): pos_marks[p] = 1 for every call at site position p < n_marks,
 * *pos_min / *pos_top = smallest position / largest + 1 (a caller whose marks are too short grows them and calls again).  counts3 and
 * pos_marks may be NULL. */
int mc_count_records(const mc_calls_view *rec, int64_t n, const int32_t *seg_read, int64_t n_seg, int64_t *counts3, int32_t *ascending,
                     uint8_t *pos_marks, int64_t n_marks, int64_t *pos_min, int64_t *pos_top);
/* The columns mc_wait_records does not send, rebuilt on the host (all cores): call_row_out[n] (see mc_calls_view.call_row),
 * close_row_out[n] (64-bit closing rows), feats_out[n_call_rows * k] (the slot means as doubles); any may be NULL. */
int mc_calls_expand(const mc_calls_view *rec, int64_t n_records, int32_t k, int32_t *call_row_out, int64_t *close_row_out,
                    double *feats_out);
void mc_free(void *p);
/* repr(float) == str(np.float64) of one value into out32 (NUL-terminated); returns its length. */
int mc_repr_double(double v, char *out32);
/* ... of d / 1e4 for a 32-bit integer d, from the integer alone (how the formatter prints the slot means that travel as
 * integers, mc_calls_view.feats_lo32): the same characters as mc_repr_double((double)d / 1e4). */
int mc_repr_fixed4(int32_t d, char *out32);
/* repr(v) by the device row writer's digit generation (mc_rowtext.h), built for the host: the same characters as mc_repr_double for
 * 1e-29 <= |v| < 1e9 and for zero; -> length, -1: a double it does not print. */
int mc_repr_double_rowtext(double v, char *out32);

/* ===== measurement plumbing: a table as nanopolish-eventalign text (13 columns), written by all host cores =====
 * For file-to-file timing on synthetic workloads (bench.py); seq = the contig's bases (k-mers of columns 3 and 10). */
int mc_synth_write_tsv(const char *path, const mc_table_view *table, const char *seq, int64_t seq_len, const char *contig,
                       const char *const *read_names, int32_t n_threads, int64_t *n_bytes);

#ifdef __cplusplus
}
#endif
#endif
