// mc_stream.hip -- libmcaller_hip.so, the host side of the device code (gfx950 / MI355X): contexts, table slots, passes, the
// device parser (mc_devparse.inc), the per-site reduction and RCCL.  C ABI: include/mcaller_hip.h.  The kernels of the passes live
// in mc_k0.hip, mc_scan.hip, mc_emit.hip, mc_fused.hip, mc_literal.hip, mc_classify.hip, mc_rowtext.hip (shared structures: mc_dev.h; the
// row-by-row walk of one window: mc_rows.h; the digits of a printed double: mc_rowtext.h); their map:
//
// The reference's hot path (extract_contexts.py:147-291 + :199) as HIP kernels over a columnar event
// table resident in HBM:
//
//   text         kp_count / kp_scan / kp_starts / kp_parse / kp_count_rows / kp_place   the eventalign TEXT of a streamed shard
//                                parsed on the device (mc_devparse.inc): line starts, tokens, numbers, name blocks and segments,
//                                the columns written straight into a table slot
//   per table     (nothing runs at upload: the first pass over a table validates it while it scans)
//                (nb_template     the pass-independent fields of the name-block descriptors: inside the first k0_first_site over a table)
//                k_summarize     a table that is scanned a second time gets unit summaries (first / last position of every
//                                eight rows): every further scan reads 1 B/row instead of streaming the columns
//   per pass     k0_first_site   first site row of every name block under the "new read" strand rule (:161-174) -> strand of
//                                the block; classifies the block (regular / no sites / irregular) in the same workgroup.  On
//                                a table no pass has validated yet the classification rests on the block's first rows
//                                (direction of the event index, position 0) and the scan confirms it
//                k0_classify / k0_extend   tables with repeated read names; irregular runs widened
//                k1_scan         THE SCAN: one wave per tile of 2048 rows, nothing persistent.  First pass over a table: the
//                                position and event-index columns (8 B/row) go from HBM into registers and every row is
//                                compared with the row before it (positions non-decreasing? event index strictly monotone?
//                                -- what makes a name block "regular"); units of eight rows that can hold a site row are
//                                found with one extract from the strand bitmask (two words per unit, straight from L2) and
//                                listed in LDS, the listed units fetch their flag bytes, their rows are tested for "last row
//                                of a window"; every closed window leaves a 32-byte payload.  Later passes over the same
//                                table read the unit summaries instead (1 B/row); one-base motifs, where every unit passes,
//                                stream the positions.
//                k1_group_scan / k1_list   file order of the windows; payloads gathered into it
//                k1_emit         eight lanes per window: which of the rows before its last row belong to which slot, slot
//                                means in NumPy pairwise order (fp64) from the rows' (event, model) pairs -> one flush record
//                                (a pipelined pass: on the side stream, in front of the pass's classifier and packing)
//                k1_fused        a dense reference (a one-base motif), pipelined passes: scan, ordering and emit as one kernel,
//                                fixed room per 960-row piece (mc_fused.hip)
//                k2_mlp          batched 7-H-1 tanh/logistic forward: one lane per record, weights as scalar operands, a quarter
//                                of the hidden units per SIMD (flush records: hidden layer in fp32, fp64 where a printed digit
//                                could depend on it)
//                k2_mlp<.., PACK>   the side stream of a pipelined pass as ONE kernel: the windows the emit left to the
//                                row-by-row walk (its own records'), the MLP, the records packed for the copy-out
//                k1_rare_dev, k_pack_count / k_pack   the same in kernels of their own (other classifiers than the MLP)
//                k3_forest, k3_simple   random forest / LR / NBC predict_proba;  k_literal / k_merge  irregular reads, row by row
//                k_rt_count / k_rt_scan / k_rt_wide / k_rt_digits / k_rt_rows<false> / k_rt_scan_len / k_rt_rows<true> / k_rt_copy   the
//                                rows of a streamed shard as TEXT, written behind its packed records (mc_rowtext.hip; mc_ctx_row_text):
//                                the digits of every 64-bit slot mean one lane per number, the rows one lane per record -- counted,
//                                placed by a scan, written --, the text sent to a pinned block
//                k_site_counts   per-site reduction (+ ncclAllReduce)
//                k_copy_bytes    small transfers by the compute units (the DMA engines serialise behind queued text)
//
// Equivalence with the sequential machine on regular blocks (one contig, positions non-decreasing, event
// index monotone in the direction the first site row implies, no site at contig position 0, read name not
// seen before) is argued in DESIGN.md; every other block is classified irregular and handled by the
// literal per-run kernel (k_literal) so results never depend on a CPU path.
#include "mc_dev.h"
#include <atomic>

// ===================================================================================================
// host side
// ===================================================================================================
#ifndef MC_PASSES_IN_FLIGHT_N
#define MC_PASSES_IN_FLIGHT_N 6
#endif
// one being copied out, one in the side stream's kernels, one computing, two or three queued: a pass is 0.6-0.7 ms from its strand resolve
// to its records in host memory, and the ctx stream takes a new one every 0.21 ms (four in flight left it waiting for the host)
constexpr int MC_PASSES_IN_FLIGHT = MC_PASSES_IN_FLIGHT_N;
constexpr int MC_ROW_TEXT_BLOCKS = 6;    // pinned blocks the rows of text leave in (mc_rowtext.hip): a pass's stays taken until the host has written it
static_assert(MC_ROW_TEXT_BLOCKS <= 8, "a block's handle is ticket * 8 + index");

// What K0 writes and K1 reads, per pass in flight
struct K0Set {
    NbDesc *desc = nullptr;
    int64_t *nb_f0 = nullptr;
};

// One resident table.  A ctx owns MC_TABLE_SLOTS of them so that a file can go through the GPU as a sequence of shards:
// one being uploaded, one being scanned, the others waiting for their records to be handed out.  All device memory of a
// slot is allocated once (mc_ctx_reserve_tables, or by the first table that needs more) -- an upload is DMA transfers, no kernel
// (the first pass over the table validates it while it scans), no hipMalloc / hipFree.
struct SmallLayout {       // byte offsets of a table's small arrays inside one block: the same on the pinned host stage and on the device
    size_t seg_begin, seg_read, seg_contig, nb_row_begin, nb_seg_begin, nb_read, nb_repeat, nb_vflags, tile_nb, qual, total;
};
static SmallLayout small_layout(int64_t n_seg, int64_t n_tiles, int64_t n_reads) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    SmallLayout L;
    size_t o = 0;
    L.seg_begin = o;    o = al(o + (size_t)(n_seg + 1) * 8);
    L.seg_read = o;     o = al(o + (size_t)n_seg * 4);
    L.seg_contig = o;   o = al(o + (size_t)n_seg * 4);
    L.nb_row_begin = o; o = al(o + (size_t)(n_seg + 1) * 8);
    L.nb_seg_begin = o; o = al(o + (size_t)(n_seg + 1) * 4);
    L.nb_read = o;      o = al(o + (size_t)n_seg * 4);
    L.nb_repeat = o;    o = al(o + (size_t)n_seg);
    L.nb_vflags = o;    o = al(o + (size_t)(n_seg + 1) * 4);
    L.tile_nb = o;      o = al(o + (size_t)(n_tiles + 1) * 4);
    L.qual = o;         o = al(o + (size_t)n_reads * 8);
    L.total = o;
    return L;
}

#include "mc_devparse.inc"

// dst / src: device memory or pinned host memory (hipHostMalloc), both 16-byte aligned
static int copy_by_kernel(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    void *d = dst;
    const void *s = src;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, dst) == hipSuccess && a.type == hipMemoryTypeHost) HIP_TRY(hipHostGetDevicePointer(&d, dst, 0));
    else (void)hipGetLastError();
    if (hipPointerGetAttributes(&a, src) == hipSuccess && a.type == hipMemoryTypeHost) HIP_TRY(hipHostGetDevicePointer((void **)&s, const_cast<void *>(src), 0));
    else (void)hipGetLastError();
    const unsigned blocks = (unsigned)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(k_copy_bytes, dim3(blocks), dim3(256), 0, st, (unsigned char *)d, (const unsigned char *)s, bytes);
    return 0;
}
constexpr size_t COPY_BY_KERNEL_MAX = (size_t)4 << 20;      // larger transfers go to the DMA engines
constexpr size_t COPY_BY_KERNEL_MAX_STREAMING = (size_t)64 << 20;     // ... a pass's records while text is being streamed in: see mc_wait_records_begin

struct TableSlot {
    DevTable T;                        // the table in the slot (pointers into the slot's allocations)
    int64_t cap_rows = 0, cap_segs = 0, cap_reads = 0;
    int32_t *pos = nullptr, *idx = nullptr;
    int2 *evmu = nullptr;
    uint8_t *flags = nullptr;
    int2 *unit_pp = nullptr;
    NbDesc *nb_tmpl = nullptr;
    unsigned char *small_dev = nullptr, *stage = nullptr;   // the small arrays: device block, pinned host stage
    size_t small_cap = 0;
    double *qual = nullptr;            // read qualities that travelled with the table (in small_dev), or nullptr
    int32_t n_qual = 0;
    hipEvent_t ev_uploaded = nullptr;  // the H2D transfers of the slot's table are done
    hipEvent_t ev_up_start = nullptr, ev_val_start = nullptr, ev_valid = nullptr;   // ... begin; the small arrays are in place (ctx stream)
    int refs = 0;                      // passes in flight that scan this table (+1 while the device parser fills the slot)
    bool holds_table = false;          // S.T describes the columns in the slot (set by fill_slot; cleared when a parse begins to
                                       // overwrite them or is abandoned): what mc_ctx_select_table may make current again
    // What the passes enqueued so far leave behind for the next one (host-side notes; the work is ordered by the ctx stream):
    int passes = 0;                    // passes enqueued over this table.  The first streams positions and event indices and
                                       // completes the validation flags (k1_scan, SCAN_VALIDATE); later ones classify on the flags.
                                       // A table that comes back a third time (other parameters, a resident table) is worth
                                       // unit summaries (k_summarize): from then on a scan reads 1 B/row
    bool summarized = false;           // ... the summaries exist
    // the device parser (mc_ctx_parse_begin .. _finish)
    char *text = nullptr;              // the shard's text on the device
    int64_t cap_text = 0;
    KpHead *kp_head = nullptr, *kp_head_h = nullptr;         // device result block, pinned host copy
    KpSeg *kp_segs = nullptr, *kp_segs_h = nullptr;
    KpUnknown *kp_unknown = nullptr, *kp_unknown_h = nullptr;
    uint8_t *kp_flags_h = nullptr;     // pinned host copy of the flag column
    int64_t kp_cap_flags = 0;
    int kp_cap_segs = 0;
    hipEvent_t ev_parsed = nullptr, ev_text_up = nullptr;
    int kp_state = 0;                  // 0: idle, 1: parse enqueued, 2: results handed out (mc_ctx_parse_end)
    bool from_parser = false;          // the slot's table was made by the device parser: its text and segments (kp_segs_h, sorted) are the table's
    int64_t kp_bytes = 0, kp_flags_sent = 0;
    std::vector<int64_t> kp_seg_row, kp_seg_off, kp_unk_off;
    std::vector<int32_t> kp_seg_contig, kp_seg_len, kp_unk_len;
    std::vector<uint8_t> kp_seg_ns;
    std::vector<void *> kp_allocs;
    long long tmpl_ref = -1;           // reference version the name-block templates were built for (-1: not built)
    std::vector<void *> allocs;
};

struct mc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    DevTable T;                        // the current table: a copy of slots[cur].T
    TableSlot slots[MC_TABLE_SLOTS];
    int cur = -1;                      // slot of the current table
    int held = -1;                     // slot of the pass handed out last (its records may still be reduced: mc_site_counts)
    DevTable last_T;                   // ... and that table
    bool in_rerun = false;             // mc_wait_records is re-running a pass synchronously
    hipStream_t up_stream = nullptr;   // H2D of tables
    KpScratch kp;                      // the device parser's line-indexed scratch and contig table
    hipStream_t parse_stream = nullptr;  // its kernels (the text of the next shard is on its way on up_stream meanwhile)
    KpContigs kc;
    int64_t res_rows = 0, res_segs = 0, res_reads = 0;     // mc_ctx_reserve_tables
    long long ref_version = 0;
    int64_t scratch_nb = 0, scratch_tiles = 0;             // what the per-pass scratch below is sized for
    std::vector<void *> scratch_allocs;
    double *qual_own = nullptr;        // mc_ctx_set_read_quality's buffer
    int32_t n_qual_own = 0;
    DevRef R;
    DevMlp M;
    DevForest F;
    DevSimple Sc;                      // -c LR / -c NBC
    std::vector<void *> forest_allocs, simple_allocs;
    double *qual = nullptr;
    int32_t n_qual = 0;
    NbDesc *desc = nullptr;
    int64_t *nb_f0 = nullptr;
    DevRecords O;            // records of the last call (view: the fast path's buffers, or the merged ones)
    DevRecords Omain;        // the fast path's buffers
    DevRecords H;            // pinned host copy of the last call's records (mc_fetch_records_view)
    int h_k = 0;
    hipStream_t copy_stream = nullptr;
    hipStream_t copy_stream2 = nullptr;     // pipelined passes alternate between the two: no turnaround gap between transfers
    std::vector<void *> lit_allocs;
    int32_t *tile_local = nullptr;
    int64_t *tile_first = nullptr;
    int64_t *group_sum = nullptr;
    int32_t *tile_cnt = nullptr, *tile_half = nullptr;
    long long *tile_chunk = nullptr;
    Payload *payload_sorted = nullptr;   // payloads in file order (k1_list)
    int64_t *rare_list = nullptr;
    Payload *payload = nullptr;
    long long payload_cap = 0;
    int n_cu = 256;
    int emit_wgs = 4;              // resident k1_emit workgroups per CU (occupancy query)
    Counters *cnt = nullptr;
    int last_k = 0;
    int64_t last_n = 0;
    int64_t last_slots = 0;        // record slots the records handed out last occupy on the device (a fused dense pass: with holes in between)
    int last_fused_room = 0, last_rerun = 0;   // how the pass handed out last ran (mc_last_pass_info)
    size_t pack_min_bytes = 0;     // a pass's packed block: at least this (what a pass that did not fit said it needed, and a quarter)
    int fused_scale = 1;           // the fused dense pass (k1_fused): room per piece x this (doubled when a piece ran out of room)
    int64_t ref_total_len = 0;    // bases of the marked reference (record capacity guess)
    float times[5] = {0, 0, 0, 0, 0};
    std::vector<void *> ref_allocs, mlp_allocs, rec_allocs;
    int64_t payload_tiles = 0;     // tiles the payload buffer was sized for
    int payload_chunk = 0;         // ... and the chunk size
    // pipelined passes (mc_extract_features_async / mc_wait_records): two record sets, exported to pinned host memory
    struct AsyncBuf {
        DevRecords O;              // device records of the pass
        DevRecords H;              // pinned host memory
        K0Set K;                   // strand resolve output of the pass
        Counters *cnt = nullptr;   // its counters (device) ...
        Counters *st_host = nullptr; // ... and where the host reads them (pinned; st_dev: the same block as the GPU sees it)
        Counters *st_dev = nullptr;
        unsigned char *pack = nullptr, *pack_host = nullptr;   // what is copied out, packed by k_pack (device staging, pinned host)
        unsigned long long *chunk_cnt = nullptr;               // k_pack_count -> k_pack
        Payload *sorted = nullptr;                             // the pass's payloads in record order (k1_list -> k1_emit, k1_rare_dev)
        int64_t *rare = nullptr;                               // records k1_emit leaves to k1_rare_dev
        int32_t *piece_cnt = nullptr;                          // the fused dense pass: records of every piece (k1_fused -> k2_mlp)
        int32_t *piece_kw = nullptr;                           // ... its calls | their wide slot means << 16 (k1_fused -> the side stream's kernel)
        int64_t piece_cap = 0;
        size_t pack_bytes = 0;                                 // bytes of pack / pack_host
        int32_t *h_lo32 = nullptr;                             // in pack_host: the slot means' 32-bit parts, the wide ones' high halves,
        uint32_t *h_hi32 = nullptr;                            // the mask byte of every call (mc_calls_view)
        unsigned char *h_wmask = nullptr;
        int64_t h_n_wide = 0;
        int32_t *h_close32 = nullptr;                          // in pack_host: 32-bit closing rows (tables below 2^31 - 1 rows), else H.close_row
        bool close32 = false;
        int64_t h_n_calls = 0;
        // stage boundaries: dependencies between the streams, and the kernel times
        hipEvent_t ev_k0_start = nullptr, ev_k0_end = nullptr, ev_scan_start = nullptr, ev_scan_end = nullptr,
                   ev_emit_end = nullptr, ev_k2_start = nullptr, ev_k2_end = nullptr, ev_done = nullptr, ev_copied = nullptr;
        mc_params prm;
        int64_t cap = 0, n_nb = 0;
        int k = 0;
        bool used = false, copying = false, timed = true;
        bool one_kernel = false;   // the side stream ran as one kernel (k2_mlp<.., PACK>): its end is ev_done
        int want_text = 0;         // the rows as text, made on the device (mc_ctx_row_text was on when the pass was enqueued)
        int text_block = -1;       // ... the pinned block they are on their way to (mc_wait_records_begin), -1: none
        hipEvent_t ev_text = nullptr;
        hipEvent_t ev_list_end = nullptr;   // the pass's payloads are in file order (k1_list): what its emit, on the side stream, waits for
        hipEvent_t ev_emit_start = nullptr; // ... and when that emit began (a timed pass: its own time, not the wait for the side stream's turn)
        bool emit_aside = false;
        int fused_room = 0;        // > 0: the pass ran as ONE kernel (k1_fused) with this many record slots per piece -- holes in between
        int64_t slots = 0;         // ... record slots in all
        int slot = -1;             // table slot the pass scans
        unsigned long long pass_no = 0;   // what Counters.irregular_pass holds if the pass classified a block irregular
        const double *qual = nullptr;   // read qualities it was enqueued with
        int32_t n_qual = 0;
        std::vector<void *> dev_allocs, k0_allocs;
    } ab[MC_PASSES_IN_FLIGHT];
    hipStream_t side_stream = nullptr;   // classifier and packing of the pipelined passes
    // rows of text made on the device (mc_rowtext.hip; mc_ctx_row_text): scratch and one text buffer on the device -- the passes'
    // row writers run one after the other on the side stream --, pinned blocks on the host
    struct RowTextCtx {
        int on = 0;
        char lab_meth[8] = {}, lab_unmeth[8] = {};
        int lab_meth_len = 0, lab_unmeth_len = 0;
        RowTextScratch S = {};
        int64_t cap_rec = 0, cap_rows = 0, cap_wide = 0, cap_num = 0;
        std::vector<void *> allocs, out_allocs;
        char *out = nullptr;
        size_t out_cap = 0;
        double bytes_per_row = 0.0;        // room per call row (raised when a pass's rows did not fit)
        bool room_forced = false;          // (tests, MCALLER_ROW_TEXT_ROOM: a pass gets the room the estimate says, not what the buffers hold)
        struct Block {
            char *p = nullptr, *p_dev = nullptr;
            size_t cap = 0;
            RowTextStatus *st = nullptr, *st_dev = nullptr;
            std::atomic<int> busy{0};          // 0: free; else the ticket of the pass whose rows it holds (what mc_row_text_release must name)
        } blocks[MC_ROW_TEXT_BLOCKS];
        int next_ticket = 0;
        long long n_text = 0, n_no_block = 0, n_host_needed = 0, n_too_small = 0, n_other = 0;     // passes, by what became of their rows (MCALLER_VERBOSE)
        // the pass handed out last
        int last_block = -1;
        int64_t last_bytes = 0, last_rows = 0;
    } rt;
    int ab_head = 0, ab_tail = 0, ab_count = 0;
    unsigned long long pass_counter = 0, sync_pass_no = 0;   // pass numbers (never 0)
    int timing_every = 1;          // pipelined passes: the two timing events go with every n-th pass (mc_ctx_set_pass_timing)
    long long pass_seq = 0;
    int last_timed = 1;            // whether the pass handed out last carried them
    // per-site reduction (mc_site_*): counts on the device, RCCL communicator
    int32_t *site_cnt = nullptr;      // [2 * n_sites]: n_meth | n_total
    int64_t *site_first = nullptr;    // [n_sites]
    int64_t site_n = 0;
    hipStream_t site_stream = nullptr;                  // the reduction's own queue: a shard's records are reduced beside the passes in flight
    unsigned long long *site_status = nullptr;          // [4] device: pending, not-a-site, cross-contig (k_site_counts)
    unsigned long long *site_status_host = nullptr, *site_status_host_dev = nullptr;   // pinned copy the host reads (written by a kernel: no DMA)
    void *comm = nullptr;             // ncclComm_t
    int comm_world = 1, comm_rank = 0;
};

template <typename Tp>
static int dev_alloc(std::vector<void *> &pool, Tp **p, size_t n) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n * sizeof(Tp), 256));
    if (e != hipSuccess) {
        mc_set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(Tp), hipGetErrorString(e));
        return -10;
    }
    pool.push_back(q);
    *p = (Tp *)q;
    // MCALLER_POISON: fill every fresh device allocation with 0xAB (tests: a kernel that reads memory nobody wrote shows up
    // as a mismatch or a fault instead of silently reading zero pages)
    static const bool poison = getenv("MCALLER_POISON") != nullptr;
    if (poison) { (void)hipMemset(q, 0xAB, std::max<size_t>(n * sizeof(Tp), 256)); (void)hipDeviceSynchronize(); }
    return 0;
}

static void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}

static void free_pinned(DevRecords &H) {
    if (H.feats) (void)hipHostFree(H.feats);
    if (H.site_pos) (void)hipHostFree(H.site_pos);
    if (H.site_seg) (void)hipHostFree(H.site_seg);
    if (H.close_row) (void)hipHostFree(H.close_row);
    if (H.info) (void)hipHostFree(H.info);
    if (H.prob) (void)hipHostFree(H.prob);
    H = DevRecords();
}

static int ensure_pinned(mc_ctx *c, int64_t n, int k) {
    if (c->H.capacity >= n && c->h_k == k) return 0;
    free_pinned(c->H);
    const int64_t cap = std::max<int64_t>(n + n / 4, 1 << 16);
    HIP_TRY(hipHostMalloc((void **)&c->H.feats, (size_t)cap * k * 8, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->H.site_pos, (size_t)cap * 4, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->H.site_seg, (size_t)cap * 4, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->H.close_row, (size_t)cap * 8, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->H.info, (size_t)cap * 4, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->H.prob, (size_t)cap * 8, hipHostMallocDefault));
    c->H.capacity = cap;
    c->h_k = k;
    return 0;
}

// D2H of everything but the probabilities (they follow when the classifier is done)
static int copy_out_features(mc_ctx *c, int64_t n, int k, hipStream_t st) {
    HIP_TRY(hipMemcpyAsync(c->H.feats, c->O.feats, (size_t)n * k * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->H.site_pos, c->O.site_pos, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->H.site_seg, c->O.site_seg, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->H.close_row, c->O.close_row, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->H.info, c->O.info, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    return 0;
}

extern "C" int mc_comm_destroy(mc_ctx *c);
static void free_async(mc_ctx *c);
static int sync_pass_streams(mc_ctx *c);
static void slot_free(TableSlot &S);

// for the other translation units of the library (mc_train.hip)
int mc_internal_device(const mc_ctx *c) { return c->device; }
hipStream_t mc_internal_stream(const mc_ctx *c) { return c->stream; }

// Multi-GPU hosts: one process per GPU, and what a process copies out lands in ITS pinned memory.  Bound to the cores of the
// NUMA node the GPU hangs off, the process allocates there (first touch) and the DMA writes do not cross the socket link.
// -> the node (>= 0) when the calling thread was bound, -1 when the topology does not say (nothing changed).
extern "C" int mc_bind_to_device_numa_node(int device) {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return -1;
    for (char *p = bus; *p; ++p) *p = (char)tolower((unsigned char)*p);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return -1;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    char list[4096] = "";
    if (FILE *f = fopen(path, "r")) { if (!fgets(list, (int)sizeof(list), f)) list[0] = 0; fclose(f); }
    cpu_set_t *set = CPU_ALLOC(8192);
    if (!set) return -1;
    const size_t bytes = CPU_ALLOC_SIZE(8192);
    CPU_ZERO_S(bytes, set);
    int n_cpus = 0;
    for (char *p = list; *p;) {                       // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long cpu = a; cpu <= b && cpu < 8192; ++cpu) { CPU_SET_S((size_t)cpu, bytes, set); ++n_cpus; }
        while (*p == ',' || *p == '\n' || *p == ' ') ++p;
    }
    int rc = -1;
    if (n_cpus > 0 && sched_setaffinity(0, bytes, set) == 0) rc = node;
    CPU_FREE(set);
    return rc;
}

extern "C" int mc_ctx_create(int device, mc_ctx **out) {
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mc_set_error("no HIP device available (%s): libmcaller_hip has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return -11;
    }
    if (device < 0 || device >= n) {
        mc_set_error("device %d out of range (%d visible)", device, n);
        return -11;
    }
    HIP_TRY(hipSetDevice(device));
    mc_ctx *c = new mc_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
        int occ = 0;
        occ = mc_emit_occupancy();
        if (occ > 0) c->emit_wgs = occ;
        if (const char *e = getenv("MCALLER_EMIT_WGS")) { if (atoi(e) > 0) c->emit_wgs = atoi(e); }
        if (getenv("MCALLER_VERBOSE")) fprintf(stderr, "mcaller_hip: %d CUs, k1_emit occupancy %d workgroups/CU\n", c->n_cu, c->emit_wgs);
    }
    for (auto &ev : c->ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipMalloc((void **)&c->cnt, sizeof(Counters)));
    HIP_TRY(hipMemset(c->cnt, 0, sizeof(Counters)));
    *out = c;
    return 0;
}

extern "C" void mc_ctx_destroy(mc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_pass_streams(c);
    for (TableSlot &S : c->slots) {
        slot_free(S);
        for (hipEvent_t e : {S.ev_uploaded, S.ev_up_start, S.ev_val_start, S.ev_valid, S.ev_parsed, S.ev_text_up})
            if (e) (void)hipEventDestroy(e);
    }
    free_pool(c->kp.allocs);
    free_pool(c->kc.allocs);
    free_pool(c->scratch_allocs);
    free_pool(c->ref_allocs);
    free_pool(c->mlp_allocs);
    free_pool(c->forest_allocs);
    free_pool(c->simple_allocs);
    free_pool(c->rec_allocs);
    free_pool(c->lit_allocs);
    if (c->qual_own) (void)hipFree(c->qual_own);
    if (c->cnt) (void)hipFree(c->cnt);
    if (c->site_cnt) (void)hipFree(c->site_cnt);
    if (c->site_first) (void)hipFree(c->site_first);
    if (c->site_status) (void)hipFree(c->site_status);
    if (c->site_status_host) (void)hipHostFree(c->site_status_host);
    if (c->site_stream) { (void)hipStreamSynchronize(c->site_stream); (void)hipStreamDestroy(c->site_stream); }
    (void)sync_pass_streams(c);
    free_async(c);
    free_pool(c->rt.allocs);
    free_pool(c->rt.out_allocs);
    for (auto &blk : c->rt.blocks) {
        if (blk.p) (void)hipHostFree(blk.p);
        if (blk.st) (void)hipHostFree(blk.st);
    }
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    for (auto &b : c->ab)
        for (hipEvent_t e : {b.ev_k0_start, b.ev_k0_end, b.ev_scan_start, b.ev_scan_end, b.ev_emit_end, b.ev_k2_start, b.ev_k2_end, b.ev_done, b.ev_copied, b.ev_text, b.ev_list_end, b.ev_emit_start})
            if (e) (void)hipEventDestroy(e);
    mc_comm_destroy(c);
    for (auto &ev : c->ev) (void)hipEventDestroy(ev);
    free_pinned(c->H);
    (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream2);
    (void)hipStreamDestroy(c->up_stream);
    if (c->parse_stream) (void)hipStreamDestroy(c->parse_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mc_ctx_sync(mc_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return sync_pass_streams(c);
}

#define UP(dst, src, n, pool)                                                                          \
    do {                                                                                               \
        if (dev_alloc(pool, &(dst), (size_t)(n)) != 0) return -10;                                     \
        if ((n) > 0) HIP_TRY(hipMemcpyAsync((void *)(dst), (src), (size_t)(n) * sizeof(*(dst)), hipMemcpyHostToDevice, c->stream)); \
    } while (0)

// passes in flight read the reference; the text uploads and the device parser do not
static int sync_streams_that_read_the_reference(mc_ctx *c) {
    if (c->side_stream) HIP_TRY(hipStreamSynchronize(c->side_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream2));
    return 0;
}

extern "C" int mc_ctx_set_reference(mc_ctx *c, const mc_ref_view *h) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_streams_that_read_the_reference(c)) return rc;
    c->ref_version += 1;                                   // the name-block templates of every slot are stale
    free_pool(c->ref_allocs);
    DevRef &R = c->R;
    R.n_contigs = h->n_contigs;
    // site numbers: per contig, all '+' sites then all '-' sites, ascending position
    std::vector<int32_t> rank_f((size_t)h->n_words + 1), rank_r((size_t)h->n_words + 1);
    std::vector<int64_t> base((size_t)h->n_contigs * 2 + 2);
    int64_t n_sites = 0;
    for (int32_t ci = 0; ci < h->n_contigs; ++ci) {
        const int64_t w0 = h->word_off[ci], w1 = ci + 1 < h->n_contigs ? h->word_off[ci + 1] : h->n_words;
        for (int st = 0; st < 2; ++st) {
            const uint32_t *bits = st ? h->mbits_rev : h->mbits_fwd;
            std::vector<int32_t> &rank = st ? rank_r : rank_f;
            base[(size_t)ci * 2 + st] = n_sites;
            int32_t run = 0;
            for (int64_t w = w0; w < w1; ++w) {
                rank[(size_t)w] = run;
                run += __builtin_popcount(bits[w]);
            }
            n_sites += run;
        }
    }
    R.n_sites = n_sites;
    c->ref_total_len = 0;
    for (int32_t ci = 0; ci < h->n_contigs; ++ci) c->ref_total_len += h->contig_len[ci];
    // Everything goes through ONE pinned stage and is moved by a kernel: while a file is streamed the DMA engines are busy
    // with the text of the shards ahead, and a transfer submitted now would complete behind all of them (k_copy_bytes).
    struct Piece { void **dev; const void *src; size_t bytes, off; };
    size_t total = 0;
    auto piece = [&](void **dev, const void *src, size_t bytes) { Piece p{dev, src, bytes, total}; total += (bytes + 255) & ~(size_t)255; return p; };
    Piece pieces[] = {
        piece((void **)&R.contig_len, h->contig_len, (size_t)h->n_contigs * 8), piece((void **)&R.seq_off, h->seq_off, (size_t)h->n_contigs * 8),
        piece((void **)&R.word_off, h->word_off, (size_t)h->n_contigs * 8), piece((void **)&R.seq, h->seq, (size_t)h->n_seq_bytes),
        piece((void **)&R.mf, h->mbits_fwd, (size_t)h->n_words * 4), piece((void **)&R.mr, h->mbits_rev, (size_t)h->n_words * 4),
        piece((void **)&R.rank_f, rank_f.data(), (size_t)h->n_words * 4), piece((void **)&R.rank_r, rank_r.data(), (size_t)h->n_words * 4),
        piece((void **)&R.site_base, base.data(), (size_t)h->n_contigs * 2 * 8)};
    unsigned char *dev_block = nullptr, *stage = nullptr;
    if (dev_alloc(c->ref_allocs, &dev_block, total + 256)) return -10;
    HIP_TRY(hipHostMalloc((void **)&stage, total + 256, hipHostMallocDefault));
    for (const Piece &p : pieces) {
        if (p.bytes) memcpy(stage + p.off, p.src, p.bytes);
        *p.dev = dev_block + p.off;
    }
    int rc = copy_by_kernel(dev_block, stage, total, c->stream);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) { mc_set_error("mc_ctx_set_reference: the upload failed"); rc = -11; }
    (void)hipHostFree(stage);
    if (rc) return rc;
    if (c->site_stream) (void)hipStreamSynchronize(c->site_stream);
    if (c->site_cnt) { (void)hipFree(c->site_cnt); c->site_cnt = nullptr; }
    if (c->site_first) { (void)hipFree(c->site_first); c->site_first = nullptr; }
    c->site_n = 0;
    return 0;
}

// The reference from its raw bases, the masks made on the device (k_mark_*): h->seq holds the FASTA bytes of every contig
// (any case), h->mbits_* are not read.  *_fwd: the motif and what str.replace puts in its place for the '+' strand, *_rev: for
// the reverse complement; the motifs must not be able to overlap themselves (the caller checks; a one-base motif cannot).
extern "C" int mc_ctx_set_reference_motif(mc_ctx *c, const mc_ref_view *h, const char *motif_fwd, const char *repl_fwd, int32_t m_fwd,
                                          const char *motif_rev, const char *repl_rev, int32_t m_rev) {
    HIP_TRY(hipSetDevice(c->device));
    if (!h || h->n_contigs < 1 || !h->seq || m_fwd < 1 || m_fwd > 16 || m_rev < 1 || m_rev > 16 || !motif_fwd || !repl_fwd || !motif_rev ||
        !repl_rev || h->n_words < 1) {
        mc_set_error("mc_ctx_set_reference_motif: bad arguments (motifs of 1..16 bases)");
        return -12;
    }
    if (int rc = sync_streams_that_read_the_reference(c)) return rc;
    c->ref_version += 1;
    free_pool(c->ref_allocs);
    DevRef &R = c->R;
    R.n_contigs = h->n_contigs;
    c->ref_total_len = 0;
    for (int32_t ci = 0; ci < h->n_contigs; ++ci) c->ref_total_len += h->contig_len[ci];
    MarkMotif F, Rv;
    memset(&F, 0, sizeof(F)); memset(&Rv, 0, sizeof(Rv));
    memcpy(F.motif, motif_fwd, (size_t)m_fwd); memcpy(F.repl, repl_fwd, (size_t)m_fwd); F.m = m_fwd;
    memcpy(Rv.motif, motif_rev, (size_t)m_rev); memcpy(Rv.repl, repl_rev, (size_t)m_rev); Rv.m = m_rev;
    // one pinned stage for the small arrays and the bases, moved by a kernel (see mc_ctx_set_reference)
    const size_t nc = (size_t)h->n_contigs, nb = (size_t)h->n_seq_bytes, nw = (size_t)h->n_words;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_len = 0, o_soff = al(nc * 8), o_woff = o_soff + al(nc * 8), o_raw = o_woff + al(nc * 8), in_total = o_raw + al(nb + 16);
    unsigned char *dev_in = nullptr, *stage = nullptr;
    uint8_t *seq = nullptr;
    long long *cnt = nullptr, *off = nullptr, *total = nullptr;
    std::vector<void *> tmp;                                 // scratch of this call
    if (dev_alloc(c->ref_allocs, &dev_in, in_total) || dev_alloc(c->ref_allocs, &seq, nb + 16) || dev_alloc(c->ref_allocs, &R.mf, nw) ||
        dev_alloc(c->ref_allocs, &R.mr, nw) || dev_alloc(c->ref_allocs, &R.rank_f, nw) || dev_alloc(c->ref_allocs, &R.rank_r, nw) ||
        dev_alloc(c->ref_allocs, &R.site_base, nc * 2) || dev_alloc(tmp, &cnt, 2 * nw + 1) || dev_alloc(tmp, &off, 2 * nw + 1) ||
        dev_alloc(tmp, &total, 1)) {
        free_pool(tmp);
        return -10;
    }
    HIP_TRY(hipHostMalloc((void **)&stage, in_total, hipHostMallocDefault));
    memcpy(stage + o_len, h->contig_len, nc * 8);
    memcpy(stage + o_soff, h->seq_off, nc * 8);
    memcpy(stage + o_woff, h->word_off, nc * 8);
    memcpy(stage + o_raw, h->seq, nb);
    memset(stage + o_raw + nb, 0, 16);
    R.contig_len = (int64_t *)(dev_in + o_len); R.seq_off = (int64_t *)(dev_in + o_soff); R.word_off = (int64_t *)(dev_in + o_woff);
    R.seq = seq;
    hipStream_t st = c->stream;
    int rc = copy_by_kernel(dev_in, stage, in_total, st);
    if (!rc) {
        hipLaunchKernelGGL(k_mark_upper, dim3(1024), dim3(256), 0, st, (const uint8_t *)(dev_in + o_raw), seq, (int64_t)nb + 16);
        const unsigned wb = (unsigned)((nw + 255) / 256);
        hipLaunchKernelGGL(k_mark_words, dim3(wb), dim3(256), 0, st, (const uint8_t *)seq, (const int64_t *)R.contig_len,
                           (const int64_t *)R.seq_off, (const int64_t *)R.word_off, h->n_contigs, (int64_t)nw, F, Rv, R.mf, R.mr, cnt);
        hipLaunchKernelGGL(kp_scan, dim3(1), dim3(1024), 0, st, (const long long *)cnt, (int64_t)(2 * nw), off, total);
        hipLaunchKernelGGL(k_mark_ranks, dim3(wb), dim3(256), 0, st, (const long long *)off, (const int64_t *)R.word_off, h->n_contigs,
                           (int64_t)nw, R.rank_f, R.rank_r, R.site_base);
        long long n_sites = 0;
        if (hipMemcpyAsync(&n_sites, total, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            mc_set_error("mc_ctx_set_reference_motif: the marking failed: %s", hipGetErrorString(hipGetLastError()));
            rc = -11;
        }
        R.n_sites = n_sites;
    }
    (void)hipHostFree(stage);
    free_pool(tmp);
    if (rc) return rc;
    if (c->site_stream) (void)hipStreamSynchronize(c->site_stream);
    if (c->site_cnt) { (void)hipFree(c->site_cnt); c->site_cnt = nullptr; }
    if (c->site_first) { (void)hipFree(c->site_first); c->site_first = nullptr; }
    c->site_n = 0;
    return 0;
}

// the reference as the device holds it, back on the host (tests: the masks made on the device against the host's marking)
extern "C" int mc_ctx_fetch_reference(mc_ctx *c, uint8_t *seq, int64_t n_seq_bytes, uint32_t *mbits_fwd, uint32_t *mbits_rev, int32_t *rank_fwd,
                                      int32_t *rank_rev, int64_t n_words, int64_t *site_base, int64_t *n_sites) {
    HIP_TRY(hipSetDevice(c->device));
    const DevRef &R = c->R;
    if (!R.seq || !R.mf) {
        mc_set_error("mc_ctx_fetch_reference: no reference set");
        return -12;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (seq) HIP_TRY(hipMemcpy(seq, R.seq, (size_t)n_seq_bytes, hipMemcpyDeviceToHost));
    if (mbits_fwd) HIP_TRY(hipMemcpy(mbits_fwd, R.mf, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    if (mbits_rev) HIP_TRY(hipMemcpy(mbits_rev, R.mr, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    if (rank_fwd) HIP_TRY(hipMemcpy(rank_fwd, R.rank_f, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    if (rank_rev) HIP_TRY(hipMemcpy(rank_rev, R.rank_r, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    if (site_base) HIP_TRY(hipMemcpy(site_base, R.site_base, (size_t)R.n_contigs * 16, hipMemcpyDeviceToHost));
    if (n_sites) *n_sites = R.n_sites;
    return 0;
}

// ---- table slots ----
static void slot_free_parser(TableSlot &S) {
    free_pool(S.kp_allocs);
    for (void *p : {(void *)S.kp_head_h, (void *)S.kp_segs_h, (void *)S.kp_unknown_h, (void *)S.kp_flags_h})
        if (p) (void)hipHostFree(p);
    S.text = nullptr; S.cap_text = 0; S.kp_head = S.kp_head_h = nullptr; S.kp_segs = S.kp_segs_h = nullptr;
    S.kp_unknown = S.kp_unknown_h = nullptr; S.kp_flags_h = nullptr; S.kp_cap_flags = 0; S.kp_cap_segs = 0; S.kp_state = 0;
}

static void slot_free(TableSlot &S) {
    free_pool(S.allocs);
    slot_free_parser(S);
    if (S.stage) (void)hipHostFree(S.stage);
    S.stage = nullptr; S.small_dev = nullptr; S.small_cap = 0;
    S.pos = S.idx = nullptr; S.evmu = nullptr; S.flags = nullptr; S.nb_tmpl = nullptr; S.unit_pp = nullptr;
    S.cap_rows = S.cap_segs = S.cap_reads = 0;
    S.T = DevTable();
    S.qual = nullptr; S.n_qual = 0; S.tmpl_ref = -1;
}

// device memory + pinned stage of a slot for tables of up to (rows, segs, reads)
static int slot_ensure(mc_ctx *c, TableSlot &S, int64_t rows, int64_t segs, int64_t reads) {
    if (!S.ev_uploaded) {
        for (hipEvent_t *e : {&S.ev_uploaded, &S.ev_up_start, &S.ev_val_start, &S.ev_valid}) HIP_TRY(hipEventCreate(e));
        HIP_TRY(hipEventRecord(S.ev_valid, c->stream));           // (so that the first upload has something to wait for)
    }
    if (S.pos && rows <= S.cap_rows && segs <= S.cap_segs && reads <= S.cap_reads) return 0;
    // growing: whatever may still read the old arrays has to finish first (only ever happens without mc_ctx_reserve_tables)
    if (int rc = sync_pass_streams(c)) return rc;
    const bool fresh = !S.pos;
    slot_free(S);
    auto grow = [&](int64_t need, int64_t reserved) { return std::max<int64_t>(fresh ? need : need + need / 4, reserved); };
    S.cap_rows = grow(rows, c->res_rows);
    S.cap_segs = std::max<int64_t>(grow(segs, c->res_segs), 16);
    S.cap_reads = std::max<int64_t>(grow(reads, c->res_reads), 16);
    const int64_t padded = ((S.cap_rows + TILE - 1) / TILE) * TILE + TILE + FRONT;     // (whole tiles of the scan)
    const SmallLayout L = small_layout(S.cap_segs, padded / TILE, S.cap_reads);
    if (dev_alloc(S.allocs, &S.pos, (size_t)padded) || dev_alloc(S.allocs, &S.idx, (size_t)padded) ||
        dev_alloc(S.allocs, &S.evmu, (size_t)padded) || dev_alloc(S.allocs, &S.flags, (size_t)padded) ||
        dev_alloc(S.allocs, &S.unit_pp, (size_t)padded / 8 + 8))
        return -10;
    // (FRONT rows of padding before row 0 of the columns k1_emit looks back into: rows -1 .. -64 are readable)
    S.pos += FRONT; S.evmu += FRONT; S.flags += FRONT;
    if (
        dev_alloc(S.allocs, &S.nb_tmpl, (size_t)S.cap_segs + 1) || dev_alloc(S.allocs, &S.small_dev, L.total))
        return -10;
    HIP_TRY(hipHostMalloc((void **)&S.stage, L.total, hipHostMallocDefault));
    S.small_cap = L.total;
    return 0;
}

// the scratch all passes share (ordered by the ctx stream): tile descriptors / counts / chunks, strand-resolve output of
// the synchronous pass
static int ensure_scratch(mc_ctx *c, int64_t n_nb, int64_t n_tiles) {
    if (c->desc && n_nb <= c->scratch_nb && n_tiles <= c->scratch_tiles) return 0;
    if (int rc = sync_pass_streams(c)) return rc;
    free_pool(c->scratch_allocs);
    const int64_t res_tiles = c->res_rows ? (c->res_rows + TILE - 1) / TILE : 0;
    // (with head room: the tables of a stream differ by a few name blocks, and growing again means waiting for everything in flight)
    const int64_t nb = std::max<int64_t>(std::max<int64_t>(n_nb + n_nb / 4 + 64, c->res_segs), c->scratch_nb);
    const int64_t nt = std::max<int64_t>(std::max<int64_t>(n_tiles + n_tiles / 8 + 16, res_tiles), c->scratch_tiles);
    std::vector<void *> &P = c->scratch_allocs;
    if (dev_alloc(P, &c->desc, (size_t)nb + 1) || dev_alloc(P, &c->nb_f0, (size_t)nb + 1) ||
        dev_alloc(P, &c->tile_chunk, ((size_t)nt + 1) * NCHUNK) || dev_alloc(P, &c->tile_local, (size_t)nt + 1) ||
        dev_alloc(P, &c->group_sum, (size_t)(nt / GROUP + 2)) || dev_alloc(P, &c->tile_cnt, (size_t)nt + 1) || dev_alloc(P, &c->tile_half, (size_t)nt + 1) ||
        dev_alloc(P, &c->tile_first, (size_t)nt + 1))
        return -10;
    c->scratch_nb = nb;
    c->scratch_tiles = nt;
    return 0;
}

extern "C" int mc_ctx_reserve_tables(mc_ctx *c, int64_t max_rows, int32_t max_segs, int32_t max_reads) {
    HIP_TRY(hipSetDevice(c->device));
    if (max_rows < 0 || max_segs < 0 || max_reads < 0) {
        mc_set_error("mc_ctx_reserve_tables: negative size");
        return -12;
    }
    c->res_rows = std::max(c->res_rows, max_rows);
    c->res_segs = std::max<int64_t>(c->res_segs, max_segs);
    c->res_reads = std::max<int64_t>(c->res_reads, max_reads);
    for (TableSlot &S : c->slots)
        if (int rc = slot_ensure(c, S, c->res_rows, c->res_segs, c->res_reads)) return rc;
    return ensure_scratch(c, c->res_segs, (c->res_rows + TILE - 1) / TILE);
}

// a free slot: not scanned by a pass in flight, not holding the records handed out last, not being filled by the device parser
static int free_slot(mc_ctx *c, const char *who) {
    for (int i = 1; i <= MC_TABLE_SLOTS; ++i) {
        const int sidx = (std::max(c->cur, 0) + i) % MC_TABLE_SLOTS;
        if (c->slots[sidx].refs == 0 && sidx != c->held) return sidx;
    }
    mc_set_error("%s: all %d table slots are being scanned; call mc_wait_records first", who, MC_TABLE_SLOTS);
    return -1;
}

// What makes the rows in slot `at` a table: the small arrays (segments, name blocks -- maximal runs of segments with one read
// name --, the name block of every tile's first row, read qualities) laid out in the pinned stage and sent, the per-table
// kernel behind them; the table becomes the current one.  cols: the host columns to send first (mc_ctx_upload_table_async),
// or nullptr: the device parser has put them into the slot already (mc_ctx_parse_finish).  seg_name_start[sg] (or, if
// nullptr, MC_F_NAME_START of the segment's first row in cols->flags): the segment starts a name block.
static int fill_slot(mc_ctx *c, int at, int64_t n, int32_t n_seg, const int64_t *seg_row_begin, const int32_t *seg_read_in,
                     const int32_t *seg_contig_in, const uint8_t *seg_name_start, int32_t n_reads, const double *read_qual,
                     const mc_table_view *cols) {
    TableSlot &S = c->slots[at];
    S.from_parser = cols == nullptr;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const SmallLayout L = small_layout(n_seg, n_tiles, read_qual ? n_reads : 0);
    unsigned char *st = S.stage;
    int64_t *seg_begin = (int64_t *)(st + L.seg_begin), *nb_row = (int64_t *)(st + L.nb_row_begin);
    int32_t *seg_read = (int32_t *)(st + L.seg_read), *seg_contig = (int32_t *)(st + L.seg_contig);
    int32_t *nb_seg = (int32_t *)(st + L.nb_seg_begin), *nb_read = (int32_t *)(st + L.nb_read), *tile_nb = (int32_t *)(st + L.tile_nb);
    uint8_t *nb_rep = st + L.nb_repeat;
    uint32_t *nb_vf = (uint32_t *)(st + L.nb_vflags);
    if (n_seg > 0) {
        memcpy(seg_begin, seg_row_begin, (size_t)n_seg * 8);
        seg_begin[n_seg] = n;
        memcpy(seg_read, seg_read_in, (size_t)n_seg * 4);
        memcpy(seg_contig, seg_contig_in, (size_t)n_seg * 4);
    } else seg_begin[0] = 0;
    std::vector<uint8_t> seen((size_t)std::max(n_reads, 1), 0);
    int has_rep = 0;
    int32_t n_nb = 0;
    for (int32_t sg = 0; sg < n_seg; ++sg) {
        const int64_t rb = seg_row_begin[sg];
        if (rb < 0 || rb >= n || (sg > 0 && rb <= seg_row_begin[sg - 1])) {
            mc_set_error("segment %d: row %lld out of order", sg, (long long)rb);
            return -12;
        }
        if (sg == 0 || (seg_name_start ? seg_name_start[sg] != 0 : (cols->flags[rb] & MC_F_NAME_START) != 0)) {
            const int32_t rd = seg_read_in[sg];
            if (rd < 0 || rd >= n_reads) {
                mc_set_error("segment %d: read id %d out of range", sg, rd);
                return -12;
            }
            nb_row[n_nb] = rb;
            nb_seg[n_nb] = sg;
            nb_read[n_nb] = rd;
            nb_rep[n_nb] = seen[(size_t)rd];
            has_rep |= seen[(size_t)rd];
            seen[(size_t)rd] = 1;
            if (n_nb > 0) nb_vf[n_nb - 1] = (sg - nb_seg[n_nb - 1] > 1) ? V_MULTI_SEG : 0u;
            ++n_nb;
        }
    }
    if (n_nb > 0) nb_vf[n_nb - 1] = (n_seg - nb_seg[n_nb - 1] > 1) ? V_MULTI_SEG : 0u;
    nb_row[n_nb] = n;
    nb_seg[n_nb] = n_seg;
    nb_vf[n_nb] = 0u;
    {
        int32_t b = 0;                                     // last block that starts at or before the tile's first row
        for (int64_t t = 0; t < n_tiles; ++t) {
            while (b + 1 < n_nb && nb_row[b + 1] <= t * TILE) ++b;
            tile_nb[t] = b;
        }
    }
    if (read_qual && n_reads > 0) memcpy(st + L.qual, read_qual, (size_t)n_reads * 8);

    // ---- the slot's table ----
    DevTable &T = S.T;
    T = DevTable();
    T.n_rows = n; T.n_seg = n_seg; T.n_reads = n_reads; T.n_nb = n_nb; T.n_tiles = n_tiles; T.has_repeats = has_rep;
    T.pos = S.pos; T.idx = S.idx; T.evmu = S.evmu; T.flags = S.flags; T.nb_tmpl = S.nb_tmpl; T.unit_pp = S.unit_pp;
    unsigned char *dv = S.small_dev;
    T.seg_begin = (int64_t *)(dv + L.seg_begin); T.seg_read = (int32_t *)(dv + L.seg_read); T.seg_contig = (int32_t *)(dv + L.seg_contig);
    T.nb_row_begin = (int64_t *)(dv + L.nb_row_begin); T.nb_seg_begin = (int32_t *)(dv + L.nb_seg_begin);
    T.nb_read = (int32_t *)(dv + L.nb_read); T.nb_repeat = dv + L.nb_repeat; T.nb_vflags = (uint32_t *)(dv + L.nb_vflags);
    T.tile_nb = (int32_t *)(dv + L.tile_nb);
    S.qual = read_qual ? (double *)(dv + L.qual) : nullptr;
    S.n_qual = read_qual ? n_reads : 0;
    S.tmpl_ref = -1;
    S.passes = 0;                                          // (the first pass over these rows validates them)
    S.summarized = false;

    // ---- H2D on the upload stream (nothing reads the slot: its passes have been handed out); the ctx stream waits for the
    //      transfer ----
    // (a device-parsed table: the upload stream is busy with the NEXT shard's text by now -- the small arrays go on the ctx
    // stream, in front of the kernels that read them)
    hipStream_t us = cols ? c->up_stream : c->stream;
    if (cols) {
        HIP_TRY(hipStreamWaitEvent(us, S.ev_valid, 0));    // the small arrays of the slot's previous table (it may never have been scanned)
        HIP_TRY(hipEventRecord(S.ev_up_start, us));
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(T.pos, cols->pos, (size_t)n * 4, hipMemcpyHostToDevice, us));
            HIP_TRY(hipMemcpyAsync(T.evmu, cols->event_model_e4, (size_t)n * 8, hipMemcpyHostToDevice, us));
            HIP_TRY(hipMemcpyAsync(T.idx, cols->event_idx, (size_t)n * 4, hipMemcpyHostToDevice, us));
            HIP_TRY(hipMemcpyAsync(T.flags, cols->flags, (size_t)n, hipMemcpyHostToDevice, us));
        }
    }
    if (!cols && L.total <= COPY_BY_KERNEL_MAX) { if (int rc = copy_by_kernel(dv, st, L.total, us)) return rc; }     // (not behind the next shard's text)
    else HIP_TRY(hipMemcpyAsync(dv, st, L.total, hipMemcpyHostToDevice, us));
    HIP_TRY(hipEventRecord(S.ev_uploaded, us));
    HIP_TRY(hipStreamWaitEvent(c->stream, S.ev_uploaded, 0));
    HIP_TRY(hipEventRecord(S.ev_val_start, c->stream));
    HIP_TRY(hipEventRecord(S.ev_valid, c->stream));
    HIP_TRY(hipGetLastError());
    c->T = T;
    c->cur = at;
    S.holds_table = true;
    if (read_qual) { c->qual = S.qual; c->n_qual = S.n_qual; }
    else { c->qual = c->qual_own; c->n_qual = c->n_qual_own; }       // mc_ctx_set_read_quality's table applies
    return 0;
}

extern "C" int mc_ctx_upload_table_async(mc_ctx *c, const mc_table_view *h, const double *read_qual, int32_t *slot_out) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot_out) *slot_out = -1;
    const int64_t n = h->n_rows;
    if (n < 0 || h->n_seg < 0 || h->n_reads < 0 || (n > 0 && h->n_seg == 0)) {
        mc_set_error("mc_ctx_upload_table_async: malformed table (%lld rows, %d segments, %d reads)", (long long)n, h->n_seg, h->n_reads);
        return -12;
    }
    const int at = free_slot(c, "mc_ctx_upload_table_async");
    if (at < 0) return MC_E_NO_FREE_SLOT;
    TableSlot &S = c->slots[at];
    if (int rc = slot_ensure(c, S, n, h->n_seg, h->n_reads)) return rc;
    HIP_TRY(hipEventSynchronize(S.ev_uploaded));          // the stage is about to be rewritten (long done: the slot was idle)
    if (int rc = fill_slot(c, at, n, h->n_seg, h->seg_row_begin, h->seg_read, h->seg_contig, nullptr, h->n_reads, read_qual, h)) return rc;
    if (slot_out) *slot_out = at;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// The device parser's host side (kernels: mc_devparse.inc).  mc_ctx_parse_begin sends a shard's text and enqueues the
// kernels that turn it into the columns of a table slot; mc_ctx_parse_end waits and hands out what the host needs to name
// things (segments with the place of their read name in the text, unknown contig tokens, the flag column);
// mc_ctx_parse_finish takes the read ids and qualities and makes the slot's rows the current table -- from there on the slot
// is what mc_ctx_upload_table_async would have left.  All on the upload stream; begin for shard i+1 may be called before end
// for shard i.
// ---------------------------------------------------------------------------------------------------
static int kp_ensure_scratch(mc_ctx *c, int64_t cap_lines, int64_t n_tiles) {
    KpScratch &K = c->kp;
    if (K.cap_lines >= cap_lines && K.cap_tiles >= n_tiles) return 0;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    if (c->parse_stream) HIP_TRY(hipStreamSynchronize(c->parse_stream));
    free_pool(K.allocs);
    K.cap_lines = std::max(cap_lines, K.cap_lines);
    K.cap_tiles = std::max(n_tiles, K.cap_tiles);
    const size_t n = (size_t)K.cap_lines + 256, nt = (size_t)std::max<int64_t>(K.cap_tiles, (K.cap_lines + 255) / 256) + 1;
    if (dev_alloc(K.allocs, &K.line_start, n + 1) || dev_alloc(K.allocs, &K.pos, n) || dev_alloc(K.allocs, &K.idx, n) ||
        dev_alloc(K.allocs, &K.ev, n) || dev_alloc(K.allocs, &K.mu, n) || dev_alloc(K.allocs, &K.contig, n) ||
        dev_alloc(K.allocs, &K.name_off, n) || dev_alloc(K.allocs, &K.name_len, n) || dev_alloc(K.allocs, &K.fl, n) ||
        dev_alloc(K.allocs, &K.status, n) || dev_alloc(K.allocs, &K.tile_cnt, nt) || dev_alloc(K.allocs, &K.tile_off, nt))
        return -10;
    return 0;
}

static int kp_set_contigs(mc_ctx *c, const char *const *names, int32_t n) {
    KpContigs &C = c->kc;
    bool same = (int)C.names.size() == n && C.hash;
    for (int i = 0; same && i < n; ++i) same = C.names[(size_t)i] == names[i];
    if (same) return 0;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    if (c->parse_stream) HIP_TRY(hipStreamSynchronize(c->parse_stream));
    free_pool(C.allocs);
    C.names.assign(names, names + n);
    int size = 16;
    while (size < 2 * n + 2) size *= 2;
    std::vector<uint32_t> hash((size_t)size, 0), off((size_t)std::max(n, 1)), len((size_t)std::max(n, 1));
    std::vector<int32_t> id((size_t)size, -1);
    std::string chars;
    for (int i = 0; i < n; ++i) {
        off[(size_t)i] = (uint32_t)chars.size();
        len[(size_t)i] = (uint32_t)C.names[(size_t)i].size();
        chars += C.names[(size_t)i];
        uint32_t h = 2166136261u;
        for (unsigned char ch : C.names[(size_t)i]) h = (h ^ ch) * 16777619u;
        if (h == 0) h = 1;
        bool dup = false;                                   // the first id of a name wins, like the FASTA scan (:77-81)
        int slot = (int)(h & (uint32_t)(size - 1));
        for (; hash[(size_t)slot]; slot = (slot + 1) & (size - 1))
            if (hash[(size_t)slot] == h && C.names[(size_t)id[(size_t)slot]] == C.names[(size_t)i]) { dup = true; break; }
        if (!dup) { hash[(size_t)slot] = h; id[(size_t)slot] = i; }
    }
    chars.push_back('\0');
    C.table_mask = size - 1;
    C.n = n;
    hipStream_t us = c->up_stream;
    if (dev_alloc(C.allocs, &C.hash, (size_t)size) || dev_alloc(C.allocs, &C.id, (size_t)size) ||
        dev_alloc(C.allocs, &C.name_off, off.size()) || dev_alloc(C.allocs, &C.name_len, len.size()) ||
        dev_alloc(C.allocs, &C.chars, chars.size()))
        return -10;
    HIP_TRY(hipMemcpyAsync(C.hash, hash.data(), (size_t)size * 4, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(C.id, id.data(), (size_t)size * 4, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(C.name_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(C.name_len, len.data(), len.size() * 4, hipMemcpyHostToDevice, us));
    HIP_TRY(hipMemcpyAsync(C.chars, chars.data(), chars.size(), hipMemcpyHostToDevice, us));
    HIP_TRY(hipStreamSynchronize(us));                      // (the vectors go out of scope)
    return 0;
}

static int kp_ensure_slot(mc_ctx *c, TableSlot &S, int64_t n_bytes) {
    if (!S.ev_parsed) { HIP_TRY(hipEventCreate(&S.ev_parsed)); HIP_TRY(hipEventCreate(&S.ev_text_up)); }
    const int cap_segs = (int)std::min<int64_t>(S.cap_segs, 1 << 24);
    if (S.text && S.cap_text >= n_bytes + 64 && S.kp_cap_flags >= S.cap_rows && S.kp_cap_segs >= cap_segs) return 0;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    if (c->parse_stream) HIP_TRY(hipStreamSynchronize(c->parse_stream));
    slot_free_parser(S);
    S.cap_text = std::max<int64_t>(n_bytes + n_bytes / 8, (int64_t)1 << 20) + 64;
    if (dev_alloc(S.kp_allocs, &S.text, (size_t)S.cap_text) || dev_alloc(S.kp_allocs, &S.kp_head, 1) ||
        dev_alloc(S.kp_allocs, &S.kp_segs, (size_t)cap_segs) || dev_alloc(S.kp_allocs, &S.kp_unknown, (size_t)KP_MAX_UNKNOWN))
        return -10;
    S.kp_cap_segs = cap_segs;
    S.kp_cap_flags = S.cap_rows;
    HIP_TRY(hipHostMalloc((void **)&S.kp_head_h, sizeof(KpHead), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&S.kp_segs_h, (size_t)cap_segs * sizeof(KpSeg), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&S.kp_unknown_h, (size_t)KP_MAX_UNKNOWN * sizeof(KpUnknown), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&S.kp_flags_h, (size_t)std::max<int64_t>(S.cap_rows, 1), hipHostMallocDefault));
    return 0;
}

extern "C" int mc_ctx_parse_begin(mc_ctx *c, const char *text, int64_t n_bytes, const char *const *contig_names, int32_t n_contigs,
                                  int64_t max_rows, int32_t *slot_out) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot_out) *slot_out = -1;
    if (!text || n_bytes < 0 || n_bytes >= ((int64_t)1 << 32) || n_contigs < 0 || max_rows < 0) {
        mc_set_error("mc_ctx_parse_begin: bad arguments (%lld bytes of text; at most 4 GB per shard)", (long long)n_bytes);
        return -12;
    }
    const int at = free_slot(c, "mc_ctx_parse_begin");
    if (at < 0) return MC_E_NO_FREE_SLOT;
    TableSlot &S = c->slots[at];
    // rows: what the caller expects (the slots were sized by mc_ctx_reserve_tables, or grow here); a shard with more rows or
    // segments than the slot holds comes back from mc_ctx_parse_end as "needs the host parser"
    const int64_t rows = std::max<int64_t>(max_rows, 1);
    if (int rc = slot_ensure(c, S, rows, std::max<int64_t>(rows / 16, 64), std::max<int64_t>(rows / 16, 64))) return rc;
    if (int rc = kp_ensure_slot(c, S, n_bytes)) return rc;
    const int64_t n_tiles = (n_bytes + KP_TILE - 1) / KP_TILE;
    if (int rc = kp_ensure_scratch(c, S.cap_rows + 65536, n_tiles)) return rc;
    if (int rc = kp_set_contigs(c, contig_names, n_contigs)) return rc;
    KpScratch &K = c->kp;
    if (!c->parse_stream) HIP_TRY(hipStreamCreateWithFlags(&c->parse_stream, hipStreamNonBlocking));
    {   // the text on the upload stream, the kernels behind it on their own: the next shard's text travels while they run
        hipStream_t up = c->up_stream;
        HIP_TRY(hipStreamWaitEvent(up, S.ev_valid, 0));    // the small arrays of the slot's previous table (it may never have been scanned)
        HIP_TRY(hipEventRecord(S.ev_up_start, up));
        static const KpHead zero_head = {0, 0, 0, 0, 0, 0x7fffffffffffffffll, 0, 0};
        HIP_TRY(hipMemcpyAsync(S.kp_head, &zero_head, sizeof(KpHead), hipMemcpyHostToDevice, up));
        if (n_bytes > 0) HIP_TRY(hipMemcpyAsync(S.text, text, (size_t)n_bytes, hipMemcpyHostToDevice, up));
        HIP_TRY(hipEventRecord(S.ev_text_up, up));
    }
    hipStream_t us = c->parse_stream;
    const int kp_debug = getenv("MCALLER_KP_SYNC") ? atoi(getenv("MCALLER_KP_SYNC")) : 0;     // (finding the kernel that faults: bit i = wait behind step i)
    int kp_step = 0;
#define KP_STEP(name) do { if ((kp_debug >> kp_step++) & 1) { HIP_TRY(hipStreamSynchronize(us)); fprintf(stderr, "kp: %s ok\n", name); } } while (0)
    HIP_TRY(hipStreamWaitEvent(us, S.ev_text_up, 0));
    if (n_tiles > 0) {
        hipLaunchKernelGGL(kp_count, dim3((unsigned)n_tiles), dim3(KP_THREADS), 0, us, (const char *)S.text, n_bytes, K.tile_cnt);
        KP_STEP("kp_count");
        hipLaunchKernelGGL(kp_scan, dim3(1), dim3(1024), 0, us, (const long long *)K.tile_cnt, n_tiles, K.tile_off, &S.kp_head->n_newlines);
        KP_STEP("kp_scan");
        hipLaunchKernelGGL(kp_starts, dim3((unsigned)n_tiles), dim3(KP_THREADS), 0, us, (const char *)S.text, n_bytes,
                           (const long long *)K.tile_off, K.line_start, K.cap_lines, S.kp_head);
        KP_STEP("kp_starts");
        const int64_t cap_lines = K.cap_lines;
        const unsigned line_blocks = (unsigned)((cap_lines + 255) / 256);
        KpParseArgs PA;
        PA.text = S.text; PA.n_bytes = n_bytes; PA.line_start = K.line_start; PA.head = S.kp_head; PA.head_w = S.kp_head;
        PA.cap_lines = cap_lines; PA.c_hash = c->kc.hash; PA.c_id = c->kc.id; PA.c_off = c->kc.name_off; PA.c_len = c->kc.name_len;
        PA.c_chars = c->kc.chars; PA.c_mask = c->kc.table_mask;
        PA.pos = K.pos; PA.idx = K.idx; PA.ev = K.ev; PA.mu = K.mu; PA.contig = K.contig; PA.name_off = K.name_off; PA.name_len = K.name_len;
        PA.fl = K.fl; PA.status = K.status;
        hipLaunchKernelGGL(kp_parse, dim3(line_blocks), dim3(256), KP_STAGE + 16, us, PA);
        KP_STEP("kp_parse");
        hipLaunchKernelGGL(kp_count_rows, dim3(line_blocks), dim3(256), 0, us, (const uint8_t *)K.status, (const KpHead *)S.kp_head,
                           cap_lines, K.tile_cnt);
        KP_STEP("kp_count_rows");
        hipLaunchKernelGGL(kp_scan, dim3(1), dim3(1024), 0, us, (const long long *)K.tile_cnt, (int64_t)line_blocks, K.tile_off,
                           &S.kp_head->n_rows);
        KP_STEP("kp_scan");
        KpPlaceArgs QA;
        QA.text = S.text; QA.head = S.kp_head; QA.head_w = S.kp_head; QA.cap_lines = cap_lines; QA.cap_rows = S.cap_rows;
        QA.blk_off = K.tile_off; QA.pos = K.pos; QA.idx = K.idx; QA.ev = K.ev; QA.mu = K.mu; QA.contig = K.contig;
        QA.name_off = K.name_off; QA.name_len = K.name_len; QA.fl = K.fl; QA.status = K.status;
        QA.t_pos = S.pos; QA.t_idx = S.idx; QA.t_evmu = S.evmu; QA.t_flags = S.flags; QA.segs = S.kp_segs; QA.cap_segs = S.kp_cap_segs;
        QA.unknown = S.kp_unknown;
        hipLaunchKernelGGL(kp_place, dim3(line_blocks), dim3(256), 0, us, QA);
        KP_STEP("kp_place");
    }
    // what mc_ctx_parse_end hands out, on its way as soon as it exists: the head, the first segments and unknown tokens (a
    // shard with more of them gets the rest when it is waited for), the flag column
    // (by kernel: a DMA transfer would queue behind the text of the shards that follow)
    S.kp_flags_sent = std::min<int64_t>(S.cap_rows, std::min<int64_t>(rows + rows / 4 + 4096, (int64_t)COPY_BY_KERNEL_MAX));
    if (int rc = copy_by_kernel(S.kp_segs_h, S.kp_segs, (size_t)std::min(S.kp_cap_segs, KP_EAGER_SEGS) * sizeof(KpSeg), us)) return rc;
    if (int rc = copy_by_kernel(S.kp_unknown_h, S.kp_unknown, (size_t)KP_EAGER_UNKNOWN * sizeof(KpUnknown), us)) return rc;
    if (int rc = copy_by_kernel(S.kp_flags_h, S.flags, (size_t)S.kp_flags_sent, us)) return rc;
    if (int rc = copy_by_kernel(S.kp_head_h, S.kp_head, sizeof(KpHead), us)) return rc;
    HIP_TRY(hipEventRecord(S.ev_parsed, us));
    KP_STEP("copies");
#undef KP_STEP
    HIP_TRY(hipGetLastError());
    S.refs += 1;                                            // the slot is taken until mc_ctx_parse_finish / _abandon
    S.holds_table = false;                                  // (the columns are being overwritten: S.T describes them no more)
    S.from_parser = false;
    S.kp_state = 1;
    S.kp_bytes = n_bytes;
    if (slot_out) *slot_out = at;
    return 0;
}

static int kp_slot(mc_ctx *c, int32_t slot, int state, const char *who, TableSlot **S) {
    if (slot < 0 || slot >= MC_TABLE_SLOTS || c->slots[slot].kp_state != state) {
        mc_set_error("%s: slot %d is not in that state", who, slot);
        return -12;
    }
    *S = &c->slots[slot];
    return 0;
}

extern "C" int mc_ctx_parse_end(mc_ctx *c, int32_t slot, mc_devparse_result *out) {
    HIP_TRY(hipSetDevice(c->device));
    TableSlot *Sp;
    if (int rc = kp_slot(c, slot, 1, "mc_ctx_parse_end", &Sp)) return rc;
    TableSlot &S = *Sp;
    memset(out, 0, sizeof(*out));
    HIP_TRY(hipEventSynchronize(S.ev_parsed));
    const KpHead H = *S.kp_head_h;
    S.kp_state = 2;
    out->n_lines = H.n_lines; out->n_rows = H.n_rows; out->n_seg = H.n_seg; out->n_unknown = H.n_unknown;
    if (H.overflow || H.first_host_line != 0x7fffffffffffffffll || H.n_rows > S.cap_rows || H.n_seg > S.kp_cap_segs) {
        out->status = 1;
        if (H.first_host_line != 0x7fffffffffffffffll)
            mc_set_error("device parser: line %lld needs the host parser (a number form or value beyond the fast path)", H.first_host_line);
        else
            mc_set_error("device parser: %lld lines, %lld rows, %d segments, %d unknown-contig lines do not fit the slot", H.n_lines, H.n_rows,
                         H.n_seg, H.n_unknown);
        return 0;
    }
    // (what did not travel with the head: blocking copies -- the streams are busy with the next shard)
    if (H.n_seg > KP_EAGER_SEGS) HIP_TRY(hipMemcpy(S.kp_segs_h, S.kp_segs, (size_t)H.n_seg * sizeof(KpSeg), hipMemcpyDeviceToHost));
    if (H.n_unknown > KP_EAGER_UNKNOWN) HIP_TRY(hipMemcpy(S.kp_unknown_h, S.kp_unknown, (size_t)H.n_unknown * sizeof(KpUnknown), hipMemcpyDeviceToHost));
    if (H.n_rows > S.kp_flags_sent) HIP_TRY(hipMemcpy(S.kp_flags_h, S.flags, (size_t)H.n_rows, hipMemcpyDeviceToHost));
    // segments and unknown lines were listed in the order the lanes got there: file order is by row / by line
    std::sort(S.kp_segs_h, S.kp_segs_h + H.n_seg, [](const KpSeg &a, const KpSeg &b) { return a.row < b.row; });
    std::sort(S.kp_unknown_h, S.kp_unknown_h + H.n_unknown, [](const KpUnknown &a, const KpUnknown &b) { return a.line < b.line; });
    S.kp_seg_row.resize((size_t)H.n_seg); S.kp_seg_off.resize((size_t)H.n_seg); S.kp_seg_contig.resize((size_t)H.n_seg);
    S.kp_seg_len.resize((size_t)H.n_seg); S.kp_seg_ns.resize((size_t)H.n_seg);
    for (int i = 0; i < H.n_seg; ++i) {
        const KpSeg &g = S.kp_segs_h[i];
        S.kp_seg_row[(size_t)i] = g.row; S.kp_seg_off[(size_t)i] = g.name_off; S.kp_seg_contig[(size_t)i] = g.contig;
        S.kp_seg_len[(size_t)i] = g.name_len; S.kp_seg_ns[(size_t)i] = (uint8_t)g.name_start;
    }
    S.kp_unk_off.resize((size_t)H.n_unknown); S.kp_unk_len.resize((size_t)H.n_unknown);
    for (int i = 0; i < H.n_unknown; ++i) { S.kp_unk_off[(size_t)i] = S.kp_unknown_h[i].off; S.kp_unk_len[(size_t)i] = S.kp_unknown_h[i].len; }
    out->seg_row_begin = S.kp_seg_row.data(); out->seg_contig = S.kp_seg_contig.data(); out->seg_name_off = S.kp_seg_off.data();
    out->seg_name_len = S.kp_seg_len.data(); out->seg_name_start = S.kp_seg_ns.data();
    out->unknown_off = S.kp_unk_off.data(); out->unknown_len = S.kp_unk_len.data();
    out->flags = S.kp_flags_h;
    return 0;
}

extern "C" int mc_ctx_parse_finish(mc_ctx *c, int32_t slot, const int32_t *seg_read, int32_t n_reads, const double *read_qual) {
    HIP_TRY(hipSetDevice(c->device));
    TableSlot *Sp;
    if (int rc = kp_slot(c, slot, 2, "mc_ctx_parse_finish", &Sp)) return rc;
    TableSlot &S = *Sp;
    const KpHead H = *S.kp_head_h;
    if (H.n_seg > S.cap_segs || n_reads > S.cap_reads) {
        // (the small arrays of the slot were sized for fewer segments / reads: grow them; the columns stay)
        mc_set_error("mc_ctx_parse_finish: %d segments, %d reads: the slot holds %lld, %lld (mc_ctx_reserve_tables)", H.n_seg, n_reads,
                     (long long)S.cap_segs, (long long)S.cap_reads);
        return -12;
    }
    HIP_TRY(hipEventSynchronize(S.ev_uploaded));           // the stage is about to be rewritten (long done: the slot was idle)
    S.kp_state = 0;
    S.refs = std::max(S.refs - 1, 0);
    if (int rc = fill_slot(c, slot, H.n_rows, H.n_seg, S.kp_seg_row.data(), seg_read, S.kp_seg_contig.data(), S.kp_seg_ns.data(), n_reads,
                           read_qual, nullptr))
        return rc;
    return 0;
}

extern "C" int mc_ctx_parse_abandon(mc_ctx *c, int32_t slot) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0 || slot >= MC_TABLE_SLOTS || c->slots[slot].kp_state == 0) {
        mc_set_error("mc_ctx_parse_abandon: slot %d holds no parse", slot);
        return -12;
    }
    TableSlot &S = c->slots[slot];
    HIP_TRY(hipEventSynchronize(S.ev_parsed));
    // (ev_valid still stands for the slot's previous table, which is all a later upload waits for)
    S.kp_state = 0;
    S.refs = std::max(S.refs - 1, 0);
    return 0;
}

// the columns of a slot's table back on the host (tests: the device parser's columns against the host parser's)
extern "C" int mc_ctx_fetch_columns(mc_ctx *c, int32_t slot, int64_t n_rows, int32_t *pos, int32_t *event_model_e4, int32_t *event_idx,
                                    uint8_t *flags) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0 || slot >= MC_TABLE_SLOTS || !c->slots[slot].pos || n_rows < 0 || n_rows > c->slots[slot].cap_rows) {
        mc_set_error("mc_ctx_fetch_columns: slot %d, %lld rows", slot, (long long)n_rows);
        return -12;
    }
    TableSlot &S = c->slots[slot];
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    if (c->parse_stream) HIP_TRY(hipStreamSynchronize(c->parse_stream));
    if (n_rows == 0) return 0;
    if (pos) HIP_TRY(hipMemcpy(pos, S.pos, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
    if (event_model_e4) HIP_TRY(hipMemcpy(event_model_e4, S.evmu, (size_t)n_rows * 8, hipMemcpyDeviceToHost));
    if (event_idx) HIP_TRY(hipMemcpy(event_idx, S.idx, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
    if (flags) HIP_TRY(hipMemcpy(flags, S.flags, (size_t)n_rows, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mc_ctx_wait_upload(mc_ctx *c, int32_t slot) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0 || slot >= MC_TABLE_SLOTS || !c->slots[slot].ev_uploaded) {
        mc_set_error("mc_ctx_wait_upload: slot %d", slot);
        return -12;
    }
    HIP_TRY(hipEventSynchronize(c->slots[slot].ev_uploaded));
    return 0;
}

extern "C" int mc_ctx_current_slot(mc_ctx *c) { return c->cur; }

// A resident table becomes the current one again (the passes enqueued afterwards scan it).  as_new != 0: what earlier passes
// left behind for later ones is set aside -- the next pass does everything the first pass over a table does (classification on
// the blocks' first rows, positions and event indices streamed, every row validated).
extern "C" int mc_ctx_select_table(mc_ctx *c, int32_t slot, int32_t as_new) {
    HIP_TRY(hipSetDevice(c->device));
    // (holds_table: set when a table's small arrays went in, fill_slot; cleared when a parse began to overwrite the columns -- a
    // parse that was abandoned, or handed out and never finished, leaves columns that S.T does not describe)
    if (slot < 0 || slot >= MC_TABLE_SLOTS || !c->slots[slot].T.pos || c->slots[slot].kp_state != 0 || !c->slots[slot].holds_table) {
        mc_set_error("mc_ctx_select_table: slot %d holds no complete table", slot);
        return -12;
    }
    TableSlot &S = c->slots[slot];
    c->T = S.T;
    c->cur = slot;
    if (S.qual) { c->qual = S.qual; c->n_qual = S.n_qual; }
    else { c->qual = c->qual_own; c->n_qual = c->n_qual_own; }
    // (passes over the slot that are still in flight keep the plan they were enqueued with; a first pass only ORs what it sees
    // into the table's validation flags, so declaring the table new beside them is safe as long as they are first passes too --
    // bench.py's steps -- and a caller that mixes pass kinds waits for them first)
    if (as_new) {
        S.passes = 0;
        S.tmpl_ref = -1;          // (the name-block templates too: they are part of what a table costs when it is scanned once)
    }
    return 0;
}

extern "C" int mc_ctx_upload_times_ms(mc_ctx *c, int32_t slot, float *h2d_ms, float *validate_ms) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0 || slot >= MC_TABLE_SLOTS || !c->slots[slot].ev_uploaded) {
        mc_set_error("mc_ctx_upload_times_ms: slot %d", slot);
        return -12;
    }
    TableSlot &S = c->slots[slot];
    HIP_TRY(hipEventSynchronize(S.ev_valid));
    if (h2d_ms) HIP_TRY(hipEventElapsedTime(h2d_ms, S.ev_up_start, S.ev_uploaded));
    if (validate_ms) HIP_TRY(hipEventElapsedTime(validate_ms, S.ev_val_start, S.ev_valid));
    return 0;
}

extern "C" int mc_ctx_parse_times_ms(mc_ctx *c, int32_t slot, float *text_h2d_ms, float *parse_ms) {
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0 || slot >= MC_TABLE_SLOTS || c->slots[slot].kp_state == 0 || !c->slots[slot].ev_parsed) {
        mc_set_error("mc_ctx_parse_times_ms: slot %d holds no parse", slot);
        return -12;
    }
    TableSlot &S = c->slots[slot];
    HIP_TRY(hipEventSynchronize(S.ev_parsed));
    if (text_h2d_ms) HIP_TRY(hipEventElapsedTime(text_h2d_ms, S.ev_up_start, S.ev_text_up));
    if (parse_ms) HIP_TRY(hipEventElapsedTime(parse_ms, S.ev_text_up, S.ev_parsed));
    return 0;
}

extern "C" int mc_ctx_upload_table(mc_ctx *c, const mc_table_view *h) {
    HIP_TRY(hipSetDevice(c->device));
    // the one-table interface: whatever is in flight finishes first, so the caller's buffers are free on return and the
    // slot that is taken over holds nothing anybody waits for
    if (int rc = sync_pass_streams(c)) return rc;
    if (c->ab_count == 0) {                                  // no pass to hand out any more: nothing is held
        c->held = -1;
        for (TableSlot &S : c->slots) S.refs = S.kp_state != 0 ? 1 : 0;      // (but a slot the device parser is filling stays taken)
    }
    int32_t slot = -1;
    if (int rc = mc_ctx_upload_table_async(c, h, nullptr, &slot)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mc_ctx_set_read_quality(mc_ctx *c, const double *qual, int32_t n_reads) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_pass_streams(c)) return rc;            // passes in flight read the old buffer
    if (c->qual_own) (void)hipFree(c->qual_own);
    c->qual_own = nullptr;
    HIP_TRY(hipMalloc((void **)&c->qual_own, std::max<size_t>((size_t)n_reads * 8, 256)));
    if (n_reads > 0) HIP_TRY(hipMemcpy(c->qual_own, qual, (size_t)n_reads * 8, hipMemcpyHostToDevice));
    c->qual = c->qual_own;
    c->n_qual = c->n_qual_own = n_reads;
    return 0;
}

extern "C" int mc_ctx_set_mlp(mc_ctx *c, int32_t n_models, int32_t n_in, int32_t n_hidden, const double *W1,
                              const double *b1, const double *W2, const double *b2, const uint8_t *sub_of_char) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_pass_streams(c)) return rc;            // (passes in flight score with the old one, on the side stream)
    if (n_in < 1 || n_in > MC_MAX_K + 1 || n_models < 1 || n_hidden < 1) {
        mc_set_error("unsupported MLP shape: %d models, %d inputs, %d hidden", n_models, n_in, n_hidden);
        return -12;
    }
    if (n_models > K2_MAXM) {
        mc_set_error("MLP with %d sub-models: k2_mlp lists at most %d", n_models, K2_MAXM);
        return -12;
    }
    free_pool(c->mlp_allocs);
    free_pool(c->forest_allocs);
    free_pool(c->simple_allocs);
    c->F = DevForest();
    c->Sc = DevSimple();
    DevMlp &M = c->M;
    M.n_models = n_models;
    M.n_in = n_in;
    M.n_hidden = n_hidden;
    UP(M.W1, W1, (size_t)n_models * n_in * n_hidden, c->mlp_allocs);
    UP(M.b1, b1, (size_t)n_models * n_hidden, c->mlp_allocs);
    UP(M.W2, W2, (size_t)n_models * n_hidden, c->mlp_allocs);
    UP(M.b2, b2, (size_t)n_models, c->mlp_allocs);
    // unit by unit: the n_in weights into hidden unit j, its bias, its output weight (alive until the copy has been waited for)
    const size_t S = (size_t)n_in + 2;
    std::vector<double> wu((size_t)n_models * n_hidden * S);
    for (int m = 0; m < n_models; ++m)
        for (int j = 0; j < n_hidden; ++j) {
            double *u = &wu[((size_t)m * n_hidden + j) * S];
            for (int i = 0; i < n_in; ++i) u[i] = W1[((size_t)m * n_in + i) * n_hidden + j];
            u[n_in] = b1[(size_t)m * n_hidden + j];
            u[n_in + 1] = W2[(size_t)m * n_hidden + j];
        }
    UP(M.wu, wu.data(), wu.size(), c->mlp_allocs);
    // The fast forward (k2_mlp<.., true>): the same weights as floats, and per sub-model how far its probability can lie from
    // the fp64 one, as K0 + sum_i K_i |x_i|.  With u = 2^-24: a hidden unit's input a_j is off by at most g_dot sum_i |x_i W1_ij|
    // (+ the bias), g_dot = (n_in + 5) u (n_in products and sums, the inputs and weights rounded to float -- the weights after the factor
    // 2 log2(e) that makes the sum the exponent of tanh32s: a relative error either way); tanh is 1-Lipschitz and tanh32s is within
    // K2_TANH32_MAX_ERR of it; the output sum picks up g_acc sum_j |W2_j| (a quarter of the units per partial sum, two chains of
    // at most 13 terms and their sum, the products, the weights' rounding: g_acc = 29 u covers 25 in one chain); the logistic
    // function's slope is at most 1/4.  Five per cent on top for the float arithmetic the bound itself is evaluated in.
    const int h2 = (n_hidden + 1) / 2;
    std::vector<float> wp32((size_t)n_models * h2 * S * 2, 0.0f), margin((size_t)n_models * (MC_MAX_K + 2), 0.0f);    // (alive until the copies have been waited for)
    {
        const double c2 = 2.8853900817779268;         // 2 log2(e): the hidden unit's sum is the exponent of tanh32s
        for (int m = 0; m < n_models; ++m)
            for (int j = 0; j < n_hidden; ++j) {
                const double *uj = &wu[((size_t)m * n_hidden + j) * S];
                float *pj = &wp32[(((size_t)m * h2 + j / 2) * S) * 2 + (j & 1)];
                for (int i = 0; i <= n_in; ++i) pj[2 * i] = (float)(c2 * uj[i]);
                pj[2 * (n_in + 1)] = (float)uj[n_in + 1];
            }
        const double u = 5.9604644775390625e-08, g_dot = (n_in + 5.0) * u, g_acc = 29.0 * u;
        for (int m = 0; m < n_models; ++m) {
            double sw2 = 0.0, cb = 0.0, ci[MC_MAX_K + 1] = {0};
            for (int j = 0; j < n_hidden; ++j) {
                const double *uj = &wu[((size_t)m * n_hidden + j) * S];
                const double w2 = std::fabs(uj[n_in + 1]);
                sw2 += w2;
                cb += w2 * std::fabs(uj[n_in]);
                for (int i = 0; i < n_in; ++i) ci[i] += w2 * std::fabs(uj[i]);
            }
            float *mg = &margin[(size_t)m * (MC_MAX_K + 2)];
            mg[0] = (float)(1.05 * 0.25 * (sw2 * K2_TANH32_MAX_ERR + g_dot * cb + g_acc * sw2) + 1e-12);
            for (int i = 0; i < n_in; ++i) mg[1 + i] = (float)(1.05 * 0.25 * g_dot * ci[i]);
        }
        UP(M.wp32, wp32.data(), wp32.size(), c->mlp_allocs);
        UP(M.margin, margin.data(), margin.size(), c->mlp_allocs);
    }
    // (MCALLER_MLP_FP64=1: every record in fp64, as rounds 1-4 scored them)
    M.fast = (getenv("MCALLER_MLP_FP64") && atoi(getenv("MCALLER_MLP_FP64")) != 0) ? 0 : 1;
    UP(M.sub_of_char, sub_of_char, 256, c->mlp_allocs);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mc_ctx_set_forest(mc_ctx *c, int32_t n_models, int32_t n_in, const int32_t *model_tree_off,
                                 const int32_t *tree_node_off, const int32_t *left, const int32_t *right,
                                 const int32_t *feature, const double *threshold, const double *value,
                                 const uint8_t *sub_of_char) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_pass_streams(c)) return rc;            // (passes in flight score with the old one, on the side stream)
    if (n_models < 1 || n_in < 1 || n_in > MC_MAX_K + 1) {
        mc_set_error("unsupported forest shape: %d models, %d inputs", n_models, n_in);
        return -12;
    }
    const int n_trees = model_tree_off[n_models];
    const int n_nodes = tree_node_off[n_trees];
    for (int i = 0; i < n_nodes; ++i)
        if (left[i] >= 0 && (feature[i] < 0 || feature[i] >= n_in || left[i] >= n_nodes || right[i] < 0 || right[i] >= n_nodes)) {
            mc_set_error("forest node %d is malformed", i);
            return -12;
        }
    free_pool(c->forest_allocs);
    free_pool(c->mlp_allocs);
    free_pool(c->simple_allocs);
    c->M = DevMlp();
    c->Sc = DevSimple();
    DevForest &F = c->F;
    F.n_models = n_models;
    F.n_in = n_in;
    UP(F.model_tree_off, model_tree_off, (size_t)n_models + 1, c->forest_allocs);
    UP(F.tree_node_off, tree_node_off, (size_t)n_trees + 1, c->forest_allocs);
    UP(F.left, left, (size_t)n_nodes, c->forest_allocs);
    UP(F.right, right, (size_t)n_nodes, c->forest_allocs);
    UP(F.feature, feature, (size_t)n_nodes, c->forest_allocs);
    UP(F.threshold, threshold, (size_t)n_nodes, c->forest_allocs);
    UP(F.value, value, (size_t)n_nodes * 2, c->forest_allocs);
    UP(F.sub_of_char, sub_of_char, 256, c->forest_allocs);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mc_ctx_set_simple_classifier(mc_ctx *c, int32_t kind, int32_t n_models, int32_t n_in, const double *params,
                                            int32_t stride, const uint8_t *sub_of_char) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_pass_streams(c)) return rc;            // (passes in flight score with the old one, on the side stream)
    const int want = kind == MC_CLF_LOGISTIC ? n_in + 1 : (kind == MC_CLF_GNB ? 4 * n_in + 2 : -1);
    if (n_models < 1 || n_in < 1 || n_in > MC_MAX_K + 1 || stride != want) {
        mc_set_error("unsupported classifier: kind %d, %d models, %d inputs, %d parameters each", kind, n_models, n_in, stride);
        return -12;
    }
    if (kind == MC_CLF_GNB)
        for (int m = 0; m < n_models; ++m)
            for (int cls = 0; cls < 2; ++cls)
                for (int i = 0; i < n_in; ++i)
                    if (!(params[(size_t)m * stride + (size_t)cls * 2 * n_in + n_in + i] > 0.0)) {
                        mc_set_error("naive Bayes model %d: variance %d of class %d is not positive", m, i, cls);
                        return -12;
                    }
    free_pool(c->forest_allocs);
    free_pool(c->mlp_allocs);
    free_pool(c->simple_allocs);
    c->M = DevMlp();
    c->F = DevForest();
    DevSimple &S = c->Sc;
    S = DevSimple();
    UP(S.params, params, (size_t)n_models * stride, c->simple_allocs);
    UP(S.sub_of_char, sub_of_char, 256, c->simple_allocs);
    S.kind = kind; S.n_models = n_models; S.n_in = n_in; S.stride = stride;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

static int classifier_inputs(const mc_ctx *c) {
    return c->F.left ? c->F.n_in : (c->Sc.params ? c->Sc.n_in : (c->M.W1 ? c->M.n_in : 0));
}
// the classifier of the context over n records (mc_classify.hip)
static void launch_classifier(mc_ctx *c, hipStream_t st, const double *feats, int k, const int32_t *site_seg, const int32_t *seg_read,
                              const double *qual, const uint32_t *info, const uint8_t *submodel_in, int64_t n, double *prob,
                              const unsigned long long *n_dev, const unsigned int *overflow, const int32_t *piece_cnt = nullptr,
                              int piece_room = 0, int64_t n_pieces = 0) {
    mc_launch_classifier(c->M, c->F, c->Sc, c->n_cu, st, feats, k, site_seg, seg_read, qual, info, submodel_in, n, prob, n_dev, overflow,
                         piece_cnt, piece_room, n_pieces);
}

// marked positions are dense (a one-base motif): the scan instance that lists every unit of a tile, bigger payload chunks
static bool dense_reference(const mc_ctx *c) {
    return c->ref_total_len > 0 && (double)c->R.n_sites * 64.0 > (double)c->ref_total_len;
}

static int ensure_records(mc_ctx *c, int64_t cap, int k) {
    const int64_t need_tiles = std::max<int64_t>(c->T.n_tiles, c->scratch_tiles);
    const int chunk = dense_reference(c) ? 256 : 64;
    if (c->Omain.capacity >= cap && c->last_k == k && c->payload_tiles >= need_tiles && c->payload_chunk >= chunk) { c->O = c->Omain; return 0; }
    if (int rc = sync_pass_streams(c)) return rc;
    free_pool(c->rec_allocs);
    for (DevRecords *D : {&c->Omain}) {
        D->capacity = cap;
        if (dev_alloc(c->rec_allocs, &D->feats, (size_t)cap * k) || dev_alloc(c->rec_allocs, &D->site_pos, (size_t)cap) ||
            dev_alloc(c->rec_allocs, &D->site_seg, (size_t)cap) || dev_alloc(c->rec_allocs, &D->close_row, (size_t)cap) ||
            dev_alloc(c->rec_allocs, &D->info, (size_t)cap) || dev_alloc(c->rec_allocs, &D->prob, (size_t)cap) ||
            dev_alloc(c->rec_allocs, &D->wmask, (size_t)cap))
            return -10;
    }
    c->payload_tiles = need_tiles;
    // (the scan hands out payload slots beyond a tile's own PT in chunks; every tile may leave most of its last chunk unused)
    c->payload_cap = cap + (need_tiles + 1) * (PT + chunk);
    c->payload_chunk = chunk;
    if (dev_alloc(c->rec_allocs, &c->payload_sorted, (size_t)cap) || dev_alloc(c->rec_allocs, &c->rare_list, (size_t)cap) || dev_alloc(c->rec_allocs, &c->payload, (size_t)c->payload_cap)) return -10;
    c->last_k = k;
    c->O = c->Omain;
    return 0;
}

static int alloc_records(std::vector<void *> &pool, DevRecords &D, int64_t cap, int k) {
    D.capacity = cap;
    if (dev_alloc(pool, &D.feats, (size_t)cap * k) || dev_alloc(pool, &D.site_pos, (size_t)cap) ||
        dev_alloc(pool, &D.site_seg, (size_t)cap) || dev_alloc(pool, &D.close_row, (size_t)cap) ||
        dev_alloc(pool, &D.info, (size_t)cap) || dev_alloc(pool, &D.prob, (size_t)cap) || dev_alloc(pool, &D.wmask, (size_t)cap))
        return -10;
    return 0;
}

// Irregular name blocks: literal row-by-row machine on the GPU, then merge with the fast path's records.
static int run_literal_path(mc_ctx *c, const mc_params *prm, int64_t *n_io) {
    const DevTable &T = c->T;
    const int k = prm->k;
    const int n_groups = (T.n_nb + GROUP - 1) / GROUP;
    LitArgs LA;
    LA.T = T; LA.R = c->R; LA.desc = c->desc; LA.nb_f0 = c->nb_f0; LA.qual = c->qual; LA.qual_thresh = prm->qual_thresh;
    LA.k = k; LA.skip_thresh = prm->skip_thresh; LA.tail_contig = prm->tail_contig; LA.entry_read = prm->entry_read;
    LA.entry_first_idx = prm->entry_first_idx;
    int32_t *run_cnt, *run_rows, *cnt_local, *rows_local;
    int64_t *cnt_group, *rows_group;
    std::vector<void *> &P = c->lit_allocs;
    if (dev_alloc(P, &run_cnt, (size_t)T.n_nb + 1) || dev_alloc(P, &run_rows, (size_t)T.n_nb + 1) ||
        dev_alloc(P, &cnt_local, (size_t)T.n_nb + 1) || dev_alloc(P, &rows_local, (size_t)T.n_nb + 1) ||
        dev_alloc(P, &cnt_group, (size_t)n_groups + 1) || dev_alloc(P, &rows_group, (size_t)n_groups + 1))
        return -10;
    LA.run_cnt = run_cnt; LA.run_rows = run_rows; LA.cnt_local = cnt_local; LA.cnt_group = cnt_group;
    LA.rows_local = rows_local; LA.rows_group = rows_group; LA.scratch = nullptr; LA.L = DevRecords(); LA.write = 0;
    const unsigned g = (unsigned)((T.n_nb + 63) / 64);
    mc_launch_literal(LA, g, c->stream);
    mc_launch_group_scan(run_cnt, (int64_t)T.n_nb, cnt_local, cnt_group, c->stream);
    mc_launch_group_scan(run_rows, (int64_t)T.n_nb, rows_local, rows_group, c->stream);
    std::vector<int64_t> hc((size_t)n_groups), hr((size_t)n_groups);
    HIP_TRY(hipMemcpyAsync(hc.data(), cnt_group, (size_t)n_groups * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(hr.data(), rows_group, (size_t)n_groups * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    int64_t n_lit = 0, n_rows_lit = 0;
    for (int i = 0; i < n_groups; ++i) { n_lit += hc[(size_t)i]; n_rows_lit += hr[(size_t)i]; }
    if (n_lit == 0) return 0;
    DevRecords L, M;
    double *scratch;
    const int64_t n_fast = *n_io;
    if (alloc_records(P, L, n_lit, k) || alloc_records(P, M, n_fast + n_lit, k) ||
        dev_alloc(P, &scratch, (size_t)std::max<int64_t>(n_rows_lit, 1) * MC_MAX_K))
        return -10;
    LA.scratch = scratch; LA.L = L; LA.write = 1;
    mc_launch_literal(LA, g, c->stream);
    mc_launch_merge(c->O, n_fast, L, n_lit, M, k, c->stream);
    HIP_TRY(hipGetLastError());
    c->O = M;
    *n_io = n_fast + n_lit;
    return 0;
}

// How a pass goes about its table: decided when it is enqueued, from what the passes before it have left (TableSlot.passes).
struct PassPlan {
    bool first;          // no pass has validated the table: classification on the blocks' first rows, the scan validates every row
    int scan_mode;       // SCAN_*
};
static PassPlan plan_pass(mc_ctx *c, hipStream_t st) {
    PassPlan P;
    P.first = true;
    P.scan_mode = SCAN_VALIDATE;
    if (c->cur < 0) return P;
    TableSlot &S = c->slots[c->cur];
    const bool dense = dense_reference(c);                 // (a one-base motif: every unit is listed, summaries would not help)
    if (S.passes > 0) {
        P.first = false;
        P.scan_mode = SCAN_STREAM;
        if (!dense && S.passes >= 2) {
            if (!S.summarized && S.T.n_rows > 0) {
                mc_launch_summarize(S.T, st);
                S.summarized = true;
            }
            P.scan_mode = SCAN_SUMMARY;
        }
    }
    S.passes += 1;
    return P;
}

// K0 (strand resolve) of one pass on stream st: counters zeroed, first site rows, classification.
// extend: also widen the irregular set (k0_extend) -- what the literal path of the synchronous pass needs; a pipelined pass
// with an irregular block is thrown away and re-run synchronously, so it never looks at the result.
static int enqueue_k0(mc_ctx *c, const mc_params *prm, const K0Set &K, Counters *cnt, hipStream_t st, unsigned long long pass_no,
                      bool extend, const PassPlan &plan) {
    const DevTable &T = c->T;
    const int k = prm->k;
    const bool lookback = T.has_repeats || prm->entry_read >= 0;      // a block may see name == last_read (:161)
    // (the name-block templates: once per (table, reference), made by the first pass's k0_first_site itself)
    const bool make_tmpl = c->cur >= 0 && c->slots[c->cur].tmpl_ref != c->ref_version;
    if (c->cur >= 0) c->slots[c->cur].tmpl_ref = c->ref_version;
    mc_launch_first_site(T, c->R, c->qual, prm->qual_thresh, k, K.desc, K.nb_f0, cnt, lookback ? 0 : 1, prm->skip_thresh, pass_no,
                         plan.first ? 1 : 0, make_tmpl ? 1 : 0, st);
    if (lookback) mc_launch_classify(T, c->R, K.desc, K.nb_f0, prm->entry_read, k, prm->skip_thresh, cnt, pass_no, st);
    if (extend) mc_launch_extend(T, K.desc, K.nb_f0, prm->entry_read, cnt, pass_no, st);
    return 0;
}

// K1 (scan, order, emit) of one pass into the record set O on stream st; ev_scan_end is recorded after the scan.
static int enqueue_k1(mc_ctx *c, const mc_params *prm, const K0Set &K, Counters *cnt, const DevRecords &O, hipStream_t st,
                      hipEvent_t ev_scan_end, K1Args *out_args, Payload *sorted, int64_t *rare_list, unsigned long long pass_no,
                      const PassPlan &plan, hipEvent_t ev_emit_end = nullptr, unsigned long long *chunk_cnt = nullptr, int fused_room = 0,
                      int32_t *piece_cnt = nullptr, int32_t *piece_kw = nullptr, hipStream_t emit_st = nullptr, hipEvent_t ev_list_end = nullptr,
                      hipEvent_t ev_emit_start = nullptr, bool *emit_went_aside = nullptr) {
    const DevTable &T = c->T;
    if (emit_went_aside) *emit_went_aside = false;
    K1Args A;
    A.T = T; A.R = c->R; A.desc = K.desc; A.tile_chunk = c->tile_chunk; A.payload = c->payload;
    A.payload_cap = c->payload_cap; A.tile_cnt = c->tile_cnt; A.tile_half = c->tile_half;
    A.tile_local = c->tile_local; A.group_sum = c->group_sum; A.tile_first = c->tile_first; A.O = O; A.cnt = cnt; A.k = prm->k;
    A.skip_thresh = prm->skip_thresh; A.tail_contig = prm->tail_contig; A.rare_list = rare_list;
    A.pass_no = pass_no;
    A.piece_cnt = piece_cnt;
    static const bool no_kw = getenv("MCALLER_NO_PIECE_KW") != nullptr;                 // (probe: what the per-piece counts cost k1_fused; the three kernels pack)
    A.piece_kw = (fused_room > 0 && fused_room < MC_SIDE_MAX_ROOM && !no_kw) ? piece_kw : nullptr;
    const bool dense = dense_reference(c);
    static const bool no_runs = getenv("MCALLER_NO_EMIT_RUNS") != nullptr;         // (tests: the eight-lane emit on a dense reference)
    const bool runs = dense && !no_runs;
    // (k1_emit and the row-by-row kernel count the packing's chunks as they write the records; the run-table emit of a dense
    // reference does not: k_pack_count goes over its records)
    A.chunk_cnt = runs ? nullptr : chunk_cnt;
    A.chunk_shift = dense ? 8 : 6;
    A.shard_shift = T.n_tiles >= 1024 ? 6 : 3;
    A.shard_mask = (1 << A.shard_shift) - 1;
    static_assert(NSHARD == 64, "shard_shift");
    if (fused_room > 0) {
        // a dense reference, a pipelined pass: scan, ordering and emit as ONE kernel with fixed room per piece (k1_fused,
        // mc_fused.hip); the pass's event rides on its dispatch packet
        hipEvent_t on_packet = (ev_emit_end && MC_EVENTS_ON_KERNELS) ? ev_emit_end : nullptr;
        mc_launch_fused(A, sorted, fused_room, plan.scan_mode == SCAN_VALIDATE, st, on_packet);
        if (ev_scan_end) HIP_TRY(hipEventRecord(ev_scan_end, st));
        if (ev_emit_end && !on_packet) HIP_TRY(hipEventRecord(ev_emit_end, st));
        *out_args = A;
        return 0;
    }
    mc_launch_scan(A, dense, plan.scan_mode, st);
    if (ev_scan_end) HIP_TRY(hipEventRecord(ev_scan_end, st));
    mc_launch_group_scan(c->tile_cnt, T.n_tiles, c->tile_local, c->group_sum, st);
    // (dense references: a workgroup per piece, the mean of every position once, see k1_emit_runs -- which takes the payloads where
    // the scan left them: no gather)
    // The eight-lane emit of a pipelined pass goes to the SIDE stream, in front of the pass's classifier and packing: it reads nothing but
    // the pass's own buffers (its sorted payloads, descriptors, records, counters) and the table, is bound by latency (three dependent
    // round trips: 39 us with a few thousand waves), and on the ctx stream the next pass's strand resolve and scan waited behind it
    const bool emit_aside = emit_st != nullptr && ev_list_end != nullptr && !runs;
    mc_launch_list(A, sorted, runs ? 0 : 1, st, (emit_aside && MC_EVENTS_ON_KERNELS) ? ev_list_end : nullptr);
    // (ev_emit_end rides on the emit's own dispatch packet: a hipEventRecord behind it is a barrier packet of its own and
    // costs the queue 5-9 us)
    const unsigned emit_grid = (unsigned)std::min<int64_t>((O.capacity * EG + 255) / 256, (int64_t)c->n_cu * c->emit_wgs);
    hipEvent_t on_packet = (ev_emit_end && MC_EVENTS_ON_KERNELS) ? ev_emit_end : nullptr;
    hipStream_t est = st;
    if (emit_aside) {
        if (!MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(ev_list_end, st));
        HIP_TRY(hipStreamWaitEvent(emit_st, ev_list_end, 0));
        est = emit_st;
    }
    if (emit_went_aside) *emit_went_aside = emit_aside;
    if (runs) mc_launch_emit_runs(A, sorted, st, on_packet);
    else mc_launch_emit(A, sorted, emit_grid, est, on_packet, (emit_aside && on_packet) ? ev_emit_start : nullptr);
    if (ev_emit_end && !on_packet) HIP_TRY(hipEventRecord(ev_emit_end, est));
    *out_args = A;
    return 0;
}

// the synchronous pass: everything on the ctx stream, ev[0..3] around the stages (mc_last_times_ms)
static int enqueue_fast_path(mc_ctx *c, const mc_params *prm, const DevRecords &O, hipEvent_t *ev, K1Args *out_args) {
    K0Set K;
    K.desc = c->desc; K.nb_f0 = c->nb_f0;
    const PassPlan plan = plan_pass(c, c->stream);            // (in front of ev[0]: a table's summaries are not part of a pass)
    HIP_TRY(hipEventRecord(ev[0], c->stream));
    if (int rc = enqueue_k0(c, prm, K, c->cnt, c->stream, c->sync_pass_no, true, plan)) return rc;
    HIP_TRY(hipEventRecord(ev[1], c->stream));
    if (int rc = enqueue_k1(c, prm, K, c->cnt, O, c->stream, ev[2], out_args, c->payload_sorted, c->rare_list, c->sync_pass_no, plan)) return rc;
    HIP_TRY(hipEventRecord(ev[3], c->stream));
    return 0;
}

// Record capacity to start with: a window closes about once per marked site a read covers -- rows x (sites per strand
// position) x ~0.52 positions per row -- with 50 % head room, and never less than one per 64 rows (GATC in a random
// genome: one per ~490 rows).  A pass that overflows it is repeated with what it actually needed.
static int64_t guess_capacity(const mc_ctx *c) {
    if (const char *e = getenv("MCALLER_RECORD_CAPACITY")) { if (atoll(e) > 0) return atoll(e); }   // (tests: force the overflow path)
    const double density = c->ref_total_len > 0 ? (double)c->R.n_sites / (2.0 * (double)c->ref_total_len) : 0.0;
    const int64_t rows = std::max<int64_t>(c->T.n_rows, c->res_rows);        // (reserved: every later table fits, no re-allocation)
    const int64_t by_sites = (int64_t)((double)rows * density * 0.52 * 1.5);
    return std::max<int64_t>(1 << 16, std::max<int64_t>(rows / 64, by_sites) + 4096);
}

// what every pass needs before it can be enqueued
static int check_pass(mc_ctx *c, const mc_params *prm) {
    const DevTable &T = c->T;
    const int k = prm->k;
    if (k < 1 || k > MC_MAX_K) {
        mc_set_error("num_variables %d not supported (1..%d)", k, MC_MAX_K);
        return -12;
    }
    if (!T.pos || !c->R.mf || !c->qual) {
        mc_set_error("mc_extract_features: table, reference and read qualities must be set first");
        return -12;
    }
    if (c->n_qual < T.n_reads) {
        mc_set_error("read quality table has %d entries, table names %d reads", c->n_qual, T.n_reads);
        return -12;
    }
    const int clf_in = classifier_inputs(c);
    if (prm->score && clf_in != k + 1) {
        mc_set_error("classifier expects %d inputs but num_variables+1 = %d", clf_in, k + 1);
        return -12;
    }
    return 0;
}

extern "C" int mc_extract_features(mc_ctx *c, const mc_params *prm, int64_t *n_records) {
    HIP_TRY(hipSetDevice(c->device));
    *n_records = 0;
    const DevTable &T = c->T;
    const int k = prm->k;
    if (int rc = check_pass(c, prm)) return rc;
    c->last_n = 0;
    c->last_slots = 0;
    if (T.n_rows == 0 || T.n_nb == 0) return 0;
    if (c->ab_count) { if (int rc = sync_pass_streams(c)) return rc; }   // pipelined passes share the scratch: let them finish
    if (int rc = ensure_scratch(c, T.n_nb, T.n_tiles)) return rc;
    c->last_T = T;
    if (!c->in_rerun) c->held = c->cur;                                   // (a re-run inside mc_wait_records: held by the caller)

    free_pool(c->lit_allocs);
    int64_t cap = std::max<int64_t>(guess_capacity(c), c->Omain.capacity);
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (int rc = ensure_records(c, cap, k)) return rc;
        K1Args A;
        c->sync_pass_no = ++c->pass_counter;
        if (int rc = enqueue_fast_path(c, prm, c->O, c->ev, &A)) return rc;
        Counters h;
        HIP_TRY(hipMemcpyAsync(&h, c->cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipGetLastError());
        int64_t n = (int64_t)h.n_records;
        if (h.overflow) {                   // the buffers were a guess; the exact need is known now (+ shard skew)
            cap = std::max<int64_t>(cap * 2, n + n / 4 + 4096);
            continue;
        }
        // the table's first pass, and a row contradicts what a block was classified on (its first rows): the validation flags
        // are complete now, the next attempt classifies on them
        if (h.violation) continue;
        if (h.n_rare) {
            mc_launch_rare(A, c->payload_sorted, c->rare_list, (int64_t)h.n_rare, c->stream);
        }
        if (h.n_big && n > 0) mc_launch_bigfix(A, n, c->stream);
        const bool irregular = h.irregular_pass == c->sync_pass_no;
        if (irregular) {
            if (int rc = run_literal_path(c, prm, &n)) return rc;
        }
        // records -> pinned host memory; the slot means and indices travel while the classifier runs
        if (int rc = ensure_pinned(c, n, k)) return rc;
        const bool early = n > 0 && !h.n_big && !h.n_rare && !irregular;      // (nothing on the ctx stream still writes records)
        if (early) { if (int rc = copy_out_features(c, n, k, c->copy_stream)) return rc; }
        if (prm->score && n > 0)
            launch_classifier(c, c->stream, c->O.feats, k, c->O.site_seg, T.seg_read, c->qual, c->O.info, (const uint8_t *)nullptr, n,
                              c->O.prob, (const unsigned long long *)nullptr, (const unsigned int *)nullptr);
        HIP_TRY(hipEventRecord(c->ev[4], c->stream));
        if (n > 0) {
            if (!early) { if (int rc = copy_out_features(c, n, k, c->stream)) return rc; }
            HIP_TRY(hipMemcpyAsync(c->H.prob, c->O.prob, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->copy_stream));
        HIP_TRY(hipGetLastError());
        for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(&c->times[i], c->ev[i], c->ev[i + 1]));
        HIP_TRY(hipEventElapsedTime(&c->times[4], c->ev[0], c->ev[4]));
        c->last_n = n;
        c->last_slots = n;
        *n_records = n;
        return 0;
    }
    mc_set_error("record buffer overflow after 4 attempts");
    return -13;
}

extern "C" int mc_fetch_records(mc_ctx *c, const mc_calls_view *out) {
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = c->last_n;
    const int k = c->last_k;
    if (out->capacity < n) {
        mc_set_error("mc_fetch_records: capacity %lld < %lld records", (long long)out->capacity, (long long)n);
        return -12;
    }
    if (n == 0) return 0;
    memcpy(out->feats, c->H.feats, (size_t)n * k * 8);
    memcpy(out->site_pos, c->H.site_pos, (size_t)n * 4);
    memcpy(out->site_seg, c->H.site_seg, (size_t)n * 4);
    memcpy(out->close_row, c->H.close_row, (size_t)n * 8);
    memcpy(out->info, c->H.info, (size_t)n * 4);
    memcpy(out->prob, c->H.prob, (size_t)n * 8);
    return 0;
}

extern "C" int mc_fetch_records_view(mc_ctx *c, mc_calls_view *out) {
    out->capacity = c->last_n;
    out->feats = c->H.feats;
    out->site_pos = c->H.site_pos;
    out->site_seg = c->H.site_seg;
    out->close_row = c->H.close_row;
    out->info = c->H.info;
    out->prob = c->H.prob;
    out->call_row = nullptr;          // means and probabilities are stored for every record here
    out->n_call_rows = 0;
    out->close_row32 = nullptr;
    out->compacted = 0;
    out->feats_lo32 = nullptr; out->feats_hi32 = nullptr; out->feats_wide = nullptr; out->n_wide = 0;
    return 0;
}

// ---- pipelined passes ----
// A pass computes on the ctx stream (K0, K1, K2, packing, back to back with the next pass) and is copied out on
// copy_stream when it is waited for.
static void free_async(mc_ctx *c) {
    for (auto &b : c->ab) {
        free_pool(b.dev_allocs);
        b.H = DevRecords();
        if (b.pack_host) (void)hipHostFree(b.pack_host);
        if (b.st_host) (void)hipHostFree(b.st_host);
        b.st_host = nullptr; b.cnt = nullptr; b.pack = nullptr; b.pack_host = nullptr;
        b.O = DevRecords();
        b.K = K0Set();
        free_pool(b.k0_allocs);
        b.cap = b.n_nb = 0; b.k = 0; b.pack_bytes = 0; b.used = false; b.copying = false;
    }
    c->ab_head = c->ab_tail = c->ab_count = 0;
}

static int pinned(void **host, size_t bytes) {
    HIP_TRY(hipHostMalloc(host, std::max<size_t>(bytes, 256), hipHostMallocDefault));
    return 0;
}

// cap: record slots; pack_rec: records the packed block is sized for (a fused pass: the records expected, not every slot -- ten
// gigabytes of pinned memory less at 10^8 rows; a pass that needs more is repeated and the next block is bigger)
static int ensure_async_buf(mc_ctx *c, mc_ctx::AsyncBuf &b, int64_t cap, int k, int64_t pack_rec = 0) {
    const DevTable &T = c->T;
    if (!b.ev_done) {
        // events between kernels of this GPU (timing, the side stream's wait for the emit) need no system-scope fence -- without
        // it a record costs the queue ~5 us instead of ~9; the two the host waits for before it reads pinned memory (ev_done,
        // ev_copied) keep the default
        const unsigned dev_flags = hipEventDisableSystemFence;
        for (hipEvent_t *e : {&b.ev_k0_start, &b.ev_k0_end, &b.ev_scan_start, &b.ev_scan_end, &b.ev_emit_end, &b.ev_k2_start, &b.ev_k2_end, &b.ev_list_end, &b.ev_emit_start})
            HIP_TRY(hipEventCreateWithFlags(e, dev_flags));
        for (hipEvent_t *e : {&b.ev_done, &b.ev_copied, &b.ev_text})
            HIP_TRY(hipEventCreate(e));
    }
    // the strand-resolve output (64 B per name block) and the record set are sized apart: tables that come in turn differ by a few
    // name blocks, and that must not cost a record set (for a one-base motif: gigabytes, pinned) -- with head room, so that it
    // happens once
    if (b.n_nb < T.n_nb) {
        if (b.used) HIP_TRY(hipEventSynchronize(b.ev_done));
        free_pool(b.k0_allocs);
        const int64_t nb = std::max<int64_t>(T.n_nb + T.n_nb / 4 + 64, c->scratch_nb);
        if (dev_alloc(b.k0_allocs, &b.K.desc, (size_t)nb + 1) || dev_alloc(b.k0_allocs, &b.K.nb_f0, (size_t)nb + 1)) return -10;
        b.n_nb = nb;
    }
    if (pack_rec <= 0 || pack_rec > cap) pack_rec = cap;
    const size_t pack_bytes = std::max((size_t)pack_rec * (20 + ((size_t)k + 1) * 8 + 1) + 128, c->pack_min_bytes);       // (every slot mean 64 bits wide at worst, a mask byte per call)
    if (b.cap >= cap && b.k == k && b.pack_bytes >= pack_bytes) return 0;
    if (b.used) HIP_TRY(hipEventSynchronize(b.ev_done));
    free_pool(b.dev_allocs);
    b.H = DevRecords();
    if (alloc_records(b.dev_allocs, b.O, cap, k)) return -10;
    if (dev_alloc(b.dev_allocs, &b.cnt, 1)) return -10;
    // (the pass mark is only ever written by the kernels: whatever fresh device memory holds must not look like a pass number)
    HIP_TRY(hipMemsetAsync(b.cnt, 0, sizeof(Counters), c->stream));
    if (cap >= (int64_t)1 << 31) {
        mc_set_error("mc_extract_features_async: %lld flush records per pass (call rows are 32 bits wide); use mc_extract_features",
                     (long long)cap);
        return -12;
    }
    if (dev_alloc(b.dev_allocs, &b.pack, pack_bytes) || dev_alloc(b.dev_allocs, &b.chunk_cnt, (size_t)PACK_PAD * PACK_WGS)) return -10;
    if (dev_alloc(b.dev_allocs, &b.sorted, (size_t)cap) || dev_alloc(b.dev_allocs, &b.rare, (size_t)cap)) return -10;
    b.piece_cap = cap / 16 + 64;                               // (a piece has at least 48 slots: mc_fused_room)
    if (dev_alloc(b.dev_allocs, &b.piece_cnt, (size_t)b.piece_cap) || dev_alloc(b.dev_allocs, &b.piece_kw, (size_t)b.piece_cap)) return -10;
    if (b.pack_host) { (void)hipHostFree(b.pack_host); b.pack_host = nullptr; }
    if (pinned((void **)&b.pack_host, pack_bytes)) return -10;
    b.pack_bytes = pack_bytes;
    b.H.capacity = cap;
    if (!b.st_host) {
        if (pinned((void **)&b.st_host, sizeof(Counters))) return -10;
        HIP_TRY(hipHostGetDevicePointer((void **)&b.st_dev, b.st_host, 0));
    }
    b.cap = cap;
    b.k = k;
    b.used = false;
    return 0;
}

// Classifier of a pass whose emit has been enqueued (ev_emit_end recorded), on the side stream.
// In front of it the windows the emit left to the row-by-row kernel (longer than 64 rows; usually none): the pass has its
// own sorted payloads and list, so this need not hold up the next pass's strand resolve on the ctx stream.
static int enqueue_k2(mc_ctx *c, mc_ctx::AsyncBuf &b, const K1Args &A) {
    const DevTable &T = c->T;
    hipStream_t st = c->side_stream;
    HIP_TRY(hipStreamWaitEvent(st, b.ev_emit_end, 0));
    mc_launch_rare_dev(A, b.sorted, b.rare, st);
    if (b.timed || !MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_k2_start, st));
    if (b.prm.score)
        launch_classifier(c, st, b.O.feats, b.k, b.O.site_seg, T.seg_read, c->qual, b.O.info, (const uint8_t *)nullptr, b.cap, b.O.prob,
                          (const unsigned long long *)&b.cnt->n_records, (const unsigned int *)&b.cnt->overflow,
                          b.fused_room > 0 ? b.piece_cnt : nullptr, b.fused_room, b.fused_room > 0 ? b.slots / b.fused_room : 0);
    if (b.timed || !MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_k2_end, st));
    return 0;
}

// The side stream of a pass as ONE kernel (k2_mlp<.., PACK>, mc_classify.hip): the windows left to the row-by-row walk, the MLP,
// the packing -- every workgroup for its own records.  -> 1: enqueued (ev_done rides on it); 0: not for this pass (enqueue_k2 +
// enqueue_pack: k1_rare_dev, the context's classifier, k_pack_count / k_pack)
static int enqueue_side(mc_ctx *c, mc_ctx::AsyncBuf &b, const K1Args &A, bool *done) {
    const DevTable &T = c->T;
    hipStream_t st = c->side_stream;
    *done = false;
    const bool other = c->F.left != nullptr || c->Sc.params != nullptr;
    if (b.prm.score && (other || !c->M.W1)) return 0;
    HIP_TRY(hipStreamWaitEvent(st, b.ev_emit_end, 0));
    if (b.timed || !MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_k2_start, st));
    if (!mc_launch_side(c->M, other, c->n_cu, st, A, b.sorted, T.seg_read, c->qual, b.cap, b.prm.score ? 1 : 0, b.pack, b.pack_bytes, b.close32 ? 1 : 0, b.st_dev,
                        b.fused_room, b.fused_room > 0 ? b.slots / b.fused_room : 0, MC_EVENTS_ON_KERNELS ? b.ev_done : nullptr))
        return 0;           // (the wait and the event stay where they are: harmless in front of the three kernels)
    if (!MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_done, st));
    HIP_TRY(hipGetLastError());
    *done = true;
    return 0;
}

// Packing of a pass whose classifier has been enqueued (ev_k2_end recorded): what mc_wait_records_begin copies out.
// count: the chunk counts are not there yet (the emit counts them as it writes the records, except k1_emit_runs)
static int enqueue_pack(mc_ctx *c, mc_ctx::AsyncBuf &b, bool count) {
    hipStream_t s2 = c->side_stream;
    const int holes = b.fused_room > 0 ? 1 : 0;
    if (count) mc_launch_pack_count(b.O, b.cnt, b.k, b.chunk_cnt, holes, s2);
    mc_launch_pack(b.O, b.cnt, b.chunk_cnt, b.pack, b.pack_bytes, b.k, b.close32 ? 1 : 0, b.st_dev, holes, count ? 1 : 0, s2, MC_EVENTS_ON_KERNELS ? b.ev_done : nullptr);
    if (!MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_done, s2));
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mc_extract_features_async(mc_ctx *c, const mc_params *prm) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = check_pass(c, prm)) return rc;
    if (c->ab_count >= MC_PASSES_IN_FLIGHT) {
        mc_set_error("mc_extract_features_async: %d passes are in flight; call mc_wait_records first", MC_PASSES_IN_FLIGHT);
        return -12;
    }
    const DevTable &T = c->T;
    const int k = prm->k;
    mc_ctx::AsyncBuf &b = c->ab[c->ab_head];
    b.prm = *prm;
    if (T.n_rows == 0 || T.n_nb == 0) {            // nothing to scan: an empty pass
        if (int rc = ensure_async_buf(c, b, 1 << 16, k)) return rc;
        memset(b.st_host, 0, sizeof(Counters));
        b.pass_no = ++c->pass_counter;
        b.used = false;
        b.want_text = 0; b.text_block = -1;
        b.fused_room = 0; b.slots = 0;
        b.slot = -1;
        c->ab_head = (c->ab_head + 1) % MC_PASSES_IN_FLIGHT;
        c->ab_count += 1;
        return 0;
    }
    int64_t cap = std::max<int64_t>(guess_capacity(c), c->Omain.capacity);
    // A dense reference (a one-base motif): the pass as ONE kernel with fixed room per piece of the table, holes in between
    // (k1_fused; MCALLER_DENSE_FUSED=0: the scan + emit pair, which the synchronous interface and every repeated pass keep)
    const bool fused_wanted = !(getenv("MCALLER_DENSE_FUSED") && atoi(getenv("MCALLER_DENSE_FUSED")) == 0) && !getenv("MCALLER_NO_EMIT_RUNS");
    int fused_room = 0;
    if (fused_wanted && dense_reference(c)) {
        const double density = (double)c->R.n_sites / (2.0 * (double)c->ref_total_len);
        if (const char *e = getenv("MCALLER_FUSED_ROOM")) fused_room = std::max(1, atoi(e));        // (tests: force the overflow path)
        else fused_room = (int)std::min<int64_t>(mc_fused_room_max(), (int64_t)mc_fused_room(density) * c->fused_scale);
        cap = std::max<int64_t>(cap, mc_fused_pieces(T) * fused_room);
    }
    b.fused_room = fused_room;
    b.slots = fused_room > 0 ? mc_fused_pieces(T) * fused_room : 0;
    if (int rc = ensure_scratch(c, T.n_nb, T.n_tiles)) return rc;
    if (int rc = ensure_records(c, cap, k)) return rc;          // the scratch all passes share (payloads, lists)
    int64_t pack_rec = fused_room > 0 ? std::min<int64_t>(cap, std::max<int64_t>(guess_capacity(c), c->Omain.capacity)) : cap;
    if (const char *e = getenv("MCALLER_PACK_RECORDS")) { if (atoll(e) > 0) pack_rec = std::min<int64_t>(cap, atoll(e)); }      // (tests: a packed block that is too small)
    if (int rc = ensure_async_buf(c, b, cap, k, pack_rec)) return rc;
    if (fused_room > 0 && mc_fused_pieces(T) > b.piece_cap) { b.fused_room = 0; b.slots = 0; }     // (room forced very small: more pieces than counts)
    for (auto &other : c->ab)               // all record sets at once: no (pinned) allocation later, in the middle of a stream
        if (!other.used && (other.cap < cap || other.n_nb < T.n_nb || other.pack_bytes < b.pack_bytes)) { if (int rc = ensure_async_buf(c, other, cap, k, pack_rec)) return rc; }
    // K0 (strand resolve), the scan and the ordering of a pass on the ctx stream, back to back with the next pass: nothing on the
    // scan's path waits for another queue.  The eight-lane EMIT of a sparse reference, K2 (classifier) and the packing on the side
    // stream, behind the pass's ordering (an event on k1_list's dispatch packet): three dependent round trips with a few thousand
    // waves, 39 us during which the next pass's strand resolve and scan stood in the queue behind it -- beside the scan it takes 70-120
    // us and nobody waits for it.  The ctx stream takes a new pass every 0.21 ms then, and a pass is 0.6-0.7 ms from its strand
    // resolve to its records in host memory: with four passes in flight at most the ctx stream ran dry every second pass waiting for
    // the host (the emit on the side stream: 0.264 against 0.262 ms per pass); with six at most, four or five kept in flight: 0.242.
    // (The fused kernel of a dense reference and the run-table emit stay where they are: the one is the scan, the other reads the
    // scan's shared scratch.)
    // Measured earlier on the 10^8-row table (rocprofv3 timelines), passes per second relative to the layout with the emit on the ctx stream:
    // K0 on the side stream beside K2 on the ctx stream -3 % (two cross-queue hand-overs of 15-25 us on the scan's path);
    // K2 + packing deferred so that they run beside the next SCAN: the same (K2 gets one wave per SIMD there and takes 195 us
    // instead of 68); separate streams for K0 and K2: they land on one hardware queue and serialise; low-priority side
    // streams: time-sliced, 40 % slower; K0 of the next pass on a stream of its own, enqueued a whole pass ahead (it touches
    // nothing but the pass's own buffers): 0.286 ms per pass instead of 0.206 (a fifth stream shares a hardware queue), 0.238
    // with GPU_MAX_HW_QUEUES=8 -- which by itself costs 9 % (0.225); odd and even passes on two streams, the scan of a pass
    // waiting for the ordering kernels of the pass before it (the scratch they share) so that it runs beside that pass's emit,
    // one copy stream: 0.268 ms -- the kernels take what they take alone, the queues hand over slowly; the classifier in front
    // of k1_rare_dev (and once more behind it, if that kernel had a window to finish), so that it starts 8 us earlier and
    // runs less beside the scan: 0.2055 instead of 0.1985.
    hipStream_t st = c->stream;
    if (!c->side_stream) HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    // (a hipEventRecord between two kernels costs this queue ~9 us -- rocprofv3 timeline -- so the two events that only time
    // the pass, unlike ev_emit_end, which the side stream waits for, can be thinned out: mc_ctx_set_pass_timing)
    b.timed = c->timing_every > 0 && (c->pass_seq++ % c->timing_every) == 0;
    const PassPlan plan = plan_pass(c, st);
    if (b.timed) HIP_TRY(hipEventRecord(b.ev_k0_start, st));
    b.pass_no = ++c->pass_counter;
    if (int rc = enqueue_k0(c, prm, b.K, b.cnt, st, b.pass_no, false, plan)) return rc;
    if (b.timed) HIP_TRY(hipEventRecord(b.ev_scan_start, st));
    K1Args A;
    // (no event between the scan and the ordering kernels here: a record costs the queue ~5 us; the feature extraction is timed
    // as one span, the split into scan and emit comes from mc_extract_features or from rocprofv3)
    static const bool emit_aside = !(getenv("MCALLER_EMIT_ON_SIDE") && atoi(getenv("MCALLER_EMIT_ON_SIDE")) == 0);      // (=0: the emit stays on the ctx stream)
    if (int rc = enqueue_k1(c, prm, b.K, b.cnt, b.O, st, nullptr, &A, b.sorted, b.rare, b.pass_no, plan, b.ev_emit_end, b.chunk_cnt, b.fused_room, b.piece_cnt, b.piece_kw,
                            emit_aside ? c->side_stream : nullptr, b.ev_list_end, b.timed ? b.ev_emit_start : nullptr, &b.emit_aside)) return rc;
    b.close32 = T.n_rows < INT32_MAX;           // (a closing row can be n_rows itself: the next shard's first row)
    b.want_text = c->rt.on;
    b.text_block = -1;
    b.one_kernel = false;
    if (int rc = enqueue_side(c, b, A, &b.one_kernel)) return rc;
    if (!b.one_kernel) {
        if (int rc = enqueue_k2(c, b, A)) return rc;
        if (int rc = enqueue_pack(c, b, A.chunk_cnt == nullptr)) return rc;
    }
    // (nothing goes on the copy stream here: it is a FIFO, and a wait for THIS pass queued now would hold back the
    // copy-out of the previous pass, which mc_wait_records enqueues later)
    HIP_TRY(hipGetLastError());
    b.used = true;
    b.slot = c->cur;
    b.qual = c->qual;
    b.n_qual = c->n_qual;
    if (b.slot >= 0) c->slots[b.slot].refs += 1;        // the table stays in its slot until the pass has been handed out
    c->ab_head = (c->ab_head + 1) % MC_PASSES_IN_FLIGHT;
    c->ab_count += 1;
    return 0;
}

static int sync_pass_streams(mc_ctx *c) {
    if (c->up_stream) HIP_TRY(hipStreamSynchronize(c->up_stream));
    if (c->parse_stream) HIP_TRY(hipStreamSynchronize(c->parse_stream));
    if (c->side_stream) HIP_TRY(hipStreamSynchronize(c->side_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream2));
    return 0;
}

// The rows of a pass as text, made on the device behind its packed block (mc_rowtext.hip) and sent to a pinned block: for passes
// over a table the device parser made (the read names are in the shard's text), scored, with the context's classifier's key
// table.  Whatever is missing -- no free block, another kind of table -- leaves the pass without text: the host formats.
static int enqueue_row_text(mc_ctx *c, mc_ctx::AsyncBuf &b, int64_t n, int64_t m, int64_t n_wide) {
    auto &R = c->rt;
    if (b.slot < 0 || m <= 0 || n <= 0 || !c->side_stream) { R.n_other += 1; return 0; }
    TableSlot &S = c->slots[b.slot];
    if (!S.from_parser || !S.text || !S.kp_segs || !S.kp_segs_h || !c->kc.chars || S.T.n_seg <= 0) { R.n_other += 1; return 0; }
    const uint8_t *soc = c->F.left ? c->F.sub_of_char : (c->Sc.params ? c->Sc.sub_of_char : c->M.sub_of_char);
    if (!soc || !b.prm.score || !b.qual || !c->R.seq) { R.n_other += 1; return 0; }
    hipStream_t st = c->side_stream;
    if (R.bytes_per_row <= 0.0) {
        size_t longest = 0;
        for (const std::string &nm : c->kc.names) longest = std::max(longest, nm.size());
        R.bytes_per_row = 64.0 + 2.0 * b.k + 20.0 * b.k + 64.0 + (double)longest;
    }
    const size_t need = (size_t)((double)m * R.bytes_per_row) + 4096;
    int at = -1;
    static const int n_blocks = getenv("MCALLER_ROW_TEXT_BLOCKS") ? std::max(0, std::min(MC_ROW_TEXT_BLOCKS, atoi(getenv("MCALLER_ROW_TEXT_BLOCKS")))) : MC_ROW_TEXT_BLOCKS;   // (tests: none free)
    for (int i = 0; i < n_blocks; ++i)
        if (R.blocks[i].busy.load() == 0 && (at < 0 || (R.blocks[at].cap < need && R.blocks[i].cap >= need))) at = i;
    if (at < 0) { R.n_no_block += 1; return 0; }
    auto &blk = R.blocks[at];
    if (blk.cap < need) {
        if (blk.p) { (void)hipHostFree(blk.p); blk.p = nullptr; blk.cap = 0; }
        const size_t cap = need + need / 4;
        if (pinned((void **)&blk.p, cap)) return -10;
        HIP_TRY(hipHostGetDevicePointer((void **)&blk.p_dev, blk.p, 0));
        blk.cap = cap;
    }
    if (!blk.st) {
        if (pinned((void **)&blk.st, sizeof(RowTextStatus))) return -10;
        HIP_TRY(hipHostGetDevicePointer((void **)&blk.st_dev, blk.st, 0));
    }
    if (R.out_cap < need) {
        HIP_TRY(hipStreamSynchronize(st));                  // (the row writer of the pass before may be at work in it)
        free_pool(R.out_allocs);
        R.out = nullptr; R.out_cap = 0;
        const size_t cap = need + need / 4;
        if (dev_alloc(R.out_allocs, &R.out, cap)) return -10;
        R.out_cap = cap;
    }
    int64_t nbr, nbw;
    mc_row_text_scratch_sizes(n, m, &nbr, &nbw);
    if (R.cap_rec < n || R.cap_rows < m || R.cap_wide < n_wide || R.cap_num < n_wide + b.n_qual) {
        HIP_TRY(hipStreamSynchronize(st));
        free_pool(R.allocs);
        R.cap_rec = n + n / 4 + 1024; R.cap_rows = m + m / 4 + 1024;
        R.cap_wide = std::max<int64_t>(n_wide + n_wide / 4 + 1024, R.cap_wide);
        R.cap_num = std::max<int64_t>(R.cap_wide + b.n_qual + b.n_qual / 4 + 1024, R.cap_num);
        int64_t cbr, cbw;
        mc_row_text_scratch_sizes(R.cap_rec, R.cap_rows, &cbr, &cbw);
        if (dev_alloc(R.allocs, &R.S.kept_blk, (size_t)cbr + 2) || dev_alloc(R.allocs, &R.S.wide_blk, (size_t)cbw + 2) ||
            dev_alloc(R.allocs, &R.S.wide_pref, (size_t)R.cap_rows) || dev_alloc(R.allocs, &R.S.rec_len, (size_t)R.cap_rec) ||
            dev_alloc(R.allocs, &R.S.rec_row, (size_t)R.cap_rec) || dev_alloc(R.allocs, &R.S.len_blk, (size_t)cbr + 2) ||
            dev_alloc(R.allocs, &R.S.wval, (size_t)R.cap_wide) || dev_alloc(R.allocs, &R.S.num_lo, (size_t)R.cap_num) ||
            dev_alloc(R.allocs, &R.S.num_meta, (size_t)R.cap_num) || dev_alloc(R.allocs, &R.S.st, 1)) {
            R.cap_rec = R.cap_rows = R.cap_wide = R.cap_num = 0;
            return -10;
        }
    }
    const DevTable &T = S.T;
    RowTextIn I;
    I.pack = b.pack; I.n = n; I.m = m; I.n_wide = n_wide; I.k = b.k; I.close32 = b.close32 ? 1 : 0;
    I.seg_begin = T.seg_begin; I.seg_read = T.seg_read; I.seg_contig = T.seg_contig; I.n_seg = T.n_seg;
    I.segs = S.kp_segs; I.text = S.text;
    I.qual = b.qual; I.n_qual = b.n_qual;
    I.R = c->R;
    I.cn_off = c->kc.name_off; I.cn_len = c->kc.name_len; I.cn_chars = c->kc.chars;
    I.sub_of_char = soc;
    I.tail_contig = b.prm.tail_contig;
    memcpy(I.lab_meth, R.lab_meth, 8); memcpy(I.lab_unmeth, R.lab_unmeth, 8);
    I.lab_meth_len = R.lab_meth_len; I.lab_unmeth_len = R.lab_unmeth_len;
    // (the parser listed the segments as its lanes got there; the host's copy is in file order since mc_ctx_parse_end)
    if (int rc = copy_by_kernel(S.kp_segs, S.kp_segs_h, (size_t)T.n_seg * sizeof(KpSeg), st)) return rc;
    mc_launch_row_text(I, R.S, R.out, R.room_forced ? std::min(need, std::min(R.out_cap, blk.cap)) : std::min(R.out_cap, blk.cap), blk.p_dev, blk.st_dev, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b.ev_text, st));
    R.next_ticket = R.next_ticket >= (1 << 24) ? 1 : R.next_ticket + 1;
    blk.busy.store(R.next_ticket);
    b.text_block = at;
    return 0;
}

// Copy-out of the oldest pass in flight whose copy-out has not been started, started but not waited for: the counters are
// read (k_pack left them in pinned memory; a wait for the pass's kernels), then one DMA transfer of exactly what the pass
// produced is enqueued on the copy stream.  Called for pass i+1 before mc_wait_records(i), the transfers run back to back:
// no host round trip sits between two copy-outs.  (No-op when every pass in flight is being copied out already.)
extern "C" int mc_wait_records_begin(mc_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    if (c->ab_count == 0) {
        mc_set_error("mc_wait_records_begin: no pass in flight");
        return -12;
    }
    int at = c->ab_tail, left = c->ab_count;
    while (left > 0 && c->ab[at].copying) { at = (at + 1) % MC_PASSES_IN_FLIGHT; --left; }
    if (left == 0) return 0;
    mc_ctx::AsyncBuf &b = c->ab[at];
    // (two copy streams, taken in turn: a transfer that is enqueued while the previous one runs starts beside its tail; on
    // one stream 15-20 us pass between the end of one transfer and the start of the next -- rocprofv3 timeline -- which is
    // 8 % of a pass that the copy-out bounds)
    hipStream_t cs = (at & 1) ? c->copy_stream2 : c->copy_stream;
    if (b.used) {                                                // the counters (k_pack stored them in st_host), then exactly
        HIP_TRY(hipEventSynchronize(b.ev_done));                 // n records with the DMA engines
        HIP_TRY(hipStreamWaitEvent(cs, b.ev_done, 0));
    }
    const Counters &st = *b.st_host;
    const bool special = st.overflow || st.irregular_pass == b.pass_no;      // (long windows were finished on the device: k1_rare_dev)
    if (b.used && !special && st.n_records > 0) {
        const size_t n = (size_t)std::min<int64_t>((int64_t)st.n_records, b.cap);
        const int k = b.k;
        const size_t m = (size_t)std::min<unsigned long long>(st.n_kept, n);
        const PackLayout L = pack_layout((int64_t)n, b.close32 ? 1 : 0);
        const size_t n_wide = (size_t)std::min<unsigned long long>(st.n_wide, (unsigned long long)m * (size_t)k);
        const PackTail PT_ = pack_tail(L.feats, m, k, n_wide);
        const size_t out_bytes = PT_.end;
        // (a small record set -- a shard of a streamed file -- by kernel: the DMA engines may be busy with text, see k_copy_bytes;
        // and on the side stream, right behind the packing: the runtime folds the streams of a process onto four hardware
        // queues, and a copy stream that shares one with the parse stream would wait behind the kernels of the shards ahead,
        // which wait for their text)
        // (... and while shards of TEXT are on their way -- a slot is being parsed -- also a big one: the 7 MB of a dense shard's
        // records queued behind six shards of text, 2.5 ms each, and the host waited 0.24 s per 10^8 rows for records that were
        // long computed.  Resident tables, nothing on the link: the DMA engines, which do not go through the shader caches)
        static const size_t by_kernel_env = getenv("MCALLER_RECORDS_BY_KERNEL_MAX") ? (size_t)atoll(getenv("MCALLER_RECORDS_BY_KERNEL_MAX")) : 0;
        bool text_on_the_link = false;
        for (const TableSlot &S : c->slots) text_on_the_link = text_on_the_link || S.kp_state == 1;
        const size_t by_kernel_max = by_kernel_env ? by_kernel_env : (text_on_the_link ? COPY_BY_KERNEL_MAX_STREAMING : COPY_BY_KERNEL_MAX);
        if (out_bytes <= by_kernel_max && c->side_stream) {
            cs = c->side_stream;
            if (int rc = copy_by_kernel(b.pack_host, b.pack, out_bytes, cs)) return rc;
        } else HIP_TRY(hipMemcpyAsync(b.pack_host, b.pack, out_bytes, hipMemcpyDeviceToHost, cs));
        b.H.close_row = b.close32 ? nullptr : reinterpret_cast<int64_t *>(b.pack_host);
        b.h_close32 = b.close32 ? reinterpret_cast<int32_t *>(b.pack_host) : nullptr;
        b.H.site_pos = reinterpret_cast<int32_t *>(b.pack_host + L.pos);
        b.H.site_seg = reinterpret_cast<int32_t *>(b.pack_host + L.seg);
        b.H.info = reinterpret_cast<uint32_t *>(b.pack_host + L.info);
        b.H.feats = nullptr;                                             // (they travel as 32-bit integers where they can)
        b.h_lo32 = reinterpret_cast<int32_t *>(b.pack_host + PT_.lo32);
        b.H.prob = reinterpret_cast<double *>(b.pack_host + PT_.prob);
        b.h_wmask = b.pack_host + PT_.wmask;
        b.h_hi32 = reinterpret_cast<uint32_t *>(b.pack_host + PT_.hi32);
        b.h_n_wide = (int64_t)n_wide;
        b.h_n_calls = (int64_t)m;
        HIP_TRY(hipEventRecord(b.ev_copied, cs));
        b.text_block = -1;
        if (b.want_text) { if (int rc = enqueue_row_text(c, b, (int64_t)n, (int64_t)m, (int64_t)n_wide)) return rc; }
    }
    b.copying = true;
    return 0;
}

extern "C" int mc_wait_records(mc_ctx *c, int64_t *n_records, mc_calls_view *out) {
    HIP_TRY(hipSetDevice(c->device));
    if (c->ab_count == 0) {
        mc_set_error("mc_wait_records: no pass in flight");
        return -12;
    }
    if (!c->ab[c->ab_tail].copying) { if (int rc = mc_wait_records_begin(c)) return rc; }
    mc_ctx::AsyncBuf &b = c->ab[c->ab_tail];
    c->ab_tail = (c->ab_tail + 1) % MC_PASSES_IN_FLIGHT;
    c->ab_count -= 1;
    b.copying = false;
    // the pass leaves flight: its table stays put as "the table of the records handed out last" (mc_site_counts) until the
    // next pass is handed out
    if (b.slot >= 0) {
        c->slots[b.slot].refs -= 1;
        c->held = b.slot;
        c->last_T = c->slots[b.slot].T;
    }
    const Counters st = *b.st_host;
    const bool special = st.overflow || st.irregular_pass == b.pass_no;
    if (b.used && !special && st.n_records > 0) HIP_TRY(hipEventSynchronize(b.ev_copied));
    c->rt.last_block = -1; c->rt.last_bytes = c->rt.last_rows = 0;
    if (b.text_block >= 0) {                                  // the rows as text (enqueue_row_text)
        auto &blk = c->rt.blocks[b.text_block];
        HIP_TRY(hipEventSynchronize(b.ev_text));
        const RowTextStatus ts = *blk.st;
        if (ts.too_small && st.n_kept > 0)
            c->rt.bytes_per_row = std::max(c->rt.bytes_per_row, 1.25 * (double)ts.n_bytes / (double)st.n_kept + 8.0);
        c->rt.n_host_needed += ts.host_needed ? 1 : 0;
        if (ts.host_needed && getenv("MCALLER_VERBOSE")) fprintf(stderr, "mcaller_hip: a pass's rows left to the host formatter (reasons 0x%x)\n", ts.host_needed);
        c->rt.n_too_small += ts.too_small ? 1 : 0;
        if (special || ts.host_needed || ts.too_small || ts.n_bytes > blk.cap) blk.busy.store(0);
        else { c->rt.n_text += 1; c->rt.last_block = b.text_block; c->rt.last_bytes = (int64_t)ts.n_bytes; c->rt.last_rows = (int64_t)ts.n_rows; }
        b.text_block = -1;
    }
    if (special && getenv("MCALLER_VERBOSE"))
        fprintf(stderr, "mcaller_hip: pass re-run synchronously (overflow %u, irregular %u, big %u, rare %u, records %llu)\n",
                st.overflow, (unsigned)(st.irregular_pass == b.pass_no), st.n_big, st.n_rare, st.n_records);
    c->last_fused_room = b.used ? b.fused_room : 0;
    c->last_rerun = special ? 1 : 0;
    if (special && st.overflow && st.pack_need > 0)           // (the packed block was too small: the next one holds that and a quarter)
        c->pack_min_bytes = std::max(c->pack_min_bytes, (size_t)st.pack_need + (size_t)st.pack_need / 4);
    else if (special && st.overflow && b.fused_room > 0 && b.fused_room < mc_fused_room_max())
        c->fused_scale = std::min(c->fused_scale * 2, 64);       // (a piece ran out of room: twice the room from the next pass on)
    if (special) {
        // a pass the fast path alone cannot finish (record buffers too small, irregular reads):
        // run it again through mc_extract_features, which handles all of that, and hand out its buffers
        // (on the table the pass was enqueued for, which need not be the current one any more)
        int64_t n = 0;
        const DevTable T_now = c->T;
        const double *q_now = c->qual;
        const int cur_now = c->cur, nq_now = c->n_qual;
        if (b.slot >= 0) { c->T = c->slots[b.slot].T; c->qual = const_cast<double *>(b.qual); c->n_qual = b.n_qual; c->cur = b.slot; }
        c->in_rerun = true;
        const int rc = mc_extract_features(c, &b.prm, &n);
        c->in_rerun = false;
        c->T = T_now; c->qual = const_cast<double *>(q_now); c->n_qual = nq_now; c->cur = cur_now;
        if (rc) return rc;
        c->last_timed = 1;         // (mc_extract_features times every pass)
        *n_records = n;
        return mc_fetch_records_view(c, out);
    }
    const int64_t n = (int64_t)st.n_records;
    c->last_timed = (b.used && b.timed) ? 1 : 0;
    if (b.used && b.timed) {
        float t_k0 = 0, t_scan = 0, t_emit = 0, t_k2 = 0;
        HIP_TRY(hipEventElapsedTime(&t_k0, b.ev_k0_start, b.ev_scan_start));
        if (b.emit_aside && MC_EVENTS_ON_KERNELS) {
            // (the emit ran on the side stream: the ctx stream's span ends with the ordering, the emit's own time beside it)
            HIP_TRY(hipEventElapsedTime(&t_scan, b.ev_scan_start, b.ev_list_end));
            HIP_TRY(hipEventElapsedTime(&t_emit, b.ev_emit_start, b.ev_emit_end));
        } else {
            HIP_TRY(hipEventElapsedTime(&t_scan, b.ev_scan_start, b.ev_emit_end));     // scan + ordering + emit, one span
            t_emit = 0.0f;
        }
        HIP_TRY(hipEventElapsedTime(&t_k2, b.ev_k2_start, b.one_kernel ? b.ev_done : b.ev_k2_end));
        c->times[0] = t_k0; c->times[1] = t_scan; c->times[2] = t_emit; c->times[3] = t_k2;
        c->times[4] = t_k0 + t_scan + t_emit + t_k2;
    }
    c->O = b.O;                    // what mc_site_counts reduces: the records of the pass just handed out
    c->last_n = n;
    c->last_slots = b.fused_room > 0 ? std::min<int64_t>(b.slots, b.cap) : n;
    c->last_k = b.k ? b.k : c->last_k;
    *n_records = n;
    out->capacity = n;
    out->feats = b.H.feats; out->site_pos = b.H.site_pos; out->site_seg = b.H.site_seg;
    out->close_row = b.H.close_row; out->info = b.H.info; out->prob = b.H.prob;
    out->close_row32 = b.h_close32;
    out->call_row = nullptr;       // (not sent: the row of record j is the number of records before it without MC_I_TOO_MANY)
    out->compacted = 1;
    const bool packed = n > 0 && b.used;
    out->feats_lo32 = packed ? b.h_lo32 : nullptr;
    out->feats_hi32 = packed ? b.h_hi32 : nullptr;
    out->feats_wide = packed ? b.h_wmask : nullptr;
    out->n_wide = packed ? b.h_n_wide : 0;
    out->n_call_rows = n > 0 && b.used ? b.h_n_calls : 0;
    return 0;
}

// ---- rows of text made on the device ----
extern "C" int mc_ctx_row_text(mc_ctx *c, int32_t on, const char *label_meth, const char *label_unmeth) {
    auto &R = c->rt;
    if (on) {
        const size_t lm = label_meth ? strlen(label_meth) : 0, lu = label_unmeth ? strlen(label_unmeth) : 0;
        if (lm == 0 || lu == 0 || lm > 8 || lu > 8) {
            mc_set_error("mc_ctx_row_text: labels of 1..8 characters");
            return -12;
        }
        memset(R.lab_meth, 0, 8); memset(R.lab_unmeth, 0, 8);
        memcpy(R.lab_meth, label_meth, lm); memcpy(R.lab_unmeth, label_unmeth, lu);
        R.lab_meth_len = (int)lm; R.lab_unmeth_len = (int)lu;
        // on == 2, the first call of a stream: whatever held a block is gone (a stream that ended on an exception never gave its blocks
        // back).  Not on the later calls: with no pass in flight the host's writer may still be reading the blocks of the passes handed out
        if (on == 2 && c->ab_count == 0)
            for (auto &blk : R.blocks) blk.busy.store(0);
        R.room_forced = getenv("MCALLER_ROW_TEXT_ROOM") != nullptr;                                           // (tests: rows that do not fit)
        if (R.room_forced && !R.on) R.bytes_per_row = std::max(1.0, atof(getenv("MCALLER_ROW_TEXT_ROOM")));
    }
    if (!on && R.on && getenv("MCALLER_VERBOSE"))
        fprintf(stderr, "mcaller_hip: rows written on the device for %lld passes; not for %lld (no free block), %lld (a record for the host), %lld (room too small), %lld (other)\n",
                R.n_text, R.n_no_block, R.n_host_needed, R.n_too_small, R.n_other);
    R.on = on ? 1 : 0;
    return 0;
}

// (*block: the block's index and the ticket it was taken with -- a handle that has been given back, or that a later stream's first
// mc_ctx_row_text(2) declared void, frees nothing when it is given back again)
extern "C" int mc_last_row_text(mc_ctx *c, const char **text, int64_t *n_bytes, int64_t *n_rows, int32_t *block) {
    const auto &R = c->rt;
    *block = R.last_block >= 0 ? R.blocks[R.last_block].busy.load() * 8 + R.last_block : -1;
    *text = R.last_block >= 0 ? R.blocks[R.last_block].p : nullptr;
    *n_bytes = R.last_block >= 0 ? R.last_bytes : 0;
    *n_rows = R.last_block >= 0 ? R.last_rows : 0;
    return 0;
}

// (any thread: the host's writer gives a block back when the rows are in the file)
extern "C" int mc_row_text_release(mc_ctx *c, int32_t block) {
    if (block < 0 || (block & 7) >= MC_ROW_TEXT_BLOCKS) {
        mc_set_error("mc_row_text_release: no such block (%d)", block);
        return -12;
    }
    int ticket = block >> 3;
    if (ticket > 0) (void)c->rt.blocks[block & 7].busy.compare_exchange_strong(ticket, 0);     // (a stale handle: the block is somebody else's by now)
    return 0;
}

extern "C" int mc_last_pass_info(mc_ctx *c, int32_t *fused_room, int32_t *rerun) {
    if (fused_room) *fused_room = c->last_fused_room;
    if (rerun) *rerun = c->last_rerun;
    return 0;
}

extern "C" int mc_ctx_set_pass_timing(mc_ctx *c, int every_n) {
    if (every_n < 0) { mc_set_error("mc_ctx_set_pass_timing: every_n < 0"); return -12; }
    c->timing_every = every_n;
    return 0;
}

extern "C" int mc_last_pass_timed(mc_ctx *c) { return c->last_timed; }

extern "C" int mc_last_times_ms(mc_ctx *c, float *out5) {
    for (int i = 0; i < 5; ++i) out5[i] = c->times[i];
    return 0;
}

// predict_proba of the context's classifier on n input rows from the host (what the reference's call site :199 does, batched)
static int classifier_forward(mc_ctx *c, int which, const double *X, const uint8_t *submodel, int64_t n, double *p) {
    HIP_TRY(hipSetDevice(c->device));
    const bool have = which == 1 ? c->F.left != nullptr : (which == 2 ? c->Sc.params != nullptr : c->M.W1 != nullptr);
    if (!have) {
        mc_set_error("classifier forward: no %s set", which == 1 ? "forest" : (which == 2 ? "logistic / naive Bayes model" : "MLP"));
        return -12;
    }
    if (n <= 0) return 0;
    double *dX = nullptr, *dp = nullptr;
    uint8_t *ds = nullptr;
    const int ni = classifier_inputs(c);
    HIP_TRY(hipMalloc((void **)&dX, (size_t)n * ni * 8));
    HIP_TRY(hipMalloc((void **)&dp, (size_t)n * 8));
    HIP_TRY(hipMalloc((void **)&ds, (size_t)n));
    HIP_TRY(hipMemcpyAsync(dX, X, (size_t)n * ni * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(ds, submodel, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(dp, 0xFF, (size_t)n * 8, c->stream));   // NaN
    launch_classifier(c, c->stream, dX, ni - 1, (const int32_t *)nullptr, (const int32_t *)nullptr, (const double *)nullptr,
                      (const uint32_t *)nullptr, ds, n, dp, (const unsigned long long *)nullptr, (const unsigned int *)nullptr);
    HIP_TRY(hipMemcpyAsync(p, dp, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    (void)hipFree(dX);
    (void)hipFree(dp);
    (void)hipFree(ds);
    return 0;
}

extern "C" int mc_mlp_forward(mc_ctx *c, const double *X, const uint8_t *submodel, int64_t n, double *p) {
    return classifier_forward(c, 0, X, submodel, n, p);
}

extern "C" int mc_forest_forward(mc_ctx *c, const double *X, const uint8_t *submodel, int64_t n, double *p) {
    return classifier_forward(c, 1, X, submodel, n, p);
}

extern "C" int mc_simple_forward(mc_ctx *c, const double *X, const uint8_t *submodel, int64_t n, double *p) {
    return classifier_forward(c, 2, X, submodel, n, p);
}

// ===================================================================================================
// Per-site reduction feeding make_bed (make_bed.py:86-96: per (chrom, pos, strand) the list of 0/1 labels; :143,:154
// its mean and length; :134 rows in first-occurrence order).  Each rank counts its own records on the device; the
// one exchange step of the multi-GPU job is an all-reduce (sum of the counts, min of the first-seen row) over RCCL.
// ===================================================================================================
namespace {

// site number of (contig, strand, position); -1 if the position is not a marked site
__device__ __forceinline__ int64_t site_number(const DevRef &R, int contig, int rev, int64_t pos) {
    if (contig < 0 || contig >= R.n_contigs || pos < 0 || pos >= R.contig_len[contig]) return -1;
    const int64_t w = R.word_off[contig] + (pos >> 5);
    const uint32_t word = (rev ? R.mr : R.mf)[w];
    if (!((word >> (pos & 31)) & 1u)) return -1;
    const int before = (rev ? R.rank_r : R.rank_f)[w] + __popc(word & ((1u << (pos & 31)) - 1u));
    return R.site_base[contig * 2 + rev] + before;
}

__global__ void k_site_fill(int32_t *cnt, int64_t *first, int64_t n_sites) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < 2 * n_sites) cnt[i] = 0;
    if (i < n_sites) first[i] = INT64_MAX;
}

// one thread per flush record: scored, unskipped records add to their site (label 'm...' <=> p >= 0.5, :200).  make_bed
// keys a row on its chrom column, and that is the contig of the row that CLOSED the window (R8, :216): a record closed by a
// row of another contig is no site of the numbering -- counted in status[2] and left to the caller (a handful per file: the
// last window before a contig switch).
__global__ void k_site_counts(DevRef R, DevRecords O, int64_t n, DevTable T, int tail_contig, int64_t row_offset,
                              int32_t *__restrict__ cnt, int64_t *__restrict__ first, int64_t n_sites,
                              unsigned long long *__restrict__ status) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t info = O.info[j];
    if (info & MC_I_TOO_MANY) return;
    const int site_contig = T.seg_contig[O.site_seg[j]];
    const int64_t cr = O.close_row[j];
    int close_contig = tail_contig;
    if (cr < T.n_rows) {
        int lo = 0, hi = T.n_seg - 1;                               // last segment that begins at or before the closing row
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (T.seg_begin[mid] <= cr) lo = mid; else hi = mid - 1;
        }
        close_contig = T.seg_contig[lo];
    }
    if (close_contig != site_contig) { atomicAdd(&status[2], 1ull); return; }
    const double p = O.prob[j];
    if (p != p) { atomicAdd(&status[0], 1ull); return; }           // scored by the host (edge records): added by the caller
    const int64_t s = site_number(R, site_contig, (info & MC_I_REV) ? 1 : 0, O.site_pos[j]);
    if (s < 0) { atomicAdd(&status[1], 1ull); return; }
    atomicAdd(&cnt[n_sites + s], 1);
    if (p >= 0.5) atomicAdd(&cnt[s], 1);
    atomicMin(reinterpret_cast<long long *>(&first[s]), (long long)(O.close_row[j] + row_offset));
}

// the status words of one accumulation to where the host reads them; zeroed for the next
__global__ void k_site_status_out(unsigned long long *__restrict__ status, unsigned long long *__restrict__ host) {
    if (threadIdx.x < 4) {
        host[threadIdx.x] = status[threadIdx.x];
        status[threadIdx.x] = 0;
    }
}

}  // namespace

// The reduction has a queue of its own (site_stream): the records it reads are those of the pass handed out last -- complete
// since mc_wait_records returned --, so nothing of it has to wait for, or hold up, the passes in flight on the ctx stream
// (a shard's reduction used to drain that stream, allocate and free a status block, and fetch 24 bytes through the DMA
// engines, behind every shard of text on its way: 2 ms per shard).  What the host reads comes back through pinned memory
// written by a kernel.
static int ensure_site_buffers(mc_ctx *c) {
    const int64_t n = c->R.n_sites;
    if (!c->site_stream) HIP_TRY(hipStreamCreateWithFlags(&c->site_stream, hipStreamNonBlocking));
    if (!c->site_status) {
        HIP_TRY(hipMalloc((void **)&c->site_status, 256));
        HIP_TRY(hipMemset(c->site_status, 0, 256));
    }
    if (!c->site_status_host) {
        HIP_TRY(hipHostMalloc((void **)&c->site_status_host, 256, hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&c->site_status_host_dev, c->site_status_host, 0));
    }
    if (c->site_cnt && c->site_n == n) return 0;
    HIP_TRY(hipStreamSynchronize(c->site_stream));
    if (c->site_cnt) (void)hipFree(c->site_cnt);
    if (c->site_first) (void)hipFree(c->site_first);
    c->site_cnt = nullptr; c->site_first = nullptr;
    HIP_TRY(hipMalloc((void **)&c->site_cnt, std::max<size_t>((size_t)n * 8, 256)));
    HIP_TRY(hipMalloc((void **)&c->site_first, std::max<size_t>((size_t)n * 8, 256)));
    c->site_n = n;
    return 0;
}

extern "C" int64_t mc_site_count(mc_ctx *c) { return c->R.n_sites; }

extern "C" int mc_site_counts_reset(mc_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    if (!c->R.mf) {
        mc_set_error("mc_site_counts_reset: no reference set");
        return -12;
    }
    if (int rc = ensure_site_buffers(c)) return rc;
    const int64_t ns = c->R.n_sites;
    hipLaunchKernelGGL(k_site_fill, dim3((unsigned)((2 * ns + 255) / 256 + 1)), dim3(256), 0, c->site_stream, c->site_cnt, c->site_first, ns);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mc_site_counts(mc_ctx *c, int64_t row_offset, int32_t tail_contig, int64_t *n_pending, int64_t *n_cross_contig) {
    if (int rc = mc_site_counts_reset(c)) return rc;
    return mc_site_counts_accumulate(c, row_offset, tail_contig, n_pending, n_cross_contig);
}

extern "C" int mc_site_counts_accumulate(mc_ctx *c, int64_t row_offset, int32_t tail_contig, int64_t *n_pending, int64_t *n_cross_contig) {
    HIP_TRY(hipSetDevice(c->device));
    if (!c->R.mf || !c->site_cnt || c->site_n != c->R.n_sites) {
        mc_set_error("mc_site_counts_accumulate: call mc_site_counts_reset first (after the reference has been set)");
        return -12;
    }
    const int64_t ns = c->R.n_sites, n = c->last_slots;      // (every slot: a hole is a record that is not a call)
    if (n > 0)
        hipLaunchKernelGGL(k_site_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->site_stream, c->R, c->O, n,
                           c->last_T.seg_contig ? c->last_T : c->T, (int)tail_contig, row_offset, c->site_cnt,
                           c->site_first, ns, c->site_status);
    hipLaunchKernelGGL(k_site_status_out, dim3(1), dim3(64), 0, c->site_stream, c->site_status, c->site_status_host_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->site_stream));       // (this queue only: the passes in flight go on)
    const unsigned long long h[3] = {c->site_status_host[0], c->site_status_host[1], c->site_status_host[2]};
    if (h[1]) {
        mc_set_error("mc_site_counts: %llu records name a position that is not a marked site", h[1]);
        return -14;
    }
    if (n_pending) *n_pending = (int64_t)h[0];
    if (n_cross_contig) *n_cross_contig = (int64_t)h[2];
    return 0;
}

extern "C" int mc_site_counts_add(mc_ctx *c, const int64_t *site, const uint8_t *is_meth, const int64_t *first_row, int64_t n) {
    HIP_TRY(hipSetDevice(c->device));
    if (!c->site_cnt) {
        mc_set_error("mc_site_counts_add: call mc_site_counts first");
        return -12;
    }
    const int64_t ns = c->site_n;
    if (n <= 0) return 0;
    // rare (records the host scored itself): read-modify-write from the host
    std::vector<int32_t> cnt((size_t)ns * 2);
    std::vector<int64_t> first((size_t)ns);
    HIP_TRY(hipMemcpy(cnt.data(), c->site_cnt, (size_t)ns * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(first.data(), c->site_first, (size_t)ns * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = site[i];
        if (s < 0 || s >= ns) {
            mc_set_error("mc_site_counts_add: site %lld out of range", (long long)s);
            return -12;
        }
        cnt[(size_t)(ns + s)] += 1;
        if (is_meth[i]) cnt[(size_t)s] += 1;
        first[(size_t)s] = std::min(first[(size_t)s], first_row[i]);
    }
    HIP_TRY(hipMemcpy(c->site_cnt, cnt.data(), (size_t)ns * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->site_first, first.data(), (size_t)ns * 8, hipMemcpyHostToDevice));
    return 0;
}

// ---- RCCL, loaded on first use (the library is only needed by multi-GPU jobs) ----
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.h) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        mc_set_error("cannot load librccl.so: %s", dlerror());
        return -15;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GetErrorString) {
        mc_set_error("librccl.so lacks an expected symbol");
        dlclose(h);
        return -15;
    }
    g_rccl.h = h;
    return 0;
}
}  // namespace

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            mc_set_error("%s failed: %s", #expr, g_rccl.GetErrorString(_r));                        \
            return -15;                                                                             \
        }                                                                                           \
    } while (0)

static_assert(sizeof(ncclUniqueId) == MC_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int mc_comm_available(void) { return rccl_load(); }

extern "C" int mc_comm_unique_id(uint8_t *out) {
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return 0;
}

extern "C" int mc_comm_init(mc_ctx *c, int32_t world, int32_t rank, const uint8_t *unique_id) {
    HIP_TRY(hipSetDevice(c->device));
    if (world < 1 || rank < 0 || rank >= world) {
        mc_set_error("mc_comm_init: rank %d of %d", rank, world);
        return -12;
    }
    if (int rc = rccl_load()) return rc;
    mc_comm_destroy(c);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->comm_world = world;
    c->comm_rank = rank;
    return 0;
}

extern "C" int mc_comm_destroy(mc_ctx *c) {
    if (c && c->comm && g_rccl.h) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
    }
    if (c) c->comm = nullptr;
    return 0;
}

extern "C" int mc_site_counts_fetch(mc_ctx *c, int32_t *n_meth, int32_t *n_total, int64_t *first_row) {
    HIP_TRY(hipSetDevice(c->device));
    if (!c->site_cnt) {
        mc_set_error("mc_site_counts_fetch: call mc_site_counts first");
        return -12;
    }
    const int64_t ns = c->site_n;
    if (ns > 0) {
        HIP_TRY(hipMemcpyAsync(n_meth, c->site_cnt, (size_t)ns * 4, hipMemcpyDeviceToHost, c->site_stream));
        HIP_TRY(hipMemcpyAsync(n_total, c->site_cnt + ns, (size_t)ns * 4, hipMemcpyDeviceToHost, c->site_stream));
        HIP_TRY(hipMemcpyAsync(first_row, c->site_first, (size_t)ns * 8, hipMemcpyDeviceToHost, c->site_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->site_stream));
    return 0;
}

extern "C" int mc_site_allreduce(mc_ctx *c, int32_t *n_meth, int32_t *n_total, int64_t *first_row, float *ms) {
    HIP_TRY(hipSetDevice(c->device));
    if (!c->site_cnt) {
        mc_set_error("mc_site_allreduce: call mc_site_counts first");
        return -12;
    }
    const int64_t ns = c->site_n;
    if (ms) *ms = 0.f;
    if (c->comm && c->comm_world > 1 && ns > 0) {
        HIP_TRY(hipEventRecord(c->ev[0], c->site_stream));
        RCCL_TRY(g_rccl.AllReduce(c->site_cnt, c->site_cnt, (size_t)ns * 2, ncclInt32, ncclSum, (ncclComm_t)c->comm, c->site_stream));
        RCCL_TRY(g_rccl.AllReduce(c->site_first, c->site_first, (size_t)ns, ncclInt64, ncclMin, (ncclComm_t)c->comm, c->site_stream));
        HIP_TRY(hipEventRecord(c->ev[1], c->site_stream));
        HIP_TRY(hipStreamSynchronize(c->site_stream));
        if (ms) HIP_TRY(hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    }
    if (ns > 0) {
        HIP_TRY(hipMemcpyAsync(n_meth, c->site_cnt, (size_t)ns * 4, hipMemcpyDeviceToHost, c->site_stream));
        HIP_TRY(hipMemcpyAsync(n_total, c->site_cnt + ns, (size_t)ns * 4, hipMemcpyDeviceToHost, c->site_stream));
        HIP_TRY(hipMemcpyAsync(first_row, c->site_first, (size_t)ns * 8, hipMemcpyDeviceToHost, c->site_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->site_stream));
    return 0;
}

