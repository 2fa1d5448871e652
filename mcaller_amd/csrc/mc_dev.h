// mc_dev.h -- what the translation units of libmcaller_hip.so's device side share: the structures that live in HBM (event table,
// marked reference, name-block descriptors, records, payloads, counters), the constants of the tile / chunk geometry, and the small
// device functions more than one unit uses.  The kernels themselves and their launch geometry: mc_k0.hip (strand resolve),
// mc_scan.hip (the scan, ordering), mc_emit.hip (window emit, row-by-row kernels), mc_literal.hip (irregular reads),
// mc_classify.hip (MLP / forest / LR / NBC, packing); the host side (contexts, table slots, passes, device parser, per-site
// reduction, RCCL): mc_stream.hip.  C ABI: include/mcaller_hip.h.
#ifndef MC_DEV_H
#define MC_DEV_H
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/mcaller_hip.h"

void mc_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            mc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -10;                                                                     \
        }                                                                                   \
    } while (0)

#ifndef MC_TILE
#define MC_TILE 2048
#endif
constexpr int TILE = MC_TILE;       // rows per workgroup tile
constexpr int O_NONE = 15;

// meta byte per staged row: bit0 valid (passes :167-168), bit1 first row of a name block, bits 2..5 offset of
// the first 'M' in the row's k-mer (O_NONE: not a site row)
constexpr uint32_t M_VALID = 1, M_NS = 2;

enum : uint8_t { MODE_NONE = 0, MODE_REGULAR = 1, MODE_IRREGULAR = 2 };

// validation flags per name block: what its rows look like when each is compared with the row before it (the first pass's
// scan ORs them together, k1_scan; V_MULTI_SEG comes from the host with the table)
constexpr uint32_t V_POS_DEC = 1, V_IDX_INC = 2, V_IDX_DEC = 4, V_IDX_EQ = 8, V_POS0 = 16, V_MULTI_SEG = 32;

constexpr int32_t NO_STRAY = INT32_MIN;
constexpr int32_t NO_SEQ_DELTA = INT32_MIN;

struct __attribute__((aligned(16))) NbDesc {
    int64_t row_begin;
    int64_t row_end;    // one past the block's last row
    int64_t mask_off;   // word offset of the contig's strand masks (both strands share it)
    int32_t first_delta;// rows >= row_begin + first_delta are tested on the block's strand; window walks stop there
    int32_t contig_len;
    int32_t contig;
    int32_t read;
    int32_t stray_q;    // pseudo-position of the stray event of a palindromic first site row once the strand flips
                        // (:276-277), NO_STRAY if none
    int32_t stray_d;    // its value, (event - model) in 1e-4 pA
    int32_t extra_mpos; // site of the one-event '+' window such a row opens (R5)
    uint8_t mode, rev, filtered, xflags;   // xflags: bit0 extra_multi, bit1 has the '+' window (its row = first - 1)
    uint32_t vf;        // the validation flags (V_*) the classification rests on: the table's (validated tables), or what the
                        // block's first rows say (first pass: the scan marks the pass if a later row says otherwise).
                        // In the template (nb_template(), mc_k0.hip): the number of segments of the block
    int32_t seq_delta;  // byte offset of the contig's sequence less 32 * mask_off (the reference's two layouts run side by side: a few
                        // bytes per contig), so that the base at a position is one load behind the descriptor, not two (R.seq_off[contig]
                        // first); NO_SEQ_DELTA: it does not fit, R.seq_off has to be asked
    __host__ __device__ int64_t first() const { return first_delta < 0 ? -1 : row_begin + first_delta; }
    __host__ __device__ int64_t extra_row() const { return (xflags & 2) ? row_begin + first_delta - 1 : -1; }
    __host__ __device__ bool extra_multi() const { return xflags & 1; }
};
static_assert(sizeof(NbDesc) == 64, "NbDesc layout");

struct DevTable {
    int64_t n_rows = 0;
    int32_t *pos = nullptr, *idx = nullptr;
    int2 *evmu = nullptr;     // (event, model) pairs as the parser wrote them: one DRAM page per window for k1_emit
    uint8_t *flags = nullptr;
    int2 *unit_pp = nullptr;  // [ceil(n_rows / 8)] positions of the first and the last row of every unit of eight rows (k_summarize,
                              // when a table is scanned a second time): all the filter of a repeated scan looks at -- 1 B/row
    int32_t n_seg = 0;
    int64_t *seg_begin = nullptr;
    int32_t *seg_read = nullptr, *seg_contig = nullptr;
    int32_t n_reads = 0;
    int32_t n_nb = 0;
    int64_t *nb_row_begin = nullptr;  // [n_nb+1]
    int32_t *nb_seg_begin = nullptr;  // [n_nb+1]
    int32_t *nb_read = nullptr;       // [n_nb]
    uint8_t *nb_repeat = nullptr;     // [n_nb] read id seen in an earlier name block
    uint32_t *nb_vflags = nullptr;    // [n_nb]
    NbDesc *nb_tmpl = nullptr;        // [n_nb] the pass-independent fields of the name-block descriptors (nb_template(), made by the first k0_first_site over the table)
    int64_t n_tiles = 0;
    int32_t *tile_nb = nullptr;       // [n_tiles] name block of the first row of every tile of the scan
    int has_repeats = 0;
};

struct DevRef {
    int32_t n_contigs = 0;
    int64_t *contig_len = nullptr, *seq_off = nullptr, *word_off = nullptr;
    uint8_t *seq = nullptr;
    uint32_t *mf = nullptr, *mr = nullptr;
    // site numbering for the per-site reduction: marked sites in (contig, strand, position) order
    int32_t *rank_f = nullptr, *rank_r = nullptr;   // [n_words] set bits of the contig's mask before this word
    int64_t *site_base = nullptr;                   // [2 * n_contigs] number of the first site of (contig, strand)
    int64_t n_sites = 0;
};

struct DevRecords {
    int64_t capacity = 0;
    double *feats = nullptr;
    int32_t *site_pos = nullptr, *site_seg = nullptr;
    int64_t *close_row = nullptr;
    uint32_t *info = nullptr;
    double *prob = nullptr;
    uint8_t *wmask = nullptr;   // bit s: slot mean s is not fl(d / 1e4) (travels as 64 bits); 0xFF: not looked at yet (k_pack looks)
};

struct DevMlp {
    int32_t n_models = 0, n_in = 0, n_hidden = 0;
    double *W1 = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr;
    double *wu = nullptr;      // per sub-model and hidden unit: W1[0..n_in)[j], b1[j], W2[j] -- what k2_mlp reads with scalar loads
    float *wp32 = nullptr;     // the fast forward (k2_mlp<.., true>): floats, the units in PAIRS -- [model][pair][n_in + 2][2]: the two
                               // units' weights side by side (one SGPR pair per packed fma); W1 and b1 times 2 log2(e) (tanh32s), W2
                               // plain; a layer with an odd number of units is padded with a unit of zeros
    float *margin = nullptr;   // per sub-model, MC_MAX_K + 2 floats: K0, K_0 .. K_{n_in-1} -- the fast forward's probability is within
                               // K0 + sum K_i |x_i| of the fp64 one (mc_ctx_set_mlp works it out from the weights)
    int fast = 0;              // flush records are scored by the fast forward (fp32 + fp64 where a printed digit could depend on it)
    uint8_t *sub_of_char = nullptr;
};
// largest |tanh32s(s) - tanh(s ln2 / 2)| over all floats s (mc_classify.hip: one v_exp_f32, one v_rcp_f32, one fma): measured by
// exhaustion on the GPU, tests/test_gpu_mlp_fast.py holds the kernel to it
constexpr double K2_TANH32_MAX_ERR = 2.5e-7;      // (measured: 2.180e-7)

struct DevForest {
    int32_t n_models = 0, n_in = 0;
    int32_t *model_tree_off = nullptr, *tree_node_off = nullptr, *left = nullptr, *right = nullptr, *feature = nullptr;
    double *threshold = nullptr, *value = nullptr;
    uint8_t *sub_of_char = nullptr;
};

// Closed-form classifiers (-c LR, -c NBC; train_model.py:55-60, scored at the same call site :199): per sub-model `stride`
// doubles.  MC_CLF_LOGISTIC: w[n_in], b -- p = expit(x . w + b) (scikit-learn's LogisticRegression, binary);
// MC_CLF_GNB: theta0[n_in], var0[n_in], theta1[n_in], var1[n_in], log prior0, log prior1 -- GaussianNB's joint log
// likelihoods, p = exp(jll1 - logsumexp(jll)).
struct DevSimple {
    int32_t kind = 0, n_models = 0, n_in = 0, stride = 0;
    double *params = nullptr;
    uint8_t *sub_of_char = nullptr;
};

// Payload-slot counters (k1_scan's TileSlots::reserve), one per tile & (NSHARD - 1), every one in a cache line of its own: a
// one-base motif has every tile fetch a chunk of slots, and 6*10^4 atomics on the eight counters of ONE line took 0.4 ms of the
// scan's 0.76 -- they are served one after the other, ~7 ns each, wherever in the line they land.  (Tables of fewer than 1024
// tiles use eight of them: every counter owns an equal share of the payload slots.)
constexpr int NSHARD = 64, SHARD_PAD = 16;

struct Counters {          // device-side status block
    unsigned long long n_records;
    unsigned int overflow;
    unsigned int violation;    // first pass over a table: a row contradicts what a regular block was classified on (the pass is
                               // repeated on the table's complete validation flags)
    unsigned int n_big;
    unsigned int n_rare;       // windows left to k1_rare
    unsigned long long n_kept; // records without MC_I_TOO_MANY (k_pack: rows of the compacted slot means / probabilities)
    unsigned long long n_wide; // slot means of those records that travel as 64 bits (k_pack: the others as 32-bit integers)
    unsigned long long side_done;  // workgroups of the side stream's kernel (k2_mlp<.., PACK>) that are through: the last one sends the counters to the host
    unsigned long long pack_need;  // bytes the packed block of the pass would take if it is more than the block has (set with `overflow`: the pass is
                                   // repeated, the next one gets a bigger block); 0: it fitted
    // the pass in which a name block was last classified irregular (mc_params-independent pass number, never 0).  Written,
    // never zeroed: k0_first_site classifies while it zeroes the other counters, so a count could lose updates -- a pass is
    // special iff this equals its own number
    unsigned long long irregular_pass;
    unsigned long long end_of_head;   // (k_pack copies everything before this field to the host)
    unsigned long long pad_to_line[SHARD_PAD];
    unsigned long long shard[NSHARD * SHARD_PAD];
};

// The small kernels on the ctx stream's critical path (strand resolve, tile descriptors, the ordering of the payloads) run
// beside the previous pass's classifier, whose waves keep the vector pipes busy: with the default wave priority the
// arbiter serves the older (classifier) waves first and these latency-bound kernels take twice as long.  (Not the scan and
// the emit: with the raised priority a pipelined pass takes 0.204 / 0.208 ms instead of 0.200.)
#define MC_FRONT_OF_THE_QUEUE __builtin_amdgcn_s_setprio(3)
#ifndef MC_SCAN_SUMMARY
#define MC_SCAN_SUMMARY 1
#endif
#ifndef MC_EVENTS_ON_KERNELS
#define MC_EVENTS_ON_KERNELS 1
#endif

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
// np.round(e - m, 4) == fl((E4 - M4) / 1e4) (:286) without the division: for EVERY int32 x the reciprocal-and-correct
// sequence below equals the IEEE quotient x / 10000.0 bit for bit (tests/tools/div1e4_check.c goes through all 2^32) --
// three fp64 operations instead of the ten of a division, eight times per slot and round in k1_emit.
__device__ __forceinline__ double div1e4(int x) {
    const double xd = (double)x, r = 1.0 / 10000.0;
    const double q0 = xd * r;
    return fma(fma(-q0, 10000.0, xd), r, q0);
}

// a slot mean that is fl(d / 1e4) for a 32-bit integer d travels as d (k_pack); the emit notes which are not (DevRecords.wmask)
__device__ __forceinline__ bool slot_is_narrow(double v, int32_t *d_out) {
    const double t = rint(v * 1e4);
    if (!(fabs(t) < 2147483648.0)) return false;            // (NaN too)
    const int32_t d = (int32_t)t;
    if (__double_as_longlong(div1e4(d)) != __double_as_longlong(v)) return false;     // bit for bit (-0.0 is wide); div1e4(d) == d / 1e4
    *d_out = d;
    return true;
}

__device__ __forceinline__ int first_m(const uint32_t *__restrict__ bits, int64_t L, int64_t pos, int k) {
    if (pos >= L) return -1;
    const int64_t w0 = pos >> 5;
    const uint64_t lo = bits[w0], hi = bits[w0 + 1];
    uint64_t w = ((hi << 32) | lo) >> (pos & 31);
    w &= (1ull << k) - 1ull;
    return w ? __builtin_ctzll(w) : -1;
}

__device__ __forceinline__ int bit_at(const uint32_t *__restrict__ bits, int64_t p) {
    return (int)((bits[p >> 5] >> (p & 31)) & 1u);
}

__device__ __forceinline__ unsigned char comp_char(unsigned char c) {
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        case 'N': return 'N';
        case 'M': return 'M';
        default: return 0xFF;
    }
}

constexpr uint32_t MC_I_BIG = 0x1000u;   // internal: a slot holds > 128 events, finished by k1_bigfix
constexpr uint32_t MC_I_HOLE = 0x2000u;  // internal: a record slot a piece of the fused dense pass (k1_fused) did not fill -- always with MC_I_TOO_MANY;
                                         // k_pack compacts the holes away, the host never sees one
constexpr int O_EXTRA = 14;              // meta nibble: the one-event '+' window of a palindromic f0 (R5)

struct RowSrc {   // the columns, for window walks
    const int32_t *g_pos;
    const int2 *g_evmu;
    const uint8_t *g_flags;
    bool stray_pending;  // the next value handed out is the block's stray event (R5), not a row
    double stray_val;
};

// rows are only ever walked inside the name block of a row that passed the quality filter, so "valid"
// (:167-168) reduces to model_kmer != NNNNNN
__device__ __forceinline__ double next_val(RowSrc &S, int64_t &cur) {
    if (S.stray_pending) {
        S.stray_pending = false;
        return S.stray_val;
    }
    for (;;) {
        const int64_t r = cur++;
        if (!(S.g_flags[r] & MC_F_MODEL_N)) {
            const int2 e = S.g_evmu[r];
            return (double)(e.x - e.y) / 10000.0;                 // np.round(e-m,4) == fl((E4-M4)/1e4)  (:286)
        }
    }
}

// NumPy pairwise_sum over the next n values, n <= 128 (np.mean, :186): n < 8 sequential from -0.0;
// else eight strided accumulators over the first n - n%8 values, combined pairwise, tail added in order.
// (Values are never -0.0 -- they are integer/1e4 -- so starting the accumulators at +0.0 is exact.)
__device__ __forceinline__ double leaf_sum(RowSrc &S, int64_t &cur, int n) {
    const int n8 = n < 8 ? 0 : n - (n % 8);
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, r5 = 0.0, r6 = 0.0, r7 = 0.0;
    for (int i = 0; i < n8; ++i) {
        const double v = next_val(S, cur);
        switch (i & 7) {
            case 0: r0 += v; break;
            case 1: r1 += v; break;
            case 2: r2 += v; break;
            case 3: r3 += v; break;
            case 4: r4 += v; break;
            case 5: r5 += v; break;
            case 6: r6 += v; break;
            default: r7 += v; break;
        }
    }
    double res = n8 ? ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) : -0.0;
    for (int i = n8; i < n; ++i) res += next_val(S, cur);
    return res;
}

// The copy-out's packing (k_pack, below) takes the records in PACK_WGS chunks; the kept records and wide slots of every chunk
// are counted where the records are written (count_for_packing): no pass over the records for the counts alone.
#ifndef MC_PACK_WGS
#define MC_PACK_WGS 512
#endif
#ifndef MC_PACK_THREADS
#define MC_PACK_THREADS 512
#endif
constexpr int PACK_WGS = MC_PACK_WGS, PACK_THREADS = MC_PACK_THREADS;

// chunk_cnt[PACK_PAD * b], chunk_cnt[PACK_PAD * b + 1]: kept records of chunk b (calls: no MC_I_TOO_MANY) and their slot means
// that are not fl(d / 1e4) -- every chunk's pair in a cache line of its own: the emit's waves work on neighbouring records, and
// atomics on one line are served one after the other (all 418 records of a chunk, sixteen chunks to a line: k1_emit 40 -> 139 us).
constexpr int PACK_PAD = 16;

// One record (q of n_rec) for the packing's counts.  Called by one lane per record (the rare paths).
__device__ __forceinline__ void count_for_packing(unsigned long long *chunk_cnt, int64_t q, int64_t n_rec, bool kept, int n_wide) {
    if (!chunk_cnt || !kept) return;
    const uint32_t per = (uint32_t)((n_rec + PACK_WGS - 1) / PACK_WGS);
    const uint32_t b = (uint32_t)q / per;
    atomicAdd(&chunk_cnt[PACK_PAD * b], 1ull);
    if (n_wide) atomicAdd(&chunk_cnt[PACK_PAD * b + 1], (unsigned long long)n_wide);
}

// ... the records of a wave, all lanes calling: kept (one lane per record), q, wmask (bit f: slot mean f is wide) of the lane's
// record.  One pair of atomics for the wave unless its records straddle a chunk boundary.
__device__ __forceinline__ void count_wave_for_packing(unsigned long long *chunk_cnt, int64_t n_rec, bool kept, int64_t q, uint32_t wmask, int k) {
    if (!chunk_cnt) return;
    const unsigned long long km = __ballot(kept);
    if (!km) return;
    uint32_t per = (uint32_t)((n_rec + PACK_WGS - 1) / PACK_WGS);
    // (the division is made here, every time: its reciprocal, worked out once in front of the caller's loop, is one more value the
    // caller's registers do not hold -- it is spilled, and the reload waits for every store of the round)
    asm volatile("" : "+s"(per));
    const uint32_t bq = kept ? (uint32_t)q / per : 0u;
    const int first = __ffsll((unsigned long long)km) - 1;
    const uint32_t b0 = (uint32_t)__shfl((int)bq, first);
    if (!__ballot(kept && bq != b0)) {
        int nw = 0;
        for (int f = 0; f < k; ++f) nw += __popcll(__ballot(kept && ((wmask >> f) & 1u)));
        if ((int)(threadIdx.x & 63) == first) {
            atomicAdd(&chunk_cnt[PACK_PAD * b0], (unsigned long long)__popcll(km));
            if (nw) atomicAdd(&chunk_cnt[PACK_PAD * b0 + 1], (unsigned long long)nw);
        }
    } else if (kept) {
        atomicAdd(&chunk_cnt[PACK_PAD * bq], 1ull);
        if (wmask) atomicAdd(&chunk_cnt[PACK_PAD * bq + 1], (unsigned long long)__popc(wmask));
    }
}

// What k1_scan hands to k1_emit per closed window (arrival order; k1_list maps file order onto it)
constexpr uint32_t PF_EXTRA = 1, PF_CLOSE_NS = 2, PF_MULTI = 8, PF_REV = 16, PF_STRAY = 32;
constexpr int WROWS = 64;   // rows before a window's last row that k1_emit looks at (longer windows: k1_rare)
constexpr int FRONT = 64;   // rows of padding in front of the pos / flags / pair columns, so that a look-back never leaves them

struct __attribute__((aligned(16))) Payload {
    int64_t r;          // last row of the window
    int64_t close_row;  // row that closes it (:179); n_rows: in the next shard; -1 cannot occur (not emitted)
    int32_t m;          // the site
    int32_t close_pos;
    uint32_t flags;     // PF_*
    int32_t nb;         // name block
};
static_assert(sizeof(Payload) == 32, "Payload layout");

constexpr int MC_SIDE_MAX_ROOM = 512;   // the side stream's one kernel takes the pieces of a fused dense pass if their room is below this (a stretch of its records)

struct K1Args {
    DevTable T;
    DevRef R;
    const NbDesc *desc;
    Payload *payload;             // [payload_cap]
    long long payload_cap;
    long long *tile_chunk;        // [n_tiles * NCHUNK] first payload slot of the tile's chunks of (1 << chunk_shift) behind its own PT slots
    int chunk_shift;              // 6: chunks of 64 (sparse motifs); 8: chunks of 256 (a one-base motif: ~270 windows per tile)
    int shard_shift;              // the chunk counters in use: 1 << shard_shift
    int shard_mask;               // ... less one
    int32_t *tile_cnt;            // [n_tiles] windows closed in the tile
    int32_t *tile_half;           // [n_tiles] ... in its first chunk (the one-base-motif scan: k1_emit_runs' pieces are the scan's chunks)
    const int32_t *tile_local;    // [n_tiles] exclusive scan of tile_cnt inside its group of 1024 tiles
    const int64_t *group_sum;     // [n_groups] windows per group
    int64_t *tile_first;          // [n_tiles] first record of the tile (k1_list writes it for k1_emit_runs: group prefix + tile_local)
    DevRecords O;                 // records, file order
    Counters *cnt;
    int k, skip_thresh, tail_contig;
    int64_t *rare_list;           // [capacity] records k1_emit leaves to k1_rare
    unsigned long long pass_no;   // what Counters.irregular_pass is set to when the pass cannot be finished by the fast path
    int32_t *piece_cnt;           // the fused dense pass (k1_fused): [pieces] records of every piece
    int32_t *piece_kw;            // ... [pieces] records | of them calls (no MC_I_TOO_MANY) << 9 | their wide slot means << 18 -- the packing's
                                  // counts, made where the records are written (windows left to the row-by-row walk: predicted,
                                  // mc_rows.h); rooms below MC_SIDE_MAX_ROOM slots only (the fields fit); nullptr: nobody packs by piece
    unsigned long long *chunk_cnt;  // pipelined passes: [PACK_PAD * PACK_WGS] kept records / their wide slots per chunk of the copy-out's packing
                                  // (k_pack), counted by the emit itself as it writes the records; nullptr: nobody packs (or k_pack_count counts)
};

// The row that closes a window whose last row is r (in name block nb_abs, which ends at my_end): the next
// unfiltered row in the file (:179).  Returns its index (T.n_rows when it lies in the next shard, -1 when there is
// none: the window is lost at EOF, R6).
__device__ __forceinline__ int64_t find_close(const DevTable &T, const NbDesc *__restrict__ desc, int tail_contig,
                                              int nb_abs, int64_t my_end, int64_t r, int &close_pos, bool &close_ns) {
    int64_t rr = r + 1;
    int bb = nb_abs;
    close_ns = false;
    close_pos = 0;
    while (rr < T.n_rows) {
        if (rr < my_end) {                          // still my name block: valid <=> not an N row
            if (!(T.flags[rr] & MC_F_MODEL_N)) { close_pos = T.pos[rr]; return rr; }
            ++rr;
            continue;
        }
        close_ns = true;                            // another read begins: closes whatever its position
        while (bb + 1 < T.n_nb && T.nb_row_begin[bb + 1] <= rr) ++bb;
        if (desc[bb].filtered) { rr = T.nb_row_begin[bb + 1]; continue; }   // skip the read whole
        if (!(T.flags[rr] & MC_F_MODEL_N)) { close_pos = T.pos[rr]; return rr; }
        ++rr;
    }
    close_ns = true;
    return tail_contig >= 0 ? T.n_rows : -1;
}

constexpr int PT = 16;              // payload slots reserved per tile; further chunks of 64 come from an atomic
constexpr int NCHUNK = TILE / 64 + 1; // ... at most this many of them (one window per row, and one more per block start)
#ifndef MC_CHUNK
#define MC_CHUNK 1024
#endif
constexpr int CHUNK = MC_CHUNK;     // k1_scan: rows a wave holds in registers at a time; it takes its tile chunk after chunk
constexpr int NCH = TILE / CHUNK;
constexpr int NQ = CHUNK / 512;     // ... stripes of 512 rows per chunk -- every lane holds EIGHT consecutive rows of a stripe
static_assert(CHUNK % 512 == 0 && TILE % CHUNK == 0, "whole stripes, whole chunks");

constexpr int GROUP = 1024;

// first record slot of a tile; wave-uniform call (all 64 lanes), n_groups <= a few hundred
__device__ __forceinline__ int64_t tile_slot(const int32_t *__restrict__ tile_local, const int64_t *__restrict__ group_sum,
                                             int64_t tile, int lane) {
    const int g = (int)(tile / GROUP);
    long long part = 0;
    for (int i = lane; i < g; i += 64) part += group_sum[i];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    return part + tile_local[tile];
}

constexpr int EG = 8;            // lanes per window (k1_emit)
constexpr int K2_MAXM = 8;       // sub-models k2_mlp lists (mc_ctx_set_mlp refuses more)

struct PackLayout { size_t pos, seg, info, feats; };       // byte offsets in the block (the closing rows come first)
__host__ __device__ inline PackLayout pack_layout(int64_t n, int close32) {
    PackLayout L;
    L.pos = (size_t)n * (close32 ? 4 : 8);
    L.seg = L.pos + (size_t)n * 4;
    L.info = L.seg + (size_t)n * 4;
    L.feats = (L.info + (size_t)n * 4 + 7) & ~(size_t)7;
    return L;
}

// The slot means of the m calls behind the narrow columns.  A slot mean is very often fl(d / 10^4) for an integer d -- every
// slot that holds ONE event is (its value is fl((E4 - M4) / 10^4), section 2 of DESIGN.md), 53 % of the slots of the headline
// workload -- and then d travels, 32 bits, and the host divides again (IEEE division: the same double); the others travel
// as they are, their low half in the slot's place and their high half in a compact list behind, one bit per slot says which.
//   lo32[m * k] | prob[m] (f64) | wide mask[m] (u8, bit s: slot s is 64 bits wide) | hi32[n_wide]
struct PackTail { size_t lo32, prob, wmask, hi32, end; };
__host__ __device__ inline PackTail pack_tail(size_t feats_off, size_t m, int k, size_t n_wide) {
    PackTail T;
    T.lo32 = feats_off;
    T.prob = (T.lo32 + m * (size_t)k * 4 + 7) & ~(size_t)7;
    T.wmask = T.prob + m * 8;
    T.hi32 = (T.wmask + m + 3) & ~(size_t)3;
    T.end = T.hi32 + n_wide * 4;
    return T;
}

struct LitArgs {
    DevTable T;
    DevRef R;
    const NbDesc *desc;
    const int64_t *nb_f0;
    const double *qual;
    double qual_thresh;
    int k, skip_thresh, tail_contig, entry_read, entry_first_idx;
    int32_t *run_cnt;          // [n_nb] flush records of the run that starts at this block (0 elsewhere)
    int32_t *run_rows;         // [n_nb] rows of that run
    const int32_t *cnt_local;  // exclusive scans of the two (inside groups of 1024 blocks) + group sums
    const int64_t *cnt_group;
    const int32_t *rows_local;
    const int64_t *rows_group;
    double *scratch;           // MC_MAX_K slot arrays per run, each as long as the run
    DevRecords L;
    int write;                 // 0: count records and rows per run; 1: produce the records
};

// NumPy pairwise_sum over an array (np.mean of a slot list, :186)

constexpr int SCAN_VALIDATE = 0;    // a table's first pass: positions, event indices and flag bytes streamed (9 B/row), every row validated
constexpr int SCAN_STREAM = 1;      // a validated table, positions and flag bytes streamed (5 B/row): one-base motifs, where every unit is listed
constexpr int SCAN_SUMMARY = 2;     // a validated table that has unit summaries (k_summarize): 1 B/row


// ---- what the kernel units hand to the host side (mc_stream.hip): every kernel with its launch geometry, on stream st; `stop`:
//      an event that rides on the kernel's own dispatch packet (hipExtLaunchKernelGGL), or nullptr ----
void mc_launch_first_site(const DevTable &T, const DevRef &R, const double *qual, double qual_thresh, int k, NbDesc *desc, int64_t *nb_f0,
                          Counters *cnt, int classify, int skip_thresh, unsigned long long pass_no, int hyp, int make_tmpl, hipStream_t st);
void mc_launch_classify(const DevTable &T, const DevRef &R, NbDesc *desc, const int64_t *nb_f0, int entry_read, int k, int skip_thresh,
                        Counters *cnt, unsigned long long pass_no, hipStream_t st);
void mc_launch_extend(const DevTable &T, NbDesc *desc, const int64_t *nb_f0, int entry_read, Counters *cnt, unsigned long long pass_no,
                      hipStream_t st);
void mc_launch_summarize(const DevTable &T, hipStream_t st);
void mc_launch_scan(const K1Args &A, bool dense, int scan_mode, hipStream_t st);
void mc_launch_group_scan(const int32_t *cnt, int64_t n, int32_t *local, int64_t *group_sum, hipStream_t st);
void mc_launch_list(const K1Args &A, Payload *sorted, int gather, hipStream_t st, hipEvent_t stop = nullptr);
int mc_emit_occupancy(void);            // resident k1_emit workgroups per CU (occupancy query)
void mc_launch_emit(const K1Args &A, const Payload *sorted, unsigned grid, hipStream_t st, hipEvent_t stop, hipEvent_t start = nullptr);
void mc_launch_emit_runs(const K1Args &A, Payload *sorted, hipStream_t st, hipEvent_t stop);
// the fused pass of a dense reference (mc_fused.hip): room per piece, pieces of a table, the launch
int mc_fused_room(double density);
int mc_fused_room_max(void);
int64_t mc_fused_pieces(const DevTable &T);
void mc_launch_fused(const K1Args &A, Payload *sorted, int cap, bool validate, hipStream_t st, hipEvent_t stop);
void mc_launch_rare(const K1Args &A, const Payload *sorted, const int64_t *rare_list, int64_t n_rare, hipStream_t st);
void mc_launch_rare_dev(const K1Args &A, const Payload *sorted, const int64_t *rare_list, hipStream_t st);
void mc_launch_bigfix(const K1Args &A, int64_t n, hipStream_t st);
void mc_launch_literal(const LitArgs &LA, unsigned grid, hipStream_t st);
void mc_launch_merge(const DevRecords &O, int64_t n_o, const DevRecords &L, int64_t n_l, const DevRecords &M, int k, hipStream_t st);
void mc_launch_classifier(const DevMlp &M, const DevForest &F, const DevSimple &S, int n_cu, hipStream_t st, const double *feats, int k,
                          const int32_t *site_seg, const int32_t *seg_read, const double *qual, const uint32_t *info,
                          const uint8_t *submodel_in, int64_t n, double *prob, const unsigned long long *n_dev, const unsigned int *overflow,
                          const int32_t *piece_cnt = nullptr, int piece_room = 0, int64_t n_pieces = 0);
void mc_launch_pack_count(const DevRecords &O, const Counters *cnt, int k, unsigned long long *chunk_cnt, int holes, hipStream_t st);
void mc_launch_pack(const DevRecords &O, const Counters *cnt, const unsigned long long *chunk_cnt, unsigned char *out, size_t out_bytes, int k,
                    int close32, Counters *host_status, int holes, int look, hipStream_t st, hipEvent_t stop);
bool mc_launch_side(const DevMlp &M, bool other_classifier, int n_cu, hipStream_t st, const K1Args &A, const Payload *sorted, const int32_t *seg_read,
                    const double *qual, int64_t cap, int score, unsigned char *out, size_t out_bytes, int close32, Counters *host_status,
                    int piece_room, int64_t n_pieces, hipEvent_t stop);

// ---- rows of text on the device (mc_rowtext.hip) ----
// a segment as the device parser leaves it (mc_devparse.inc): first row, where its read name stands in the shard's text, contig
struct KpSeg { long long row; long long name_off; int name_len; int contig; int name_start; int pad; };

struct RowTextStatus {          // what the row writer leaves for the host (a pinned copy travels with the text)
    unsigned long long n_bytes; // the rows' text (too_small: what it would have taken)
    unsigned long long n_rows;
    unsigned int host_needed;   // a record the device does not print: the shard's rows come from the host formatter
    unsigned int too_small;     // the text did not fit the room it was given
};

struct RowTextIn {
    const unsigned char *pack;  // the pass's packed records on the device (pack_layout / pack_tail)
    int64_t n, m, n_wide;       // records, call rows, wide slot means
    int k, close32;
    const int64_t *seg_begin;   // the pass's table: [n_seg + 1]
    const int32_t *seg_read, *seg_contig;
    int32_t n_seg;
    const KpSeg *segs;          // [n_seg] (the device parser's)
    const char *text;           // the shard's text
    const double *qual;         // read qualities the pass ran with
    int32_t n_qual;
    DevRef R;
    const uint32_t *cn_off, *cn_len;    // contig names (KpContigs)
    const char *cn_chars;
    const uint8_t *sub_of_char;
    int32_t tail_contig;
    char lab_meth[8], lab_unmeth[8];
    int lab_meth_len, lab_unmeth_len;
};

struct RowTextScratch {
    uint32_t *kept_blk, *wide_blk;      // [record blocks + 1], [call-row blocks + 1]
    uint32_t *wide_pref;                // [call rows]
    uint32_t *rec_len, *rec_row;        // [records]
    unsigned long long *len_blk;        // [record blocks + 1]
    double *wval;                       // [wide slot means] the 64-bit slot means, in the order the rows hold them
    unsigned long long *num_lo;         // [wide slot means + reads] their printed digits, and the read qualities' (mc_rowtext.h RtNum, packed)
    uint32_t *num_meta;
    RowTextStatus *st;
};

void mc_row_text_scratch_sizes(int64_t n, int64_t m, int64_t *rec_blocks, int64_t *row_blocks);
void mc_launch_row_text(const RowTextIn &I, const RowTextScratch &S, char *out, size_t out_cap, char *out_host, RowTextStatus *st_host,
                        hipStream_t st);

#endif  // MC_DEV_H
