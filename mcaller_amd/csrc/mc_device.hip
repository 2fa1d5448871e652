// libmcaller_hip.so -- device side (gfx950 / MI355X).  C ABI: include/mcaller_hip.h.
//
// The reference's hot path (extract_contexts.py:147-291 + :199) as HIP kernels over a columnar event
// table resident in HBM:
//
//   text         kp_count / kp_scan / kp_starts / kp_parse / kp_count_rows / kp_place   the eventalign TEXT of a streamed shard
//                                parsed on the device (mc_devparse.inc): line starts, tokens, numbers, name blocks and segments,
//                                the columns written straight into a table slot
//   per table     (nothing runs at upload: the first pass over a table validates it while it scans)
//                k_nb_template   the pass-independent fields of the name-block descriptors
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
//                k1_rare_dev     windows longer than 64 rows, row by row
//                k2_mlp          batched 7-H-1 tanh/logistic forward in fp64: one lane per record, weights as scalar operands,
//                                a quarter of the hidden units per SIMD
//                k3_forest       random-forest predict_proba;  k_literal / k_merge  irregular reads, row by row
//                k_site_counts   per-site reduction (+ ncclAllReduce);  k_pack  record columns packed for the copy-out
//                k_copy_bytes    small transfers by the compute units (the DMA engines serialise behind queued text)
//
// Equivalence with the sequential machine on regular blocks (one contig, positions non-decreasing, event
// index monotone in the direction the first site row implies, no site at contig position 0, read name not
// seen before) is argued in DESIGN.md; every other block is classified irregular and handled by the
// literal per-run kernel (k_literal) so results never depend on a CPU path.
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

namespace {

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
                        // In the template (k_nb_template): the number of segments of the block
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
    NbDesc *nb_tmpl = nullptr;        // [n_nb] the pass-independent fields of the name-block descriptors (k_nb_template)
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
    uint8_t *sub_of_char = nullptr;
};

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

// ---------------------------------------------------------------------------------------------------
// per-table kernels
// ---------------------------------------------------------------------------------------------------
// k_summarize: the unit summaries of a table that is scanned again (other parameters; a resident table): first and last
// position of every unit of eight rows, 1 B/row -- what the filter of a repeated scan reads instead of the columns.  A flat
// stream over the positions; the thread that holds the first or the second half of a unit writes one word, consecutive
// threads consecutive words.  (A table that is scanned once never pays for this: its scan streams the columns themselves.)
__global__ __launch_bounds__(256) void k_summarize(DevTable T) {
    const int64_t g4 = blockIdx.x * (int64_t)256 + threadIdx.x;       // group of four rows (the columns are padded to whole tiles)
    if (g4 * 4 >= T.n_rows) return;
    const int4 p = *reinterpret_cast<const int4 *>(T.pos + g4 * 4);
    reinterpret_cast<int32_t *>(T.unit_pp)[g4] = (g4 & 1) ? p.w : p.x;
}

// ---------------------------------------------------------------------------------------------------
// K0: strand resolve
// ---------------------------------------------------------------------------------------------------
// One wave per name block.  Under the reference's rule for a read it has not seen a site row of yet
// (`read_name != last_read`, :161-174) each unfiltered row is tested on the strand `rev = (col3 != col10)`;
// the first row that holds an 'M' in its k-mer becomes the block's first site row f0.
// (the fields of a descriptor that do not depend on the pass -- rows, contig, mask offset, read -- are prepared once per
// table/reference by k_nb_template, so that a pass reads one 64-byte line per block instead of walking five tables)
__global__ void k_nb_template(DevTable T, DevRef R) {
    const int b = (int)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (b >= T.n_nb) return;
    NbDesc d;
    d.row_begin = T.nb_row_begin[b];
    d.row_end = T.nb_row_begin[b + 1];
    d.contig = T.seg_contig[T.nb_seg_begin[b]];
    d.mask_off = R.word_off[d.contig];
    d.first_delta = -1;
    d.contig_len = (int32_t)R.contig_len[d.contig];
    d.read = T.nb_read[b];
    d.stray_q = NO_STRAY;
    d.stray_d = 0;
    d.extra_mpos = 0;
    d.mode = MODE_NONE;
    d.rev = 0;
    d.filtered = 0;
    d.xflags = 0;
    d.vf = (uint32_t)(T.nb_seg_begin[b + 1] - T.nb_seg_begin[b]);      // segments (contigs) of the block
    T.nb_tmpl[b] = d;
}

// NS stripes of 64 rows from `base`: all loads of the round are issued before any is used.  -> first site row or -1
// NS stripes of 64 rows from `base`: their flags and positions ...
template <int NS>
__device__ __forceinline__ void first_site_rows(const DevTable &T, int64_t base, int64_t se, int lane, uint32_t (&fl)[NS], int (&ps)[NS]) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int64_t r = base + u * 64 + lane;
        const int64_t ra = r < se ? r : se - 1;
        fl[u] = r < se ? (uint32_t)T.flags[ra] : (uint32_t)MC_F_MODEL_N;
        ps[u] = T.pos[ra];
    }
}

// ... and the first site row among them (-1: none).  `prefetch` runs after the mask loads have been issued and before they are
// waited for: the next round's rows travel with this round's masks.
template <int NS, typename Prefetch>
__device__ __forceinline__ int64_t first_site_among(const uint32_t *__restrict__ mf, const uint32_t *__restrict__ mr, int64_t L, int64_t base,
                                                    int k, const uint32_t (&fl)[NS], const int (&ps)[NS], int &f0rev, Prefetch prefetch) {
    uint64_t wm[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {               // (the mask of the strand the row is tested on: its flag came with its position)
        const int64_t p = ps[u] < L ? ps[u] : 0;
        const int64_t w0 = p >> 5;
        const uint32_t *__restrict__ mm = (fl[u] & MC_F_KMER_EQ) ? mf : mr;
        wm[u] = ((uint64_t)mm[w0 + 1] << 32) | mm[w0];
    }
    prefetch();
    int64_t f0 = -1;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int rev = (fl[u] & MC_F_KMER_EQ) ? 0 : 1;
        uint64_t w = wm[u] >> (ps[u] & 31);
        w &= (1ull << k) - 1ull;
        const bool c = !(fl[u] & MC_F_MODEL_N) && ps[u] < L && w != 0ull;
        const unsigned long long mask = __ballot(c);
        if (mask && f0 < 0) {
            const int first = __builtin_ctzll(mask);
            f0 = base + u * 64 + first;
            f0rev = __shfl(rev, first);
        }
    }
    return f0;
}

// (also zeroes the pass's counters: nothing in here uses them, every later kernel of the pass does.  hipMemsetAsync would
// do too, but the runtime's fill ends with a system-scope release, and that release waits behind the PCIe writes of a
// copy-out running on the other stream)
// (the body of the classification: block b with descriptor d and first site row f0; lookback: the block may see
// name == last_read, i.e. the table repeats read names or continues a previous shard's read)
__device__ __forceinline__ void classify_block(const DevTable &T, const DevRef &R, NbDesc &d, int b, int64_t f0, uint32_t vf, bool lookback,
                                               const int64_t *__restrict__ nb_f0, int entry_read, int k, int skip_thresh,
                                               Counters *cnt, unsigned long long pass_no) {
    // `last_read` when the block starts = name of the latest earlier block that has a site row (:282)
    bool h1 = false;
    if (lookback && (T.nb_repeat[b] || entry_read >= 0)) {
        int j = b - 1;
        while (j >= 0 && nb_f0[j] < 0) --j;
        const int last_read = j >= 0 ? T.nb_read[j] : entry_read;
        h1 = (last_read == d.read);
    }
    uint8_t mode = MODE_NONE;
    if (d.filtered) {
        mode = MODE_NONE;                                 // every row fails :167, nothing else reads them
    } else if (h1) {
        mode = MODE_IRREGULAR;                            // rows see name == last_read: literal machine
    } else if (f0 >= 0) {
        bool regular = !(vf & (V_POS_DEC | V_IDX_EQ | V_MULTI_SEG));
        const bool inc = vf & V_IDX_INC, dec = vf & V_IDX_DEC;
        if (inc && dec) regular = false;
        // rows after f0 take rev = !(idx > idx[f0]) (:169)
        if (inc && d.rev) regular = false;
        const uint32_t *mf = R.mf + d.mask_off, *mr = R.mr + d.mask_off;
        const int64_t L = d.contig_len;
        if ((vf & V_POS0) && L > 0 && ((mf[0] | mr[0]) & 1u)) { regular = false; d.xflags |= 4; }   // falsy mpos (:179,:272,:279)
        if (regular && dec && !d.rev) {
            if (k < 2) {
                regular = false;
            } else {
                // palindromic first site row of a reverse read
                const int p = T.pos[f0];
                const int o_f = first_m(mf, L, p, k);
                const int mpos_f = p + o_f;
                int64_t r1 = f0 + 1;
                const int64_t re = T.nb_row_begin[b + 1];
                while (r1 < re && (T.flags[r1] & MC_F_MODEL_N)) ++r1;
                d.first_delta = (int32_t)(f0 + 1 - d.row_begin);
                d.rev = 1;
                { const int2 e0 = T.evmu[f0]; d.stray_d = e0.x - e0.y; }
                d.extra_mpos = mpos_f;
                if (r1 >= re) {
                    d.xflags |= 2;                          // closed by the next read (or lost at EOF)
                } else {
                    const int p1 = T.pos[r1];
                    const int o_r = first_m(mr, L, p1, k);
                    if (p1 >= mpos_f + 1) {
                        d.xflags |= 2;
                        if (o_r >= 0 && p1 <= mpos_f + skip_thresh + 1) {
                            if (o_r != 0) d.xflags |= 1;
                            if (p1 + o_r - p < k) d.stray_q = p;
                        }
                    } else if (o_r >= 0) {
                        d.stray_q = p1 + o_r - o_f;
                    }
                }
            }
        }
        mode = regular ? MODE_REGULAR : MODE_IRREGULAR;
    }
    d.mode = mode;
    if (mode == MODE_IRREGULAR) *reinterpret_cast<volatile unsigned long long *>(&cnt->irregular_pass) = pass_no;
}

// classify != 0: the block is classified here as well (classify_block, by the wave's first lane) -- for tables without
// repeated read names that do not continue a previous shard's read, where no block looks at another block's result; the
// separate k0_classify launch is then skipped.
// hyp != 0: no pass has validated the table yet (its first pass is this one).  The validation flags the classification needs
// -- which way the event index runs, whether position 0 occurs -- are then taken from the block's first rows: in a regular
// block every pair of rows says the same as the first pair, and positions do not decrease, so position 0 can only be the
// first row's.  The pass's scan compares EVERY row with the row before it (k1_scan, validate_units) and marks the pass if a row of
// a block classified regular says otherwise; whatever the first rows say is true of the block, so it is OR-ed into the
// table's flags here and the scan adds the rest: after the pass the table's flags are complete.
// W: waves per name block.  The kernel's time is its slowest block -- a read that starts in a long stretch without a site needs
// round after round of rows, each a dependent trip to memory: with W = 4 a round is 2048 rows (one block in seventy needs a
// second one; with 512 rows one in three did, and the slowest of 4000 blocks needed eight).  Tables of short reads (more blocks
// than there are rounds to save) take W = 1, four blocks per workgroup.
#ifndef MC_K0_WAVES
#define MC_K0_WAVES 1
#endif
template <int W>
__global__ __launch_bounds__(W == 1 ? 256 : 64 * W) void k0_first_site(DevTable T, DevRef R, const double *__restrict__ qual, double qual_thresh, int k,
                              NbDesc *__restrict__ desc, int64_t *__restrict__ nb_f0, Counters *__restrict__ cnt, int classify,
                              int skip_thresh, unsigned long long pass_no, int hyp) {
    static_assert(W == 1 || W == 2 || W == 4, "one wave per block, or a workgroup of two or four");
    __shared__ long long s_f0[4];
    __shared__ int s_rev[4];
    MC_FRONT_OF_THE_QUEUE;
    if (blockIdx.x == 0) {             // (everything but the pass mark, which is only ever written)
        unsigned int *w = reinterpret_cast<unsigned int *>(cnt);
        for (unsigned i = threadIdx.x; i < offsetof(Counters, irregular_pass) / 4; i += blockDim.x) w[i] = 0u;
        for (unsigned i = threadIdx.x; i < NSHARD; i += blockDim.x) cnt->shard[i * SHARD_PAD] = 0ull;
    }
    const int b = W == 1 ? (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6) : (int)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = W == 1 ? 0 : (int)(threadIdx.x >> 6);
    if (b >= T.n_nb) return;
    NbDesc d = T.nb_tmpl[b];
    const int n_seg = (int)d.vf;
    // (the quality decides whether the rows are looked at at all, but its load is not waited for before theirs go out: the
    // block's latency is a chain of dependent loads, and this removes one link)
    const double q_read = qual[d.read];
    // (for the classification: in flight with everything else)
    uint32_t vf;
    if (hyp) {
        const int64_t r1 = d.row_begin + 1 < d.row_end ? d.row_begin + 1 : d.row_begin;
        const int p_first = T.pos[d.row_begin], i_first = T.idx[d.row_begin], i_second = T.idx[r1];
        vf = n_seg > 1 ? V_MULTI_SEG : 0u;
        if (r1 != d.row_begin) vf |= i_second > i_first ? V_IDX_INC : (i_second < i_first ? V_IDX_DEC : V_IDX_EQ);
        if (p_first == 0) vf |= V_POS0;
    } else vf = T.nb_vflags[b];
    int64_t f0 = -1;
    int f0rev = 0;
    {
        const int seg0 = n_seg == 1 ? 0 : T.nb_seg_begin[b];
        for (int si = 0; si < n_seg && f0 < 0; ++si) {
            int64_t L, sb, se;
            const uint32_t *mf, *mr;
            if (n_seg == 1) {                      // the usual case: everything is in the template
                L = d.contig_len; sb = d.row_begin; se = d.row_end;
                mf = R.mf + d.mask_off; mr = R.mr + d.mask_off;
            } else {
                const int seg = seg0 + si;
                const int contig = T.seg_contig[seg];
                L = R.contig_len[contig];
                mf = R.mf + R.word_off[contig]; mr = R.mr + R.word_off[contig];
                sb = T.seg_begin[seg]; se = T.seg_begin[seg + 1];
            }
            // A round is two dependent loads (rows, then their mask words); one wave per block: the rows of the round after it are
            // requested together with the mask words, so every further round costs ONE.
            constexpr int FS = 8;
            uint32_t fl[FS], fl_next[FS];
            int ps[FS], ps_next[FS];
            if (W == 1) {
                if (sb < se) first_site_rows<FS>(T, sb, se, lane, fl, ps);
                for (int64_t base = sb; base < se && f0 < 0; base += 64 * FS) {
                    const bool more = base + 64 * FS < se;
                    f0 = first_site_among<FS>(mf, mr, L, base, k, fl, ps, f0rev,
                                              [&]() { if (more) first_site_rows<FS>(T, base + 64 * FS, se, lane, fl_next, ps_next); });
#pragma unroll
                    for (int u = 0; u < FS; ++u) { fl[u] = fl_next[u]; ps[u] = ps_next[u]; }
                }
            } else {
                for (int64_t base = sb; base < se && f0 < 0; base += 64 * FS * W) {        // (the same rounds for all four waves)
                    const int64_t mine = base + (int64_t)wave * 64 * FS;
                    long long f = -1;
                    int frev = 0;
                    if (mine < se) {
                        first_site_rows<FS>(T, mine, se, lane, fl, ps);
                        f = first_site_among<FS>(mf, mr, L, mine, k, fl, ps, frev, []() {});
                    }
                    if (lane == 0) { s_f0[wave] = f; s_rev[wave] = frev; }
                    __syncthreads();
#pragma unroll
                    for (int w = W - 1; w >= 0; --w)
                        if (s_f0[w] >= 0) { f0 = s_f0[w]; f0rev = s_rev[w]; }      // (the first in row order)
                    __syncthreads();
                }
            }
        }
    }
    const bool filtered = q_read < qual_thresh;
    if (filtered) { f0 = -1; f0rev = 0; }
    if (lane == 0 && wave == 0) {
        d.first_delta = f0 >= 0 ? (int32_t)(f0 - d.row_begin) : -1;
        d.rev = (uint8_t)f0rev;
        d.filtered = filtered ? 1 : 0;
        d.vf = vf;
        if (hyp && vf) atomicOr(&T.nb_vflags[b], vf);
        nb_f0[b] = f0;
        if (classify) classify_block(T, R, d, b, f0, vf, false, nb_f0, -1, k, skip_thresh, cnt, pass_no);
        desc[b] = d;
    }
}

// One thread per name block: is the block regular?
//
// Regular = the sequential machine reduces to the local window rule: the read name is new
// (`last_read` differs when the block starts), one contig, positions non-decreasing, event indices strictly
// monotone, and every row after the first site row f0 takes the strand f0 was tested on.  One irregularity is
// common enough (~1 % of reads) to be folded into the fast path exactly: a reverse read whose f0 is a
// reverse-complement-palindromic k-mer (R5).  f0 is then scored on '+', opening a one-event '+' window; the
// rows after it are all '-' (event index decreasing, :169).  What the machine does with that event depends only
// on the next unfiltered row r1 (:179, :242-256, :272-279):
//   pos(r1) >  site of the '+' window: the window is flushed with k-1 empty slots (a too-many-skips record);
//              if r1 continues the chain (a '-' site row within skip_thresh+1) the event shifts with the slots
//              and stays at its own position p, else it is dropped;
//   pos(r1) <= site: if r1 is a '-' site row the strand flips, mpos is re-set to r1's site but the slots are
//              kept: the event now sits at pseudo-position pos(r1)+o_r-o_f; else everything is cleared.
// From then on the block behaves as a regular '-' block starting at f0+1 with one extra event, first in its
// slot, at that (pseudo-)position.
__global__ void k0_classify(DevTable T, DevRef R, NbDesc *__restrict__ desc, const int64_t *__restrict__ nb_f0,
                            int entry_read, int k, int skip_thresh, Counters *cnt, unsigned long long pass_no) {
    const int b = (int)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (b >= T.n_nb) return;
    NbDesc d = desc[b];
    classify_block(T, R, d, b, nb_f0[b], d.vf, true, nb_f0, entry_read, k, skip_thresh, cnt, pass_no);
    desc[b] = d;
}

// ---------------------------------------------------------------------------------------------------
// K1: the window scan, as two launches
//
//   k1_scan  streams the position column (and, on a table's first pass, the event-index column, validating every row),
//            finds the units of eight rows that can hold a site row at all (one extract from the strand bitmask per unit)
//            and decides for every site row of those whether it is the LAST row of its window: the next unfiltered row
//            starts another read or lies beyond the site (:179).  Output: a 32-byte payload per closed window + a count
//            per tile.
//   k1_emit  eight lanes per closed window: which of the rows before its last row belong to which slot, reading the event
//            and model columns only for these rows, and builds the flush record (slot means in NumPy pairwise order).
//            Records land in file order (slot = exclusive scan of the tile counts + rank inside the tile).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t MC_I_BIG = 0x1000u;   // internal: a slot holds > 128 events, finished by k1_bigfix
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
    const uint32_t per = (uint32_t)((n_rec + PACK_WGS - 1) / PACK_WGS);
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
    DevRecords O;                 // records, file order
    Counters *cnt;
    int k, skip_thresh, tail_contig;
    int64_t *rare_list;           // [capacity] records k1_emit leaves to k1_rare
    unsigned long long pass_no;   // what Counters.irregular_pass is set to when the pass cannot be finished by the fast path
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

// ---- k1_scan's rare paths, kept out of line (everything they need comes from global memory) ----
struct ScanGlobals {
    const int32_t *pos;
    const uint8_t *flags;
    const int64_t *nb_row_begin;
    const NbDesc *desc;
    int64_t n_rows;
    int n_nb, tail_contig, k, skip_thresh;
};
struct CloseRes { int64_t row; int pos; int ns; };
struct RowRes { int64_t cr; int m, cp, closed; uint32_t pf; };

__device__ __forceinline__ CloseRes far_close_body(const ScanGlobals &G, int nb_abs, int64_t my_end, int64_t r) {
    DevTable T;
    T.n_rows = G.n_rows; T.flags = const_cast<uint8_t *>(G.flags); T.pos = const_cast<int32_t *>(G.pos);
    T.nb_row_begin = const_cast<int64_t *>(G.nb_row_begin); T.n_nb = G.n_nb;
    CloseRes c;
    bool ns;
    c.row = find_close(T, G.desc, G.tail_contig, nb_abs, my_end, r, c.pos, ns);
    c.ns = ns ? 1 : 0;
    return c;
}
__device__ __noinline__ CloseRes far_close(const ScanGlobals G, int nb_abs, int64_t my_end, int64_t r) { return far_close_body(G, nb_abs, my_end, r); }

// word w of a strand mask of n_words words (0 outside)
__device__ __forceinline__ uint32_t mask_word_global(const uint32_t *__restrict__ gbits, int64_t n_words, int64_t w) {
    return (w < 0 || w >= n_words) ? 0u : gbits[w];
}
__device__ __forceinline__ int site_off_global(const uint32_t *__restrict__ gbits, int contig_len, int k, int p) {
    if (p >= contig_len) return -1;
    const int64_t n_words = (((int64_t)contig_len + 31) >> 5) + 2, w = p >> 5;
    uint64_t bits = (((uint64_t)mask_word_global(gbits, n_words, w + 1) << 32) | mask_word_global(gbits, n_words, w)) >> (p & 31);
    bits &= (1ull << k) - 1ull;
    return bits ? __builtin_ctzll(bits) : -1;
}

// Is `row` (unfiltered, inside its regular name block) the last row of a window?  Everything from global memory.
// (_body: inlined where the caller has many values alive -- they would all have to sit in the callee-saved half of the
// registers across a call: the one-base-motif scan went from 120 to 180 registers with the call in its row loop)
__device__ __forceinline__ RowRes far_row_body(const ScanGlobals &G, const uint32_t *gbits, int contig_len, int nb_abs, int64_t my_end,
                                               int64_t row) {
    RowRes res;
    res.cr = 0; res.m = 0; res.cp = 0; res.closed = 0; res.pf = 0;
    const int p = G.pos[row];
    const int o = site_off_global(gbits, contig_len, G.k, p);
    if (o < 0) return res;
    res.m = p + o;
    const CloseRes c = far_close_body(G, nb_abs, my_end, row);
    res.cr = c.row; res.cp = c.pos;
    res.closed = (c.row >= 0 && (c.ns || c.pos > res.m)) ? 1 : 0;
    if (c.ns) res.pf |= PF_CLOSE_NS;
    if (!c.ns && c.pos <= res.m + G.skip_thresh + 1 && site_off_global(gbits, contig_len, G.k, c.pos) > 0) res.pf |= PF_MULTI;
    return res;
}
__device__ __noinline__ RowRes far_row(const ScanGlobals G, const uint32_t *gbits, int contig_len, int nb_abs, int64_t my_end, int64_t row) {
    return far_row_body(G, gbits, contig_len, nb_abs, my_end, row);
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

struct __attribute__((aligned(16))) CandUnit {    // eight rows that may hold a site row, with the two rows behind them
    int32_t pos[10];
    uint8_t fl[10];     // their flag bytes
    uint16_t i0;        // first row of the unit (tile-relative)
    uint32_t mw[3];     // two words of the block's strand mask, and the word they start at (-1: none)
};
static_assert(sizeof(CandUnit) == 64, "CandUnit layout");

// bits [sh, sh+32) of the 64-bit value hi:lo (sh < 32): one v_alignbit
__device__ __forceinline__ uint32_t bits_from(uint32_t lo, uint32_t hi, int sh) {
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
}

// payload slot of the tile's window number `rank`: the first PT in the tile's own strip, the rest in chunks of 64
struct TileSlots {
    const K1Args &A;
    int64_t tile;
    long long *s_chunk;           // [NCHUNK] first slot of the tile's chunks (LDS)
    int total;                    // windows closed so far
    int lane;
    // A one-base motif: every tile needs a chunk, and the wave would wait 3 us for the counter's answer when it gets there -- it
    // asks at once (-> ahead, lane 0: the shard counter's value before the chunk; -1: none) and looks at the answer when the
    // first chunk is due: reserve() and put() with that variable.  (A tile that closes fewer than PT windows leaves the chunk
    // unused: the payload array has a chunk to spare for every tile.)
    __device__ __forceinline__ long long take_ahead() const {
        return lane == 0 ? (long long)atomicAdd(&A.cnt->shard[(int)(tile & A.shard_mask) * SHARD_PAD], 1ull << A.chunk_shift) : -1;
    }
    __device__ __forceinline__ long long slot_of(int rank) const {
        const int cs = A.chunk_shift;
        return rank < PT ? tile * PT + rank : s_chunk[(rank - PT) >> cs] + ((rank - PT) & ((1 << cs) - 1));
    }
    __device__ __forceinline__ void reserve(int new_total, long long *ahead = nullptr) {   // chunks for ranks < new_total (wave-uniform call)
        const int cs = A.chunk_shift, cm = (1 << cs) - 1;
        const int c0 = total <= PT ? 0 : (total - PT + cm) >> cs, c1 = new_total <= PT ? 0 : (new_total - PT + cm) >> cs;
        if (c1 > c0) {
            if (lane == 0) {
                const int sh = (int)(tile & A.shard_mask);
                const long long per = (A.payload_cap - A.T.n_tiles * PT) >> A.shard_shift;      // (a shift: a 64-bit division is a hundred instructions)
                int cf = c0;
                if (ahead && *ahead >= 0) {         // the chunk fetched ahead is the first of these
                    const long long base = *ahead + (1ll << cs) > per ? -1 : A.T.n_tiles * PT + sh * per + *ahead;
                    if (base < 0) atomicOr(&A.cnt->overflow, 1u);
                    s_chunk[cf] = base;
                    A.tile_chunk[tile * NCHUNK + cf] = base;
                    *ahead = -1;
                    ++cf;
                }
                if (cf < c1) {
                    const int n = c1 - cf;
                    const long long off = (long long)atomicAdd(&A.cnt->shard[sh * SHARD_PAD], (unsigned long long)n << cs);
                    long long base = A.T.n_tiles * PT + sh * per + off;
                    if (off + ((long long)n << cs) > per) { atomicOr(&A.cnt->overflow, 1u); base = -1; }
                    for (int c = cf; c < c1; ++c) {
                        s_chunk[c] = base < 0 ? -1 : base + ((long long)(c - cf) << cs);
                        A.tile_chunk[tile * NCHUNK + c] = s_chunk[c];
                    }
                }
            }
            // lane 0's s_chunk entries, before any lane reads them: LDS operations of one wave execute in order, so this only
            // has to keep the compiler from moving the reads (a workgroup-scope fence would also wait for every payload store
            // in flight -- 2 us, four times per tile in dense mode)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    // the lanes with `closed` write their payloads, in lane order
    __device__ __forceinline__ void put(bool closed, const Payload &P, long long *ahead = nullptr) {
        const unsigned long long bal = __ballot(closed);
        if (!bal) return;
        const int n_new = __popcll(bal);
        reserve(total + n_new, ahead);
        if (closed) {
            const long long slot = slot_of(total + __popcll(bal & ((1ull << lane) - 1ull)));
            if (slot >= 0) A.payload[slot] = P;
        }
        total += n_new;
    }
};

// The name blocks of a chunk beyond the two the register path keeps track of (reads of a few hundred rows or less): every row of
// a regular block is examined from global memory, 64 rows at a time.  Exact, slow, rare.  Blocks nb0 .. nb0 + nnb - 1, rows [c0, c1).
__device__ __forceinline__ void scan_blocks_slowly(const K1Args &A, int nb0, int nnb, int64_t c0, int64_t c1, TileSlots &S, int lane) {
    const DevTable &T = A.T;
    ScanGlobals G;
    G.pos = T.pos; G.flags = T.flags; G.nb_row_begin = T.nb_row_begin; G.desc = A.desc; G.n_rows = T.n_rows;
    G.n_nb = T.n_nb; G.tail_contig = A.tail_contig; G.k = A.k; G.skip_thresh = A.skip_thresh;
    for (int bi = 0; bi < nnb; ++bi) {
        const int nb_abs = nb0 + bi;
        const NbDesc d = A.desc[nb_abs];
        if (d.mode != MODE_REGULAR) continue;
        const uint32_t *gbits = (d.rev ? A.R.mr : A.R.mf) + d.mask_off;
        if (d.extra_row() >= c0 && d.extra_row() < c1) {
            const CloseRes xc = far_close(G, nb_abs, d.row_end, d.extra_row());
            Payload P;
            P.r = d.extra_row(); P.close_row = xc.row; P.m = d.extra_mpos; P.close_pos = xc.pos;
            P.flags = PF_EXTRA | (xc.ns ? PF_CLOSE_NS : 0u);
            P.nb = nb_abs;
            S.put(lane == 0 && xc.row >= 0, P);
        }
        const int64_t lo = max(max(d.row_begin, d.first()), c0), hi = min(d.row_end, c1);
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t row = base + lane;
            RowRes fr;
            fr.closed = 0; fr.cr = 0; fr.m = 0; fr.cp = 0; fr.pf = 0;
            if (row < hi && !(T.flags[row] & MC_F_MODEL_N)) fr = far_row(G, gbits, d.contig_len, nb_abs, d.row_end, row);
            Payload P;
            P.r = row; P.close_row = fr.cr; P.m = fr.m; P.close_pos = fr.cp;
            P.flags = fr.pf | (d.stray_q != NO_STRAY ? PF_STRAY : 0u) | (d.rev ? PF_REV : 0u);
            P.nb = nb_abs;
            S.put(fr.closed != 0, P);
        }
    }
}

// ---- validation: what the rows of a name block look like when each is compared with the row before it (first pass over a
// table; what makes a block "regular", see classify_block) ----
// A name block's flags (V_*) that the classification of the pass did not rest on have come to light in a chunk: they go into the
// table's flags, and if the block was taken for regular the pass cannot be finished by the fast path (mc_wait_records repeats
// it on the table's flags, which are complete by then).  Called by one lane.
__device__ __forceinline__ void note_validation(const K1Args &A, int nb_abs, uint32_t seen) {
    const NbDesc *dp = A.desc + nb_abs;
    if (!(seen & ~dp->vf)) return;
    atomicOr(&A.T.nb_vflags[nb_abs], seen);
    if (dp->mode == MODE_REGULAR) {
        *reinterpret_cast<volatile unsigned int *>(&A.cnt->violation) = 1u;
        *reinterpret_cast<volatile unsigned long long *>(&A.cnt->irregular_pass) = A.pass_no;
    }
}

__device__ __forceinline__ uint32_t row_vflags(int p, int x, int prev_p, int prev_x, bool has_pred) {
    uint32_t f = p == 0 ? V_POS0 : 0u;
    if (has_pred) {
        if (p < prev_p) f |= V_POS_DEC;
        f |= x > prev_x ? V_IDX_INC : (x < prev_x ? V_IDX_DEC : V_IDX_EQ);
    }
    return f;
}

// ... of the rows of one unit (r0 .. r0 + 7) that lie in [lo, hi); rows from pred_from on have their predecessor in the block
__device__ __noinline__ uint32_t cut_unit_vflags(const int32_t *pos, const int32_t *idx, int64_t r0, int64_t lo, int64_t hi, int64_t pred_from) {
    uint32_t f = 0;
    for (int e = 0; e < 8; ++e) {
        const int64_t r = r0 + e;
        if (r < lo || r >= hi) continue;
        const bool pred = r >= pred_from;
        f |= row_vflags(pos[r], idx[r], pred ? pos[r - 1] : 0, pred ? idx[r - 1] : 0, pred);
    }
    return f;
}

// ... of the rows [r0, r1) of a chunk that lie in its third name block or beyond (nb_from: the block of r0): row by row from
// global memory (exact, slow, rare)
__device__ __forceinline__ void validate_rows_slowly(const K1Args &A, int nb_from, int64_t r0, int64_t r1, int lane) {
    const DevTable &T = A.T;
    for (int64_t base = r0; base < r1; base += 64) {
        const int64_t row = base + lane;
        if (row >= r1) continue;
        int b = nb_from;
        while (b + 1 < T.n_nb && T.nb_row_begin[b + 1] <= row) ++b;
        const bool has_pred = row > T.nb_row_begin[b];
        const uint32_t f = row_vflags(T.pos[row], T.idx[row], has_pred ? T.pos[row - 1] : 0, has_pred ? T.idx[row - 1] : 0, has_pred);
        note_validation(A, b, f);
    }
}

// A name-block descriptor through the scalar cache into SGPRs: b is wave-uniform, and nothing in the kernel that calls this writes
// descriptors (K0 of the pass wrote them).  Spelled out as an instruction: a plain `dp->mode` is a VECTOR load of one byte --
// there is no scalar byte load, and the compiler will not use scalar loads at all for memory that a store of the kernel might
// alias -- and the wait for a vector load (vmcnt counts in order) is a wait for every column load in flight as well.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <typename T_>
__device__ __forceinline__ const T_ *uniform_ptr(const T_ *p) {       // (a pointer that is the same in all lanes, said so to the compiler:
    const uint64_t v = reinterpret_cast<uint64_t>(p);                  // out-of-line callers get theirs through a vector register)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const T_ *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ NbDesc desc_uniform(const NbDesc *desc, int b) {
    const NbDesc *p = uniform_ptr(desc) + __builtin_amdgcn_readfirstlane(b);
    union { u32x16 w; NbDesc d; } u;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u.w) : "s"(p));
    return u.d;
}
__device__ __forceinline__ int64_t desc_row_end_uniform(const NbDesc *desc, int b) {
    const int64_t *p = &(uniform_ptr(desc) + __builtin_amdgcn_readfirstlane(b))->row_end;
    int64_t v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
}

constexpr int SCAN_VALIDATE = 0;    // a table's first pass: positions, event indices and flag bytes streamed (9 B/row), every row validated
constexpr int SCAN_STREAM = 1;      // a validated table, positions and flag bytes streamed (5 B/row): one-base motifs, where every unit is listed
constexpr int SCAN_SUMMARY = 2;     // a validated table that has unit summaries (k_summarize): 1 B/row

#ifndef MC_STREAM_FLAGS
#define MC_STREAM_FLAGS 1           // (variant builds: 0 = the flag bytes are not streamed, the listed units fetch theirs)
#endif
template <int MODE> constexpr bool flags_streamed() { return MODE != SCAN_SUMMARY && MC_STREAM_FLAGS != 0; }

// The columns of one chunk in registers: every lane holds eight consecutive rows (a unit) of each 512-row stripe.
template <int MODE>
struct ChunkCols {
    int4 pa[NQ], pb[NQ];            // positions of rows i0 .. i0+3, i0+4 .. i0+7 of the lane's unit in stripe j (SCAN_SUMMARY: pa.x, pb.w only)
    int4 xa[NQ], xb[NQ];            // ... their event indices (SCAN_VALIDATE)
    uint2 fl[NQ];                   // ... their flag bytes (not SCAN_SUMMARY)
    __device__ __forceinline__ void load(const DevTable &T, int64_t c0, int crows, int lane) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int i0 = (j * 64 + lane) * 8;
            pa[j] = pb[j] = xa[j] = xb[j] = make_int4(0, 0, 0, 0);
            fl[j] = make_uint2(0x02020202u, 0x02020202u);       // (rows past the table: MC_F_MODEL_N, never looked at anyway)
            if (i0 < crows) {                       // (arrays are padded: whole units stay in bounds)
                if (MODE == SCAN_SUMMARY) {
                    const int2 pp = T.unit_pp[(c0 + i0) >> 3];
                    pa[j].x = pp.x;
                    pb[j].w = pp.y;
                } else {
                    pa[j] = *reinterpret_cast<const int4 *>(T.pos + c0 + i0);
                    pb[j] = *reinterpret_cast<const int4 *>(T.pos + c0 + i0 + 4);
                    if (flags_streamed<MODE>()) fl[j] = *reinterpret_cast<const uint2 *>(T.flags + c0 + i0);
                    if (MODE == SCAN_VALIDATE) {
                        xa[j] = *reinterpret_cast<const int4 *>(T.idx + c0 + i0);
                        xb[j] = *reinterpret_cast<const int4 *>(T.idx + c0 + i0 + 4);
                    }
                }
            }
        }
    }
};

// The rows of one regular name block inside a chunk, for a one-base motif (k1_scan<CG_DENSE>): four rows in five are site rows
// there and every unit would be listed -- so no list: the chunk's columns are in LDS, every lane its own units (k1_scan put
// them there: the registers carry the next chunk's columns meanwhile), and every lane examines the eight rows of its unit of
// each stripe with the two rows behind them (the next lane's): first which of them are last rows of windows, then -- the
// lanes' counts added up -- the payloads, in row order.
struct DenseStash { const int4 *pa, *pb, *fw; const int *dec; };      // [NQ * 64]: positions 0..3, 4..7 | flag bytes 0..7, the unit's two mask words | decidable
__device__ __forceinline__ void dense_block_rows(const K1Args &A, TileSlots &S, long long &ahead, DenseStash L, int nb_abs, int64_t c0,
                                                 int64_t c1, int2 tail_p, uint32_t tail_f) {
    const DevTable &T = A.T;
    const int lane = threadIdx.x, k = A.k;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;
    ScanGlobals G;
    G.pos = T.pos; G.flags = T.flags; G.nb_row_begin = T.nb_row_begin; G.desc = A.desc; G.n_rows = T.n_rows;
    G.n_nb = T.n_nb; G.tail_contig = A.tail_contig; G.k = k; G.skip_thresh = A.skip_thresh;
    const NbDesc d = desc_uniform(A.desc, nb_abs);      // (again: the descriptors need not live in SGPRs through the phases)
    if (d.mode != MODE_REGULAR) return;
    const uint32_t *gbits = (d.rev ? A.R.mr : A.R.mf) + d.mask_off;
    if (d.extra_row() >= c0 && d.extra_row() < c1) {     // the '+' window of a palindromic first site row (R5): first of the block
        const CloseRes xc = far_close_body(G, nb_abs, d.row_end, d.extra_row());
        Payload P;
        P.r = d.extra_row(); P.close_row = xc.row; P.m = d.extra_mpos; P.close_pos = xc.pos;
        P.flags = PF_EXTRA | (xc.ns ? PF_CLOSE_NS : 0u);
        P.nb = nb_abs;
        S.put(lane == 0 && xc.row >= 0, P, &ahead);
    }
    const int64_t lb_abs = max(d.row_begin, d.first());
    const int lo = (int)(max(lb_abs, c0) - c0), hi = (int)(min(d.row_end, c1) - c0);
    const int hi_close = (int)(min(d.row_end, c1 + 2) - c0);        // (rows that can close a window: the two behind the chunk too)
    const uint32_t base_flags = (d.stray_q != NO_STRAY ? PF_STRAY : 0u) | (d.rev ? PF_REV : 0u);
#pragma unroll 1
    for (int j = 0; j < NQ; ++j) {
        if (j * 512 + 512 <= lo || j * 512 >= hi) continue;      // (wave-uniform)
        const int i0 = (j * 64 + lane) * 8;
        const int4 qa = L.pa[j * 64 + lane], qb = L.pb[j * 64 + lane], qw = L.fw[j * 64 + lane];
        const uint2 qf = make_uint2((uint32_t)qw.x, (uint32_t)qw.y);
        // the two rows behind the unit: the next lane's first two (lane 63: the next stripe's, or the rows behind the chunk)
        const int un = j * 64 + lane + 1;           // (unit behind this one, < NQ * 64 unless this is the chunk's last)
        const bool last = un >= NQ * 64;
        const int4 na = L.pa[last ? 0 : un], nw = L.fw[last ? 0 : un];
        const int nx = last ? tail_p.x : na.x, ny = last ? tail_p.y : na.y;
        const uint32_t nf = last ? tail_f : (uint32_t)nw.x;
        const int ps[10] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, nx, ny};
        // bit e: row e of the ten has an 'N' model k-mer (MC_F_MODEL_N is bit 1 of the flag byte)
        static_assert(MC_F_MODEL_N == 2, "the bit picked out of the flag bytes below");
        uint32_t nbits = 0;
#pragma unroll
        for (int e = 0; e < 10; ++e) {
            const uint32_t wd = e < 4 ? qf.x : e < 8 ? qf.y : nf;
            nbits |= ((wd >> (8 * (e & 3) + 1)) & 1u) << e;
        }
        // ... is a row of the block that is tested (>= lo) / lies in front of the block's (the chunk's) end: all ten at once
        const uint32_t below_hi = (1u << min(max(hi - i0, 0), 10)) - 1u, from_lo = ~((1u << min(max(lo - i0, 0), 10)) - 1u);
        const uint32_t act = below_hi & from_lo & ~nbits & 0xFFu;
        // the next unfiltered row of the read inside the chunk: the row behind this one (bit e of dc1), or the one behind an
        // 'N' row (dc2)
        const uint32_t free_rows = ((1u << min(max(hi_close - i0, 0), 10)) - 1u) & ~nbits;
        const uint32_t dc1 = free_rows >> 1, dc2 = ~dc1 & (nbits >> 1) & (free_rows >> 2);
        // first 'M' in meth_ref[p:p+k] (:176,:270) from the unit's two mask words W (bit 0 = position w_base); ok = false
        // when they do not hold all k bits (the row is then looked at out of line)
        const bool w_any = L.dec[j * 64 + lane] != 0;
        const int w_base = ps[0] & ~31;
        const uint64_t W = ((uint64_t)(uint32_t)qw.w << 32) | (uint32_t)qw.z;
        auto site_off = [&](int p, bool &ok) -> int {
            const uint32_t q = (uint32_t)(p - w_base);
            const bool beyond = p >= d.contig_len;
            ok = beyond | (w_any & (q <= (uint32_t)(64 - k)));
            const uint32_t bits = beyond ? 0u : (uint32_t)(W >> (q & 63u)) & kmask;
            return __ffs(bits) - 1;
        };
        uint32_t cbits = 0, farbits = 0, multibits = 0, how = 0;    // how: four bits per row, m - pos | (closing row is two behind) << 3
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = ps[e];
            bool ok, ok2;
            const int o = site_off(p, ok);
            const int m = p + o;
            const bool a = (act >> e) & 1u, has_dc = ((dc1 | dc2) >> e) & 1u, two = (dc2 >> e) & 1u;
            const int cp = two ? ps[e + 2] : ps[e + 1];
            const int o2 = site_off(cp, ok2);
            const bool site = a & ok & (o >= 0);
            const bool near = cp <= m + A.skip_thresh + 1;
            const bool shut = site & has_dc & (cp > m);
            // out of line: mask bits beyond the unit's two words; a closing row past the chunk / the block or behind two 'N' rows
            const bool far = (a & !ok) | (site & !has_dc) | (shut & near & !ok2);
            const bool closed = shut & !(near & !ok2);
            farbits |= (far ? 1u : 0u) << e;
            cbits |= (closed ? 1u : 0u) << e;
            multibits |= ((closed & near & (o2 > 0)) ? 1u : 0u) << e;
            how |= (closed ? (uint32_t)(o | (two ? 8 : 0)) : 0u) << (4 * e);
            // (the rows one after the other, their results gathered as they come: the compiler would keep the 32 of them apart until
        // the end of the loop, and their lane masks side by side do not fit the scalar registers)
        asm volatile("" : "+v"(cbits), "+v"(farbits), "+v"(multibits), "+v"(how));
        __builtin_amdgcn_sched_barrier(0);
        }
        if (__ballot(farbits != 0u)) {              // rare
            for (uint32_t fb = farbits; fb; fb &= fb - 1u) {
                const int e = __ffs(fb) - 1;
                const RowRes fr = far_row_body(G, gbits, d.contig_len, nb_abs, d.row_end, c0 + i0 + e);
                if (fr.closed) cbits |= 1u << e;
            }
        }
        const unsigned long long any = __ballot(cbits != 0u);
        if (!any) continue;
        // the lanes' counts (0 .. 8) added up bit by bit: four ballots, no trip through the LDS crossbar
        const int mine = __popc(cbits);
        int before = 0, n_new = 0;
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            const unsigned long long mb = __ballot((mine >> bt) & 1);
            before += __popcll(mb & below) << bt;
            n_new += __popcll(mb) << bt;
        }
        S.reserve(S.total + n_new, &ahead);
        const int rank0 = S.total + before;
        // (a lane's payloads lie side by side unless a chunk ends between them: one look at the chunk table per lane)
        const int cs = A.chunk_shift;
        const long long slot0 = S.slot_of(rank0);
        const int room = rank0 < PT ? PT - rank0 : (1 << cs) - ((rank0 - PT) & ((1 << cs) - 1));    // slots from slot0 to the end of its strip
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!((cbits >> e) & 1u) || ((farbits >> e) & 1u)) continue;
            const uint32_t h = (how >> (4 * e)) & 15u;
            const int dc = 1 + (int)(h >> 3);
            Payload P;
            P.r = c0 + i0 + e; P.close_row = c0 + i0 + e + dc; P.m = ps[e] + (int)(h & 7u);
            P.close_pos = dc == 1 ? ps[e + 1] : ps[e + 2];
            P.flags = base_flags | (((multibits >> e) & 1u) ? PF_MULTI : 0u);
            P.nb = nb_abs;
            const int t = __popc(cbits & ((1u << e) - 1u));
            const long long slot = t < room ? (slot0 < 0 ? -1 : slot0 + t) : S.slot_of(rank0 + t);
            if (slot >= 0) A.payload[slot] = P;
            __builtin_amdgcn_sched_barrier(0);      // (one payload at a time: eight side by side are eighty registers)
        }
        if (__ballot((cbits & farbits) != 0u)) {    // rare: once more, for what the payload holds
            for (uint32_t fb = cbits & farbits; fb; fb &= fb - 1u) {
                const int e = __ffs(fb) - 1;
                const RowRes fr = far_row_body(G, gbits, d.contig_len, nb_abs, d.row_end, c0 + i0 + e);
                Payload P;
                P.r = c0 + i0 + e; P.close_row = fr.cr; P.m = fr.m; P.close_pos = fr.cp;
                P.flags = fr.pf | base_flags;
                P.nb = nb_abs;
                const long long slot = S.slot_of(rank0 + __popc(cbits & ((1u << e) - 1u)));
                if (slot >= 0) A.payload[slot] = P;
            }
        }
        S.total += n_new;
    }
}

// k1_scan: THE SCAN.  One wave per tile of TILE rows, nothing persistent, no barrier.  The wave takes its tile in chunks of
// CHUNK rows and keeps the memory system busy throughout: the columns of the next chunk are requested as soon as the registers
// of the current one are free, and travel while the current chunk's candidates are examined.
//
// The columns of a chunk go from HBM into REGISTERS -- every lane holds eight consecutive rows (a unit) of each 512-row stripe.
// On a table's first pass (SCAN_VALIDATE) these are the positions, the event indices and the flag bytes, 9 B/row, and every row
// is compared with the row before it -- its neighbour in the lane, the previous lane's last row (one shuffle), the last row of
// the chunk before -- which gives the validation flags of the chunk's name blocks: a chunk inside one block (the usual case)
// ends with wave-wide flags that are held against what the block was classified on, and nothing is written unless they say
// more.  The comparisons run while the mask words below are on their way.
//
// 95 % of the units never leave the registers: one 32-bit extract from the strand bitmask (two words per unit, fetched
// straight from L2 -- the masks of a bacterial genome are 0.6 MB per strand) tells whether any of the unit's k-mers holds an
// 'M' at all.  Only the units that pass are written to an LDS list, with the two rows behind them, their flag bytes and their
// mask words (where the filter read unit summaries instead of the columns, SCAN_SUMMARY, one lane per listed unit fetches its
// rows and flag bytes now); then -- the columns' registers are free again, the next chunk's columns are on their way -- one lane
// per row of the listed units decides whether the row is the LAST row of a window: its k-mer holds an 'M' (first one: the
// site m, :176) and the next unfiltered row of the read lies beyond m, or there is none and another read (or the next shard)
// follows (:179).  Every closed window leaves a 32-byte payload (last row, site, closing row); which of the rows before it
// belong to which slot is worked out by k1_emit, eight lanes per window.  Whatever needs more than the list holds (a closing
// row beyond the chunk or behind two 'N' rows, mask words beyond the unit's two) is an out-of-line call that reads global
// memory.  The descriptors of the chunk's first two name blocks sit in SGPRs; a third block (reads of a few hundred rows) is
// examined row by row from global memory.
// CG: capacity of the candidate list (per chunk).  The sparse instance (a GATC-like motif: one unit in 20 is listed) bails out
// to scan_blocks_slowly if a chunk overflows it; the dense instance holds every unit of a chunk.
#ifdef MC_SCAN_WPE                  // (variant builds, tools/variants.sh)
#define MC_SCAN_ATTR __attribute__((amdgpu_waves_per_eu(MC_SCAN_WPE, MC_SCAN_WPE)))
#else
// (the instance that reads unit summaries fits 80 registers -- six waves per SIMD -- give or take one: said to the compiler)
#define MC_SCAN_ATTR __attribute__((amdgpu_waves_per_eu(MODE == SCAN_SUMMARY ? 6 : 1)))
#endif
template <int CG, int MODE>
__global__ __launch_bounds__(64) MC_SCAN_ATTR void k1_scan(K1Args A) {
    __shared__ __attribute__((aligned(16))) CandUnit s_cand[CG];
    __shared__ long long s_chunk[NCHUNK];           // first payload slot of the tile's 64-record chunks
    const DevTable &T = A.T;
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int k = A.k;
    const int64_t t0 = tile * TILE;
    const int nrows = (int)(min(t0 + (int64_t)TILE, T.n_rows) - t0);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;

    ChunkCols<MODE> C;
    C.load(T, t0, min(nrows, CHUNK), lane);
    // (a one-base motif: the two rows behind the chunk -- the columns are padded by a tile -- so that the windows of the chunk's
    // last rows are closed like all others: one in eight chunks ends on a site row, and a row that is looked at out of line costs
    // the wave a chain of five loads)
    int2 tail_p = make_int2(0, 0);
    uint32_t tail_f = 0;
    auto load_tail = [&](int64_t c1) {
        tail_p = make_int2(T.pos[c1], T.pos[c1 + 1]);
        tail_f = (uint32_t)T.flags[c1] | ((uint32_t)T.flags[c1 + 1] << 8);
    };
    if constexpr (CG > 64) load_tail(t0 + min(nrows, CHUNK));
    int before_p = 0, before_x = 0;                 // the row before the chunk (its first row's predecessor, if that is in its block)
    if (MODE == SCAN_VALIDATE && t0 > 0) { before_p = T.pos[t0 - 1]; before_x = T.idx[t0 - 1]; }
    int nb0 = __builtin_amdgcn_readfirstlane(T.tile_nb[tile]);      // first name block that overlaps the chunk
    ScanGlobals G;
    G.pos = T.pos; G.flags = T.flags; G.nb_row_begin = T.nb_row_begin; G.desc = A.desc; G.n_rows = T.n_rows;
    G.n_nb = T.n_nb; G.tail_contig = A.tail_contig; G.k = k; G.skip_thresh = A.skip_thresh;
    TileSlots S{A, tile, s_chunk, 0, lane};
    long long ahead = -1;
    if constexpr (CG > 64) ahead = S.take_ahead();
    int half = 0;                                   // windows closed in the first chunk
    static_assert(CG <= 64 || NCH == 2, "tile_half: two chunks per tile");

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const int crows = min(nrows - ch * CHUNK, CHUNK);
        if (crows <= 0) break;
        const int64_t c0 = t0 + (int64_t)ch * CHUNK, c1 = c0 + crows;
        const bool more = ch + 1 < NCH && nrows > (ch + 1) * CHUNK;
        // ---- the name blocks that overlap the chunk: A = nb0, B = nb0 + 1 (if any), and whether there are more ----
        while (nb0 + 1 < T.n_nb && desc_row_end_uniform(A.desc, nb0) <= c0) ++nb0;
        const NbDesc da = desc_uniform(A.desc, nb0);
        const bool has_b = nb0 + 1 < T.n_nb && da.row_end < c1;
        const NbDesc db = desc_uniform(A.desc, nb0 + (has_b ? 1 : 0));
        const bool has_c = has_b && nb0 + 2 < T.n_nb && db.row_end < c1;     // a third block: rows from db.row_end on take the slow path
        const int nfast = has_b ? 2 : 1;

        // ---- the mask words of the units: a unit that lies wholly inside block A or B (from the block's first tested row on) spans
        // positions [p0, p7]; its rows' k-mers cover mask bits [p0, p7 + k) of that block's strand.  If that is at most 32 bits,
        // the two words from p0 >> 5 decide whether the unit can hold a site row; units cut by a block's ends and spans that do
        // not fit are listed unconditionally ----
        uint32_t mlo[NQ], mhi[NQ];
        bool decidable[NQ];
        auto fetch_mask_words = [&]() {
            const bool rega = da.mode == MODE_REGULAR, regb = has_b && db.mode == MODE_REGULAR;
            const int loa = (int)(max(max(da.row_begin, da.first()), c0) - c0), hia = (int)(min(da.row_end, c1) - c0);
            const int lob = (int)(max(max(db.row_begin, db.first()), c0) - c0), hib = (int)(min(db.row_end, c1) - c0);
            const uint32_t *ga = (da.rev ? A.R.mr : A.R.mf) + da.mask_off, *gb = (db.rev ? A.R.mr : A.R.mf) + db.mask_off;
            const int nwa = ((da.contig_len + 31) >> 5) + 2, nwb = ((db.contig_len + 31) >> 5) + 2;     // (two zero words behind every contig's mask)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int i0 = (j * 64 + lane) * 8;
                const int p0 = C.pa[j].x, p7 = C.pb[j].w;
                const bool ina = rega && i0 >= loa && i0 + 8 <= hia, inb = regb && i0 >= lob && i0 + 8 <= hib;
                const uint32_t span = (uint32_t)p7 - (uint32_t)p0 + (uint32_t)k;
                const int w = p0 >> 5;
                decidable[j] = (ina || inb) && span - 1u < 32u && p0 >= 0 && w + 1 < (ina ? nwa : nwb);
                // (every lane loads, the undecidable ones the mask's first words: no branch, so all stripes' loads are in flight
                // together and are waited for once)
                const uint32_t *g = (inb ? gb : ga) + (decidable[j] ? w : 0);
                mlo[j] = g[0];
                mhi[j] = g[1];
            }
        };
        // (the one-base-motif instance fetches them behind the validation: what is alive across the validation's out-of-line
        // calls has to sit in the callee-saved half of the registers, and six values more there are a wave per SIMD less)
        if (CG <= 64 || MODE != SCAN_VALIDATE) fetch_mask_words();

        // ---- first pass over the table: every row against the row before it (while the mask words are on their way) ----
        // (per-lane COUNTS of what the pairs of rows say -- a comparison and an add-with-carry each, two vector instructions and
        // no scalar state; accumulating the lane masks of the comparisons themselves costs this kernel more scalar registers than
        // it has)
        if (MODE == SCAN_VALIDATE) {
            for (int bi = 0; bi < nfast; ++bi) {
                const int64_t rb = bi ? db.row_begin : da.row_begin, re = bi ? db.row_end : da.row_end;
                const uint32_t vf_known = bi ? db.vf : da.vf;
                const int vlo = (int)(max(rb, c0) - c0), vhi = (int)(min(re, c1) - c0);
                const int pred_from = rb < c0 ? 0 : vlo + 1;      // rows from here on have their predecessor in the block
                int n_pdec = 0, n_inc = 0, n_dec = 0, n_pairs = 0, n_pos0 = 0;
                uint32_t f_cut = 0;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (j * 512 + 512 <= vlo || j * 512 >= vhi) continue;      // (wave-uniform)
                    const int i0 = (j * 64 + lane) * 8;
                    // the row before the unit: the previous lane's last row (lane 0: the previous stripe's, or the row before the chunk)
                    int qp = __shfl_up(C.pb[j].w, 1), qx = __shfl_up(C.xb[j].w, 1);
                    {
                        const int sp = j > 0 ? __shfl(C.pb[(j + NQ - 1) % NQ].w, 63) : before_p;
                        const int sx = j > 0 ? __shfl(C.xb[(j + NQ - 1) % NQ].w, 63) : before_x;
                        if (lane == 0) { qp = sp; qx = sx; }
                    }
                    const int ps[8] = {C.pa[j].x, C.pa[j].y, C.pa[j].z, C.pa[j].w, C.pb[j].x, C.pb[j].y, C.pb[j].z, C.pb[j].w};
                    const int xs[8] = {C.xa[j].x, C.xa[j].y, C.xa[j].z, C.xa[j].w, C.xb[j].x, C.xb[j].y, C.xb[j].z, C.xb[j].w};
                    const bool whole = i0 >= pred_from && i0 + 8 <= vhi;
                    const bool cut = !whole && i0 + 8 > vlo && i0 < vhi;
                    if (whole) {                                // the unit and the row before it inside the block: the usual case
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int pp = e ? ps[e - 1] : qp, px = e ? xs[e - 1] : qx;
                            n_pdec += ps[e] < pp;
                            n_inc += xs[e] > px;
                            n_dec += xs[e] < px;
                        }
                        n_pairs += 8;
                        const uint32_t m01 = min((uint32_t)ps[0], (uint32_t)ps[1]), m23 = min((uint32_t)ps[2], (uint32_t)ps[3]);
                        const uint32_t m45 = min((uint32_t)ps[4], (uint32_t)ps[5]), m67 = min((uint32_t)ps[6], (uint32_t)ps[7]);
                        n_pos0 += min(min(m01, m23), min(m45, m67)) == 0u;
                    }
                    // a unit cut by the block's ends (or the block's very first rows): its rows once more, from memory, out of line
                    // (written here on the registers, the compiler shares the comparisons with the usual case above and keeps
                    // their lane masks alive for every unit: 130 spilled scalar registers)
                    if (__ballot(cut) != 0ull && cut) f_cut |= cut_unit_vflags(T.pos, T.idx, c0 + i0, c0 + vlo, c0 + vhi, c0 + pred_from);
                }
                const uint32_t seen = (__ballot(n_pdec != 0 || (f_cut & V_POS_DEC)) ? V_POS_DEC : 0u) | (__ballot(n_inc != 0 || (f_cut & V_IDX_INC)) ? V_IDX_INC : 0u) |
                                      (__ballot(n_dec != 0 || (f_cut & V_IDX_DEC)) ? V_IDX_DEC : 0u) |
                                      (__ballot(n_inc + n_dec != n_pairs || (f_cut & V_IDX_EQ)) ? V_IDX_EQ : 0u) |
                                      (__ballot(n_pos0 != 0 || (f_cut & V_POS0)) ? V_POS0 : 0u);
                if ((seen & ~vf_known) && lane == 0) note_validation(A, nb0 + bi, seen);
            }
            if (has_c) validate_rows_slowly(A, nb0 + 2, db.row_end, c1, lane);
            // the next chunk's "row before": this chunk's last row
            if (more) {
                before_p = __shfl(C.pb[NQ - 1].w, 63);
                before_x = __shfl(C.xb[NQ - 1].w, 63);
            }
        }

        if (CG > 64 && MODE == SCAN_VALIDATE) fetch_mask_words();
        // ---- a one-base motif: four rows in five are site rows and every unit would be listed -- the rows are examined where they
        // are, in the registers: every lane its eight rows of a stripe (with the two behind them from the next lane), first
        // which of them are last rows of windows, then -- the lanes' counts added up -- the payloads, in row order ----
        if constexpr (CG > 64) {
            // (the chunk out of the registers into LDS, every lane its own units: the loop below is one copy of the code for both
            // stripes, the rows behind a unit are the next lane's without a shuffle, and the registers are free for the next
            // chunk's columns, which set out now)
            __shared__ int4 s_pa[NQ][64], s_pb[NQ][64], s_fw[NQ][64];      // positions 0..3, 4..7 | flag bytes 0..7, the unit's two mask words
            __shared__ int s_dec[NQ][64];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                s_pa[j][lane] = C.pa[j];
                s_pb[j][lane] = C.pb[j];
                s_fw[j][lane] = make_int4((int)C.fl[j].x, (int)C.fl[j].y, (int)mlo[j], (int)mhi[j]);
                s_dec[j][lane] = decidable[j] ? 1 : 0;
            }
            const int2 tail_p_now = tail_p;
            const uint32_t tail_f_now = tail_f;
#if defined(MC_DENSE_PREFETCH) && MC_DENSE_PREFETCH          // (variant build: the next chunk's columns under way while this one's rows are examined -- no gain, 16 registers more)
            if (more) {
                C.load(T, c0 + CHUNK, min(nrows - (ch + 1) * CHUNK, CHUNK), lane);
                load_tail(c0 + CHUNK + min(nrows - (ch + 1) * CHUNK, CHUNK));
            }
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (one wave: its LDS operations execute in order)
            const DenseStash stash{&s_pa[0][0], &s_pb[0][0], &s_fw[0][0], &s_dec[0][0]};
            // (a third name block and beyond -- reads of a few hundred rows -- the same way: their units have no mask words here,
            // every row of theirs is looked at out of line.  No call in this instance of the kernel: whatever is alive across a
            // call has to sit in the callee-saved half of the registers, and the chunk under way alone is twenty of them)
            int nblk = nfast;
            if (has_c) while (nb0 + nblk < T.n_nb && T.nb_row_begin[nb0 + nblk] < c1) ++nblk;
#pragma unroll 1
            for (int bi = 0; bi < nblk; ++bi) dense_block_rows(A, S, ahead, stash, nb0 + bi, c0, c1, tail_p_now, tail_f_now);
#if !defined(MC_DENSE_PREFETCH) || !MC_DENSE_PREFETCH
            if (more) {
                C.load(T, c0 + CHUNK, min(nrows - (ch + 1) * CHUNK, CHUNK), lane);
                load_tail(c0 + CHUNK + min(nrows - (ch + 1) * CHUNK, CHUNK));
            }
#endif
        } else {
        // ---- all lanes, block by block and stripe by stripe: which units of eight rows can hold a site row at all? ----
        int ncand = 0, seg_end_a = 0;
        bool overflow = false;
        for (int bi = 0; bi < nfast; ++bi) {
            const bool reg = (bi ? db.mode : da.mode) == MODE_REGULAR;
            if (reg) {
                const int64_t lb_abs = bi ? max(db.row_begin, db.first()) : max(da.row_begin, da.first());
                const int lo = (int)(max(lb_abs, c0) - c0), hi = (int)(min(bi ? db.row_end : da.row_end, c1) - c0);
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (j * 512 + 512 <= lo || j * 512 >= hi) continue;      // (wave-uniform)
                    const int i0 = (j * 64 + lane) * 8;
                    const int p0 = C.pa[j].x, p7 = C.pb[j].w;
                    const bool touches = i0 + 8 > lo && i0 < hi;             // (a decidable unit touches its own block only)
                    const int span = p7 - p0 + k;
                    const uint32_t bits = bits_from(mlo[j], mhi[j], p0 & 31) & (0xFFFFFFFFu >> ((32 - span) & 31));
                    const bool cand = touches && (!decidable[j] || bits != 0u);
                    const unsigned long long bal = __ballot(cand);
                    if (!bal) continue;
                    if (ncand + __popcll(bal) > CG) { overflow = true; continue; }
                    // the two rows behind the unit: the next lane's first two rows (lane 63: the next stripe's)
                    int nx = 0, ny = 0;
                    uint32_t nf = 0x0202u;
                    if (MODE != SCAN_SUMMARY) {
                        nx = __shfl_down(C.pa[j].x, 1);
                        ny = __shfl_down(C.pa[j].y, 1);
                        nf = __shfl_down(C.fl[j].x, 1);
                        if (j + 1 < NQ) {
                            const int sx = __shfl(C.pa[(j + 1) % NQ].x, 0), sy = __shfl(C.pa[(j + 1) % NQ].y, 0);
                            const uint32_t sf = __shfl(C.fl[(j + 1) % NQ].x, 0);
                            if (lane == 63) { nx = sx; ny = sy; nf = sf; }
                        }
                    }
                    if (cand) {
                        CandUnit *g = s_cand + (ncand + __popcll(bal & below));
                        int4 *gp = reinterpret_cast<int4 *>(g);
                        if (MODE != SCAN_SUMMARY) {
                            gp[0] = C.pa[j];
                            gp[1] = C.pb[j];
                            // pos[8], pos[9] | flag bytes 0..7
                            gp[2] = make_int4(nx, ny, (int)C.fl[j].x, (int)C.fl[j].y);
                        }
                        // flag bytes 8, 9 and the unit's first row | its two mask words and the word they start at (-1: none, every
                        // lookup out of line)
                        gp[3] = make_int4((int)((nf & 0xFFFFu) | ((uint32_t)i0 << 16)), (int)mlo[j], (int)mhi[j], decidable[j] ? (p0 >> 5) : -1);
                    }
                    ncand += __popcll(bal);
                }
            }
            if (bi == 0) seg_end_a = ncand;
        }
        // ---- the columns' registers are free: the next chunk's columns set out ----
        if (more) C.load(T, c0 + CHUNK, min(nrows - (ch + 1) * CHUNK, CHUNK), lane);
        if (overflow) {                                 // (nothing of the chunk has been written yet)
            scan_blocks_slowly(A, nb0, nfast, c0, c1, S, lane);
        } else {
            __syncthreads();                            // (one wave: orders the list's words between the lanes)
            if (!flags_streamed<MODE>()) {
                // the flag bytes of the listed units and of the two rows behind each (the columns are padded beyond the table's
                // last row); SCAN_SUMMARY: their positions as well
                for (int base = 0; base < ncand; base += 64) {
                    const int gi = base + lane;
                    if (gi < ncand) {
                        CandUnit *g = s_cand + gi;
                        const int i0 = (int)(reinterpret_cast<const uint32_t *>(g)[12] >> 16);
                        const uint8_t *fr = T.flags + c0 + i0;
                        const uint2 f8 = *reinterpret_cast<const uint2 *>(fr);
                        const uint32_t nf = *reinterpret_cast<const uint16_t *>(fr + 8);
                        if (MODE == SCAN_SUMMARY) {
                            const int32_t *pr = T.pos + c0 + i0;
                            const int4 a = *reinterpret_cast<const int4 *>(pr), b4 = *reinterpret_cast<const int4 *>(pr + 4);
                            const int2 nx = *reinterpret_cast<const int2 *>(pr + 8);
                            int4 *gp = reinterpret_cast<int4 *>(g);
                            gp[0] = a;
                            gp[1] = b4;
                            reinterpret_cast<int2 *>(g)[4] = nx;
                        }
                        reinterpret_cast<uint2 *>(g)[5] = f8;
                        reinterpret_cast<uint32_t *>(g)[12] = nf | ((uint32_t)i0 << 16);
                    }
                }
                __syncthreads();
            }

            // ---- one lane per row of the listed units, block by block: is this row the last row of a window? ----
            for (int bi = 0; bi < nfast; ++bi) {
                const int nb_abs = nb0 + bi;
                const int first_g = bi ? seg_end_a : 0, seg_end = bi ? ncand : seg_end_a;
                const NbDesc d = desc_uniform(A.desc, nb_abs);      // (again: the descriptors need not live in SGPRs through the phases)
                if (d.mode != MODE_REGULAR) continue;
                const uint32_t *gbits = (d.rev ? A.R.mr : A.R.mf) + d.mask_off;
                // first 'M' in meth_ref[p:p+k] (:176,:270) from the unit's two mask words; ok = false when they do not hold all k
                // bits (the row then takes the out-of-line path)
                auto site_off = [&](const CandUnit *g, int p, bool &ok) -> int {
                    const int wb = (int)g->mw[2], wi = (p >> 5) - wb, sh = p & 31;
                    const uint32_t two = bits_from(g->mw[0], g->mw[1], sh), one = g->mw[1] >> sh;
                    ok = wb >= 0 && (wi == 0 || (wi == 1 && sh + k <= 32));
                    const uint32_t bits = (wi == 0 ? two : one) & kmask;
                    int o = bits ? (int)__builtin_ctz(bits) : -1;
                    if (p >= d.contig_len) { o = -1; ok = true; }
                    return o;
                };
                // -- the '+' window of a palindromic first site row (R5): one record, first of the block --
                if (d.extra_row() >= c0 && d.extra_row() < c1) {
                    const CloseRes xc = far_close(G, nb_abs, d.row_end, d.extra_row());
                    Payload P;
                    P.r = d.extra_row(); P.close_row = xc.row; P.m = d.extra_mpos; P.close_pos = xc.pos;
                    P.flags = PF_EXTRA | (xc.ns ? PF_CLOSE_NS : 0u);
                    P.nb = nb_abs;
                    S.put(lane == 0 && xc.row >= 0, P);
                }
                const int64_t lb_abs = max(d.row_begin, d.first());
                const int lo = (int)(max(lb_abs, c0) - c0), hi = (int)(min(d.row_end, c1) - c0);
                for (int base = first_g * 8; base < seg_end * 8; base += 64) {
                    const int idx = base + lane;
                    const bool have = idx < seg_end * 8;
                    const CandUnit *g = s_cand + (have ? idx >> 3 : first_g);
                    const int e = idx & 7;
                    const int p = g->pos[e], p1 = g->pos[e + 1], p2 = g->pos[e + 2];
                    const uint32_t f = g->fl[e], f1 = g->fl[e + 1], f2 = g->fl[e + 2];
                    const int i = (int)g->i0 + e;
                    bool closed = false, far = false;
                    int m = 0, cp = 0;
                    int64_t cr = 0;
                    uint32_t pf = 0;
                    if (have && i >= lo && i < hi && !(f & MC_F_MODEL_N)) {
                        bool ok;
                        const int o = site_off(g, p, ok);
                        if (!ok) far = true;
                        else if (o >= 0) {
                            m = p + o;
                            // the next unfiltered row of the read inside the chunk: the row behind this one, or the one behind an 'N' row
                            int c = -1;
                            if (i + 1 < hi && !(f1 & MC_F_MODEL_N)) { c = i + 1; cp = p1; }
                            else if (i + 2 < hi && (f1 & MC_F_MODEL_N) && !(f2 & MC_F_MODEL_N)) { c = i + 2; cp = p2; }
                            if (c >= 0) {
                                cr = c0 + c;
                                closed = cp > m;
                                if (closed && cp <= m + A.skip_thresh + 1) {
                                    bool ok2;
                                    const int o2 = site_off(g, cp, ok2);
                                    if (!ok2) far = true;
                                    else if (o2 > 0) pf |= PF_MULTI;
                                }
                            } else far = true;              // past the chunk / the block, or behind two 'N' rows
                        }
                    }
                    if (__ballot(far)) {                       // rare
                        if (far) {
                            const RowRes fr = far_row(G, gbits, d.contig_len, nb_abs, d.row_end, c0 + i);
                            closed = fr.closed; m = fr.m; cp = fr.cp; cr = fr.cr; pf = fr.pf;
                        }
                    }
                    Payload P;
                    P.r = c0 + i; P.close_row = cr; P.m = m; P.close_pos = cp;
                    P.flags = pf | (d.stray_q != NO_STRAY ? PF_STRAY : 0u) | (d.rev ? PF_REV : 0u);
                    P.nb = nb_abs;
                    S.put(closed, P);
                }
            }
        }
        }
        // ---- a third name block and beyond: row by row ----
        if (CG <= 64 && has_c) {
            int nslow = 1;
            while (nb0 + 2 + nslow < T.n_nb && T.nb_row_begin[nb0 + 2 + nslow] < c1) ++nslow;
            scan_blocks_slowly(A, nb0 + 2, nslow, c0, c1, S, lane);
        }
        if (CG <= 64 && more) __syncthreads();          // (the list is rewritten by the next chunk)
        if (CG > 64 && ch == 0) half = S.total;
    }
    if (lane == 0) A.tile_cnt[tile] = S.total;
    if (CG > 64 && lane == 0) A.tile_half[tile] = half;
}

// Tile counts -> first record slot of every tile, two levels: groups of 1024 tiles are scanned here (coalesced),
// the prefix over the group totals is added by the consumers (tile_slot()).
constexpr int GROUP = 1024;

__global__ __launch_bounds__(GROUP) void k1_group_scan(const int32_t *__restrict__ tile_cnt, int64_t n_tiles,
                                                       int32_t *__restrict__ tile_local, int64_t *__restrict__ group_sum) {
    MC_FRONT_OF_THE_QUEUE;
    __shared__ int s_w[GROUP / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = blockIdx.x * (int64_t)GROUP + tid;
    const int c = t < n_tiles ? tile_cnt[t] : 0;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < GROUP / 64; ++w) {
        if (w < wave) off += s_w[w];
        total += s_w[w];
    }
    if (t < n_tiles) tile_local[t] = off + incl - c;
    if (tid == 0) group_sum[blockIdx.x] = total;
}

// first record slot of a tile; wave-uniform call (all 64 lanes), n_groups <= a few hundred
__device__ __forceinline__ int64_t tile_slot(const int32_t *__restrict__ tile_local, const int64_t *__restrict__ group_sum,
                                             int64_t tile, int lane) {
    const int g = (int)(tile / GROUP);
    long long part = 0;
    for (int i = lane; i < g; i += 64) part += group_sum[i];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    return part + tile_local[tile];
}

// Record for the window of site m whose last row is r, in name block nb_abs (descriptor d).
__device__ __forceinline__ void emit_record(const K1Args &A, RowSrc &S, const NbDesc &d, int nb_abs, int64_t r, int m,
                                            int64_t slot) {
    const DevTable &T = A.T;
    const int k = A.k;
    const int64_t L = d.contig_len;
    const uint32_t *bits = (d.rev ? A.R.mr : A.R.mf) + d.mask_off;
    int close_pos;
    bool close_ns;
    const int64_t close_row = find_close(T, A.desc, A.tail_contig, nb_abs, d.row_end, r, close_pos, close_ns);
    uint32_t info = d.rev ? MC_I_REV : 0u;

    // ---- window rows: back to the first row at position >= m-k+1 (never before d.first() / the block start) ----
    // per-slot event counts packed 8 bits each (a slot with > 128 events goes to k1_bigfix)
    const int64_t lb = max(d.row_begin, d.first());
    unsigned long long cnt8 = 0;
    bool big = false;
    int64_t ws = r;
    for (int64_t rr = r; rr >= lb; --rr) {
        if (T.flags[rr] & MC_F_MODEL_N) continue;
        const int p = T.pos[rr];
        if (p < m - k + 1) break;
        if (p > m) continue;        // (cannot happen in a regular block; a block taken for regular on its first rows may not be)
        const int sh = 8 * (m - p);
        if (((cnt8 >> sh) & 0xFFull) >= 128ull) big = true;
        else cnt8 += 1ull << sh;
        ws = rr;
    }
    // the stray event of a palindromic first site row: first in the slot of its pseudo-position
    int stray_slot = -1;
    if (d.stray_q != NO_STRAY) {
        const int sq = m - d.stray_q;
        if (sq >= 0 && sq < k) {
            stray_slot = sq;
            if (((cnt8 >> (8 * sq)) & 0xFFull) >= 128ull) big = true;
            else cnt8 += 1ull << (8 * sq);
        }
    }
    int nskip = 0;
    for (int s = 0; s < k; ++s) nskip += (((cnt8 >> (8 * s)) & 0xFFull) == 0ull);

    if (nskip > A.skip_thresh) {
        info |= MC_I_TOO_MANY;
        for (int s = 0; s < k; ++s) A.O.feats[slot * k + s] = 0.0;
    } else {
        int64_t cur = ws;
        for (int s = k - 1; s >= 0; --s) {             // positions ascend => slots descend
            const int dst = d.rev ? s : k - 1 - s;      // :187-188
            const int n = (int)((cnt8 >> (8 * s)) & 0xFFull);
            double f = 0.0;
            if (n == 0) info |= 1u << dst;
            else if (!big) {
                if (s == stray_slot) { S.stray_pending = true; S.stray_val = (double)d.stray_d / 10000.0; }
                f = (0.0 + leaf_sum(S, cur, n)) / (double)n;
            }
            A.O.feats[slot * k + dst] = f;
        }
        if (big) { info |= MC_I_BIG; atomicAdd(&A.cnt->n_big, 1u); }   // k1_bigfix recomputes the record
        // context[k], the character after the 'M', picks the sub-model (:197)
        if (m - k + 1 < 0 || (int64_t)m + k > L || m < 1 || m + 1 >= L) {
            info |= MC_I_EDGE;                   // the 2k-1 context leaves the contig: Python slicing decides
        } else {
            unsigned char ch;
            const uint8_t *seq = A.R.seq + A.R.seq_off[d.contig];
            if (!d.rev) ch = bit_at(bits, m + 1) ? 'M' : seq[m + 1];
            else ch = bit_at(bits, m - 1) ? 'M' : comp_char(seq[m - 1]);
            info |= ((uint32_t)ch) << MC_I_NEXT_SHIFT;
        }
    }
    // the closing row shifts the window when it continues the chain with kmer[0] != 'M' (:242-248)
    if (!close_ns && close_pos <= m + A.skip_thresh + 1) {
        if (first_m(bits, L, close_pos, k) > 0) info |= MC_I_MULTI;
    }
    A.O.site_pos[slot] = m;
    A.O.site_seg[slot] = T.nb_seg_begin[nb_abs];        // regular blocks have one segment
    A.O.close_row[slot] = close_row;
    A.O.info[slot] = info;
    A.O.wmask[slot] = 0xFF;                  // (which slot means need 64 bits: k_pack looks)
    A.O.prob[slot] = __longlong_as_double(0x7ff8000000000000LL);
}

// The one-event '+' window a reverse read opens on a palindromic first site row (R5): flushed with k-1 empty slots.
__device__ __forceinline__ void emit_extra(const K1Args &A, const NbDesc &d, int nb_abs, int64_t slot) {
    int close_pos;
    bool close_ns;
    const int64_t close_row = find_close(A.T, A.desc, A.tail_contig, nb_abs, d.row_end, d.extra_row(), close_pos, close_ns);
    for (int s = 0; s < A.k; ++s) A.O.feats[slot * A.k + s] = 0.0;
    A.O.site_pos[slot] = d.extra_mpos;
    A.O.site_seg[slot] = A.T.nb_seg_begin[nb_abs];
    A.O.close_row[slot] = close_row;
    A.O.info[slot] = MC_I_TOO_MANY | ((!close_ns && d.extra_multi()) ? MC_I_MULTI : 0u);
    A.O.wmask[slot] = 0xFF;
    A.O.prob[slot] = __longlong_as_double(0x7ff8000000000000LL);
}


// Eight lanes per tile: the tile's payloads (arrival order) are gathered into file order, so that k1_emit reads them
// with unit stride.  The first record slot of the tile = the windows of all earlier groups of 1024 tiles (summed by the
// eight lanes) + the tile's offset inside its group.
// gather == 0 (dense references: k1_emit_runs takes a tile's payloads where the scan left them, a tile at a time): only the
// total, and the checks.
__global__ __launch_bounds__(256) void k1_list(K1Args A, Payload *__restrict__ sorted, int gather) {
    MC_FRONT_OF_THE_QUEUE;
    const DevTable &T = A.T;
    if (A.chunk_cnt && blockIdx.x == 0)              // (the packing's counts: the emit behind this kernel adds them up)
        for (int i = threadIdx.x; i < PACK_WGS; i += blockDim.x) { A.chunk_cnt[PACK_PAD * i] = 0ull; A.chunk_cnt[PACK_PAD * i + 1] = 0ull; }
    constexpr int LG = 8;
    const int64_t tile = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / LG;
    const int l = threadIdx.x & (LG - 1);
    const bool have = tile < T.n_tiles;
    const int64_t tl = have ? tile : T.n_tiles - 1;
    const int g = (int)(tl / GROUP);
    long long part = 0;
    for (int i = l; i < g; i += LG) part += A.group_sum[i];
#pragma unroll
    for (int o = LG / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (!have) return;
    const int c = A.tile_cnt[tile];
    const int64_t first = part + A.tile_local[tile];
    if (tile == T.n_tiles - 1 && l == 0) A.cnt->n_records = (unsigned long long)(first + c);   // the total
    if (c == 0) return;
    if (first + c > A.O.capacity) { if (l == 0) atomicOr(&A.cnt->overflow, 1u); return; }
    if (!gather) {                                  // (a chunk the scan could not get: the pass is repeated with more room)
        for (int ci = l; PT + (ci << A.chunk_shift) < c; ci += LG)
            if (A.tile_chunk[tile * NCHUNK + ci] < 0) atomicOr(&A.cnt->overflow, 1u);
        return;
    }
    for (int j = l; j < c; j += LG) {
        long long slot = tile * PT + j;
        if (j >= PT) {
            const long long cb = A.tile_chunk[tile * NCHUNK + ((j - PT) >> A.chunk_shift)];
            if (cb < 0) { atomicOr(&A.cnt->overflow, 1u); continue; }
            slot = cb + ((j - PT) & ((1 << A.chunk_shift) - 1));
        }
        sorted[first + j] = A.payload[slot];
    }
}

// the j-th payload of a tile, where the scan left it (arrival order inside a tile is file order)
__device__ __forceinline__ Payload tile_payload(const K1Args &A, int64_t tile, int j) {
    long long slot = tile * PT + j;
    if (j >= PT) slot = A.tile_chunk[tile * NCHUNK + ((j - PT) >> A.chunk_shift)] + ((j - PT) & ((1 << A.chunk_shift) - 1));
    return A.payload[slot];
}

// Eight lanes per closed window, lane s = slot s of the window (k <= 8).  First the eight lanes together look at the 64 rows
// before the window's last row (positions and flag bytes: 320 bytes around one place) and work out which row belongs to which
// slot; then every lane fetches the (event, model) pairs of its slot's rows from the pair column (all lanes of a window hit
// the same DRAM page) and adds them in NumPy's pairwise order (n < 8: sequentially from -0.0, oldest row first; 8..: eight
// strided accumulators, then the tail).  The k slot means of a window leave as k consecutive doubles, adjacent windows
// adjacent: the wave's stores are one contiguous run.  Windows longer than 64 rows go to k1_rare.
constexpr int EG = 8;            // lanes per window
static_assert(EG >= MC_MAX_K, "one lane per slot");

// Windows longer than WROWS rows (a handful per 10^8 rows, if any): k1_emit lists them, k1_rare walks them row by row,
// one thread each, after the host has seen the count.
__device__ __noinline__ void bigfix_record(const K1Args &A, int64_t j);

__global__ void k1_rare(K1Args A, const Payload *__restrict__ sorted, const int64_t *__restrict__ rare_list, int64_t n_rare) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_rare) return;
    const int64_t q = rare_list[i];
    const Payload P = sorted[q];
    const NbDesc d = A.desc[P.nb];
    RowSrc S{A.T.pos, A.T.evmu, A.T.flags, false, 0.0};
    emit_record(A, S, d, P.nb, P.r, P.m, q);
    bigfix_record(A, q);            // (a slot of more than 128 events: finished here, not by a pass of k1_bigfix over all records)
}

// (six waves per SIMD: the register allocator fits 80 VGPRs without scratch; the kernel's time is rounds x latency, so resident
// waves count -- four: 76 us for ordering + emit, five: 59, six: 55, seven (72 VGPRs, 20 bytes of scratch): 57.  Six lanes per
// window for k <= 6, ten windows per wave instead of eight: 66 us -- six loads per lane and step instead of four, the lane
// arithmetic of groups that are not a power of two, and 20 bytes of scratch eat more than the fifth fewer waves give)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k1_emit(K1Args A, const Payload *__restrict__ sorted) {
    const DevTable &T = A.T;
    const int lane = threadIdx.x & 63;
    if (A.cnt->overflow) return;       // the record buffers were too small: k1_list left payloads unwritten, the pass is repeated
    const int64_t n_rec = min((int64_t)A.cnt->n_records, A.O.capacity);
    const int s = lane & (EG - 1);
    const int gsh = lane & ~(EG - 1);                        // first lane of my group
    const int k = A.k;
    // grid-stride over groups of 64/EG windows per wave (the record count is only known on the device)
    // (the payload of the wave's next round is fetched while the current one is worked on)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    Payload Pn;
    Pn.flags = PF_EXTRA; Pn.nb = 0; Pn.r = 0; Pn.m = 0; Pn.close_row = 0; Pn.close_pos = 0;
    {
        const int64_t q0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / EG;
        if (q0 < n_rec) Pn = sorted[q0];
    }
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; (t - lane) / EG < n_rec; t += stride) {
    const int64_t q = t / EG;
    const bool live = q < n_rec;
    const Payload P = Pn;
    {
        const int64_t qn = (t + stride) / EG;
        Pn.flags = PF_EXTRA;
        if (qn < n_rec) Pn = sorted[qn];
    }
    const int64_t r = P.r;
    const int m = P.m;
    const bool window = live && !(P.flags & PF_EXTRA);
    if (live && !window && s == 0) {            // the one-event '+' window of a palindromic first site row (R5)
        for (int s2 = 0; s2 < k; ++s2) A.O.feats[q * k + s2] = 0.0;
        A.O.wmask[q] = 0;
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = T.nb_seg_begin[P.nb];
        A.O.close_row[q] = P.close_row;
        A.O.info[q] = MC_I_TOO_MANY | ((!(P.flags & PF_CLOSE_NS) && (A.desc[P.nb].xflags & 1)) ? MC_I_MULTI : 0u);
        A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
    }
    // ---- which of the rows before the window's last row belong to which slot?  Lane l of the group looks at rows r-l,
    // r-l-8, r-l-16, r-l-24: four independent loads of the position and of the flag byte, eight consecutive rows per load
    // instruction and group (the columns have FRONT rows of padding in front: no clamping).  A row is in the window iff it
    // is unfiltered, not before the block's first tested row, and its k-mer offset m - pos is one of 0..k-1; positions are
    // non-decreasing in a regular block, so the first unfiltered row with pos < m-k+1 (or the block's start) ends the window.
    // One window in a hundred is longer than 32 rows: the groups that saw no end look at rows 32..63 in a second step; a
    // window longer than 64 rows goes to the row-by-row kernel (k1_rare).  (Fewer rows looked at = fewer DRAM lines per
    // window: the kernel's time is the number of scattered lines it touches.) ----
    const NbDesc *dp = A.desc + P.nb;
    uint32_t W = 0xFFFFFFFFu;                   // my eight rows' slots, four bits each (15: not in the window)
    bool stop_any = false;
    int back = 0;
    // rows r-l-8e, e = E0 .. E0+3 -> their nibbles of W
    auto look = [&](const int E0) {
        const int32_t *pp = T.pos + (r - s);
        const uint8_t *fp = T.flags + (r - s);
        int pj[4];
        uint32_t fj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pj[e] = pp[-8 * (E0 + e)];
            fj[e] = fp[-8 * (E0 + e)];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool inb = 8 * (E0 + e) <= back;          // not before the block's first tested row
            const bool nj = fj[e] & MC_F_MODEL_N;
            const int code = m - pj[e];
            const bool inw = inb && !nj && code >= 0 && code < k;
            stop_any = stop_any || !inb || (!nj && code >= k);
            W = (W & ~(15u << (4 * (E0 + e)))) | ((inw ? (uint32_t)code : 15u) << (4 * (E0 + e)));
        }
    };
    if (window) {
        back = (int)min(r - max(dp->row_begin, dp->first()), (int64_t)1 << 20) - s;
        look(0);
    }
    bool covered = ((__ballot(stop_any) >> gsh) & 0xFFull) != 0ull;
    if (__ballot(window && !covered)) {             // (one round in twelve)
        if (window && !covered) look(4);
        covered = ((__ballot(stop_any) >> gsh) & 0xFFull) != 0ull;
    }
    const bool fast = window && covered;
    if (window && !covered && s == 0) A.rare_list[atomicAdd(&A.cnt->n_rare, 1u)] = q;
    // ---- lane 0 of the group: what the info word and the segment column need from the descriptor and the reference.  These
    // are three dependent loads (descriptor -> sequence offset -> base / mask word); issued here they are in flight beside
    // the (event, model) loads below instead of behind them ----
    uint32_t ctx_bits = 0u;               // MC_I_EDGE, or context[k] in its place
    int32_t seg_of = 0;
    if (fast && s == 0) {
        const int64_t L = dp->contig_len;
        const bool rev0 = P.flags & PF_REV;
        seg_of = T.nb_seg_begin[P.nb];
        if (m - k + 1 < 0 || (int64_t)m + k > L || m < 1 || m + 1 >= L) {
            ctx_bits = MC_I_EDGE;                // the 2k-1 context leaves the contig: Python slicing decides
        } else {
            // context[k], the character after the 'M', picks the sub-model (:197)
            const uint32_t *bits = (rev0 ? A.R.mr : A.R.mf) + dp->mask_off;
            const uint8_t *seq = A.R.seq + A.R.seq_off[dp->contig];
            unsigned char ch;
            if (!rev0) ch = bit_at(bits, m + 1) ? 'M' : seq[m + 1];
            else ch = bit_at(bits, m - 1) ? 'M' : comp_char(seq[m - 1]);
            ctx_bits = ((uint32_t)ch) << MC_I_NEXT_SHIFT;
        }
    }
    // ---- my slot's rows: bit j of ms <=> row r-j belongs to slot s.  Every lane fetches the eight slot words of its group and
    // picks the nibbles that equal its slot: bit 4e of Z <=> row r-l-8e is mine ----
    uint32_t lo4 = 0u, hi4 = 0u;                // nibble e: rows of lanes 0..3 / 4..7 at distance 8e
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const uint32_t X = (uint32_t)__shfl((int)W, gsh + l) ^ ((uint32_t)s * 0x11111111u);
        const uint32_t Z = ~(X | (X >> 1) | (X >> 2) | (X >> 3)) & 0x11111111u;
        if (l < 4) lo4 |= Z << l; else hi4 |= Z << (l - 4);
    }
    auto spread = [](uint32_t x) -> uint64_t {  // nibble e -> the low half of byte e
        uint64_t y = x;
        y = (y | (y << 16)) & 0x0000FFFF0000FFFFull;
        y = (y | (y << 8)) & 0x00FF00FF00FF00FFull;
        y = (y | (y << 4)) & 0x0F0F0F0F0F0F0F0Full;
        return y;
    };
    uint64_t ms = spread(lo4) | (spread(hi4) << 4);
    if (!fast || s >= k) ms = 0ull;
    // the stray event of a palindromic first site row (R5): first in the slot of its pseudo-position
    bool has_stray = false;
    double stray_val = 0.0;
    if (fast && (P.flags & PF_STRAY)) {
        const NbDesc *ds = dp;
        const int sq = m - ds->stray_q;
        if (s < k && sq == s) { has_stray = true; stray_val = (double)ds->stray_d / 10000.0; }
    }
    const int n = __popcll(ms) + (has_stray ? 1 : 0);
    const uint32_t empties = (uint32_t)(__ballot(fast && s < k && n == 0) >> gsh) & 0xFFu;    // bit s: slot s is empty
    bool kept_rec = false;                  // (lane 0 of a group: its record is a call; wide_bit: which of its slot means are wide)
    unsigned wide_bit = 0u;
    if (fast) {
    const bool too_many = __popc(empties) > A.skip_thresh;
    const bool rev = P.flags & PF_REV;
    if (s < k) {
        double f = 0.0;
        if (!too_many && n > 0) {
            // The slot's values in order (the stray event first, then the rows from the oldest to the newest), eight at a time:
            // all loads of a batch are issued before any value is used, so a slot costs one memory round trip per eight
            // events -- the wave waits for its slowest lane, and with one load per loop iteration a single long slot
            // among the 48 made the whole wave walk it event by event.
            //   n < 8: NumPy adds sequentially, starting from -0.0.
            //   n >= 8: eight strided accumulators over the first n - n%8 values (value i goes to accumulator i%8 = its
            //   place in the batch), combined pairwise, then the tail in order.
            const int n8 = n >= 8 ? n - (n % 8) : 0;
            double acc = -0.0;
            double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, r5 = 0.0, r6 = 0.0, r7 = 0.0;
            uint64_t mm = ms;
            bool stray_next = has_stray;
            for (int base = 0; base < n; base += 8) {
                int2 e[8];
                bool is_stray[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    is_stray[p] = false;
                    e[p] = make_int2(0, 0);
                    if (base + p < n) {
                        if (stray_next) { stray_next = false; is_stray[p] = true; }
                        else {
                            const int j = 63 - __clzll(mm);
                            mm &= ~(1ull << j);
                            e[p] = T.evmu[r - j];
                        }
                    }
                }
                double v[8];
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    v[p] = is_stray[p] ? stray_val : div1e4(e[p].x - e[p].y);
                if (base + 8 <= n8) {
                    r0 += v[0]; r1 += v[1]; r2 += v[2]; r3 += v[3]; r4 += v[4]; r5 += v[5]; r6 += v[6]; r7 += v[7];
                    if (base + 8 == n8) acc = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                } else {
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        if (base + p < n) acc += v[p];
                }
            }
            f = (0.0 + acc) / (double)n;
        }
        const int dst = (too_many || rev) ? s : k - 1 - s;           // :187-188
        A.O.feats[q * k + dst] = f;
        int32_t as_int;
        if (!too_many && !slot_is_narrow(f, &as_int)) wide_bit = 1u << dst;
    }
    wide_bit |= (unsigned)__shfl_xor((int)wide_bit, 1);              // the group's eight lanes: which slot means need 64 bits
    wide_bit |= (unsigned)__shfl_xor((int)wide_bit, 2);
    wide_bit |= (unsigned)__shfl_xor((int)wide_bit, 4);
    if (s == 0) {
        A.O.wmask[q] = (uint8_t)wide_bit;
        uint32_t info = rev ? MC_I_REV : 0u;
        if (too_many) info |= MC_I_TOO_MANY;
        else {
            // bit dst of the info word: feature dst came from an empty slot (:186)
            uint32_t em = empties;
            if (!rev) em = (__brev(empties) >> 24) >> (8 - k);
            info |= em & MC_I_EMPTY_MASK;
            info |= ctx_bits;
        }
        if (P.flags & PF_MULTI) info |= MC_I_MULTI;         // the closing row shifted the window (:242-248)
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = seg_of;
        A.O.close_row[q] = P.close_row;
        A.O.info[q] = info;
        A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
        kept_rec = !too_many;
    }
    }
    count_wave_for_packing(A.chunk_cnt, n_rec, kept_rec, q, wide_bit, k);       // (the packing's counts, k_pack)
    }
}

// ---------------------------------------------------------------------------------------------------
// k1_emit_runs: the emit for references whose marked positions are dense (a one-base motif: a window closes every eight rows).
// There consecutive windows of a read share five of their six positions -- the reference's shift (:242-256) carries the
// slots from site to site -- so the mean of the events at one position is computed ONCE, not once per window that holds the
// position: a workgroup takes a tile of the table in pieces of ET rows (with EH rows of the rows in front), stages positions,
// flag bytes and (event - model) in LDS, cuts the unfiltered rows of every regular name block into RUNS of one position,
// gives every run its mean (NumPy's pairwise order, exactly as k1_emit adds a slot), and every closed window whose last row
// lies in the piece picks up the means of the runs that end at that row and lie inside [m - k + 1, m] -- at most k runs,
// counted back in the run table: no walk over rows, no second look at the event column.  What the piece cannot answer (a
// window that reaches behind the rows in front, a run of more than 128 events, the stray event of a palindromic first site
// row, more name blocks than the table holds) goes to the row-by-row kernel like k1_emit's long windows.
// ---------------------------------------------------------------------------------------------------
#ifndef MC_ET
#define MC_ET 1024
#endif
#ifndef MC_ER_WAVES
#define MC_ER_WAVES 6
#endif
constexpr int ET = MC_ET;           // rows per piece
#ifndef MC_EH
#define MC_EH 128
#endif
constexpr int EH = MC_EH;           // rows in front of the piece that are staged with it
constexpr int ER = ET + EH;
constexpr int E_THREADS = 256;
constexpr int E_MAXB = 16;          // name blocks per staged range
constexpr uint8_t RUN_WIDE = 1, RUN_UNUSABLE = 2;
static_assert(TILE % ET == 0, "whole pieces per tile");

#ifdef MC_ER_TRACE      // (variant build: 100 MHz time stamps of the phases of 1024 workgroups in the middle of the grid)
__device__ unsigned long long g_er_trace[1024 * 8];
#define ER_STAMP(i) do { if (tid == 0 && blockIdx.x >= 40000 && blockIdx.x < 41024) g_er_trace[(blockIdx.x - 40000) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ER_STAMP(i) do { } while (0)
#endif

struct RunBlock {                   // a name block that overlaps the staged rows (staged indices), and what its windows need of it
    int end, lb, id, contig;        // lb: first row that is in a run (-1: before the staged rows; >= the staged rows: none)
    int contig_len, stray_q;
    uint32_t xflags;
    int64_t mask_off;
};

// (the barriers of k1_emit_runs order LDS traffic only: __syncthreads() would also wait for every global load in flight -- the
// rows a wave keeps in registers, what a window needs from the reference -- although nobody shares those)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int E_CHUNKS = ER / 64;                          // the staged rows in chunks of 64: a wave owns 5 or 4 consecutive ones
constexpr int E_CPW = 5;
static_assert(ER % 64 == 0 && E_THREADS == 256 && (E_CHUNKS + 3) / 4 <= E_CPW, "the chunks are split over 4 waves, at most E_CPW each");
static_assert(ER < (1 << 12), "s_rrow keeps RUN_* above the row");
constexpr int E_RF_SHIFT = 12;

__global__ __launch_bounds__(E_THREADS) __attribute__((amdgpu_waves_per_eu(MC_ER_WAVES, MC_ER_WAVES))) void k1_emit_runs(K1Args A, Payload *__restrict__ sorted) {
    __shared__ uint16_t s_rid[ER];              // run at or before the row
    __shared__ int32_t s_dc[ER + 8];            // (event - model) of the rows in runs, run after run
    __shared__ double s_mean[ER];
    __shared__ int32_t s_rpos[ER];
    __shared__ uint16_t s_rrow[ER];             // first row of the run (staged index) | RUN_* << 12
    __shared__ uint16_t s_rc0[ER + 2];          // where the run's rows begin in s_dc; one more: where the last run's end
    __shared__ RunBlock s_blk[E_MAXB];
    __shared__ int s_nblk, s_wheads[E_THREADS / 64], s_wins[E_THREADS / 64];
    const DevTable &T = A.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (A.cnt->overflow) return;       // the record buffers were too small: k1_list left payloads unwritten, the pass is repeated
    constexpr int PIECES = TILE / ET;
    const int64_t tile = blockIdx.x / PIECES;
    static_assert(ET == CHUNK && PIECES == 2, "a piece is a chunk of the scan: its windows are the tile's first tile_half, or the rest");
    const int piece = blockIdx.x % PIECES;
    const int w_lo = piece ? A.tile_half[tile] : 0, w_hi = piece ? A.tile_cnt[tile] : A.tile_half[tile];
    if (w_lo >= w_hi) return;
    const int64_t s0 = tile * TILE + (int64_t)piece * ET, s1 = min(s0 + (int64_t)ET, T.n_rows);
    if (s0 >= T.n_rows) return;
    const int64_t n_rec = min((int64_t)A.cnt->n_records, A.O.capacity);
    const int64_t first_rec = tile_slot(A.tile_local, A.group_sum, tile, lane);
    const int k = A.k;
    const int64_t h0 = max(s0 - (int64_t)EH, (int64_t)0);
    const int nst = (int)(s1 - h0);
    // (the tile's payloads are in file order; this thread's first one sets out now and is long there when the run table stands)
    ER_STAMP(0);
    Payload P0;
    P0.r = -1; P0.m = 0; P0.flags = 0; P0.nb = 0; P0.close_row = 0; P0.close_pos = 0;
    if (w_lo + tid < w_hi && first_rec + w_lo + tid < n_rec) P0 = tile_payload(A, tile, w_lo + tid);
    // ---- the rows: a wave owns consecutive chunks of 64 (lane = row in the chunk) and keeps them in registers; with them the
    // chunk in front of its first one (which row in a run came last before the wave's rows) ----
    const int c_lo = (wave * E_CHUNKS + 3) >> 2, c_hi = ((wave + 1) * E_CHUNKS + 3) >> 2;
    int32_t rp[E_CPW], rd[E_CPW];
    uint32_t rfl[E_CPW];
#pragma unroll
    for (int c = 0; c < E_CPW; ++c) {
        const int i = (c_lo + c) * 64 + lane;
        rp[c] = 0; rd[c] = 0; rfl[c] = MC_F_MODEL_N;
        if (c_lo + c < c_hi && i < nst) {
            const int2 e = T.evmu[h0 + i];
            rp[c] = T.pos[h0 + i];
            rd[c] = e.x - e.y;
            rfl[c] = T.flags[h0 + i];
        }
    }
    int32_t pre_p = 0;
    uint32_t pre_f = MC_F_MODEL_N;
    if (c_lo > 0 && (c_lo - 1) * 64 + lane < nst) { pre_p = T.pos[h0 + (c_lo - 1) * 64 + lane]; pre_f = T.flags[h0 + (c_lo - 1) * 64 + lane]; }
    // ---- the name blocks that overlap the staged rows: the first wave looks at the 64 blocks from the one the tile before
    // began in (the staged rows begin at most EH rows in front of this tile), all at once ----
    if (wave == 0) {
        const int bfrom = T.tile_nb[(h0 >= tile * TILE || tile == 0) ? tile : tile - 1];
        const int b = bfrom + lane;
        bool over = false, ends_early = false;
        RunBlock rb;
        if (b < T.n_nb) {
            const NbDesc *dp = A.desc + b;
            const int64_t rbeg = dp->row_begin, rend = dp->row_end;
            over = rbeg < s1 && rend > h0;
            ends_early = rend < s1 && b + 1 < T.n_nb;          // (the block behind this one begins before the piece ends)
            if (over) {
                rb.end = (int)min(rend - h0, (int64_t)nst);
                // first row that belongs to a run: the block's first tested row (rows in front of it, and blocks that are not
                // regular, are in no run)
                rb.lb = dp->mode == MODE_REGULAR ? (int)max(max(rbeg, dp->first()) - h0, (int64_t)-1) : nst;
                rb.id = b;
                rb.contig = dp->contig;
                rb.contig_len = dp->contig_len;
                rb.stray_q = dp->stray_q;
                rb.xflags = dp->xflags;
                rb.mask_off = dp->mask_off;
            }
        }
        const unsigned long long bal = __ballot(over);
        // (blocks are in row order: the overlapping ones are consecutive lanes; one behind the 64 looked at -- reads of a dozen
        // rows -- makes the table unusable, like more than E_MAXB of them)
        const int n = __popcll(bal), at = __popcll(bal & ((1ull << lane) - 1ull));
        const bool more_behind = (__ballot(ends_early) >> 63) & 1ull;
        if (over && at < E_MAXB) s_blk[at] = rb;
        if (lane == 0) s_nblk = more_behind ? E_MAXB + 1 : n;
    }
    lds_barrier();
    ER_STAMP(1);
    const int nblk = s_nblk;
    const bool usable = nblk <= E_MAXB;
    // ---- what the info word of a window needs from the reference (the character after the 'M', the segment of its block):
    // two trips, the first sets out now for this thread's first window, the second when the runs are numbered ----
    struct WinCtx { int bj, seg, at; bool edge; uint32_t word; int64_t soff; unsigned char base; };
    auto ctx_begin = [&](const Payload &P, WinCtx &X) {
        X.bj = 0; X.seg = 0; X.at = 0; X.edge = true; X.word = 0; X.soff = 0; X.base = 0;
        if (!usable || (P.flags & PF_EXTRA)) return;
        while (X.bj + 1 < nblk && s_blk[X.bj].id != P.nb) ++X.bj;
        const RunBlock &B0 = s_blk[X.bj];
        const int m = P.m;
        const int64_t L = B0.contig_len;
        X.seg = T.nb_seg_begin[P.nb];
        X.edge = m - k + 1 < 0 || (int64_t)m + k > L || m < 1 || m + 1 >= L;
        if (!X.edge) {
            const bool rev = P.flags & PF_REV;
            X.at = rev ? m - 1 : m + 1;
            X.word = ((rev ? A.R.mr : A.R.mf) + B0.mask_off)[X.at >> 5];
            X.soff = A.R.seq_off[B0.contig];
        }
    };
    auto ctx_end = [&](WinCtx &X) { if (!X.edge) X.base = (A.R.seq + X.soff)[X.at]; };
    const bool w0_mine = P0.r >= s0 && P0.r < s1;           // (no payload: r = -1)
    WinCtx X0;
    X0.bj = 0; X0.seg = 0; X0.at = 0; X0.edge = true; X0.word = 0; X0.soff = 0; X0.base = 0;
    if (w0_mine) ctx_begin(P0, X0);
    int n_runs = 0;
    if (usable) {
        const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
        // ---- which row in a run came last before the wave's rows (none: row -1) ----
        int carry_row = -1, carry_pos = 0;
        for (int cc = c_lo - 1; cc >= 0; --cc) {
            const int i = cc * 64 + lane;
            int32_t p = pre_p;
            uint32_t f = pre_f;
            if (cc != c_lo - 1 && i < nst) { p = T.pos[h0 + i]; f = T.flags[h0 + i]; }
            int bj = 0;
            while (bj + 1 < nblk && i >= s_blk[bj].end) ++bj;
            const bool in = i >= max(s_blk[bj].lb, 0) && i < s_blk[bj].end && !(f & MC_F_MODEL_N);
            const unsigned long long m = __ballot(in);
            if (m) {
                const int top = 63 - __clzll(m);
                carry_row = cc * 64 + top;
                carry_pos = __shfl(p, top);
                break;
            }
        }
        // ---- the wave's rows: which are in runs, which begin one (the row before it in its block that is in a run lies at
        // another position, or there is none) ----
        unsigned long long inm[E_CPW], headm[E_CPW];
        uint32_t cutm = 0;              // bit c: the lane's row of chunk c begins a run whose first rows may lie in front of the staged ones
        int nh = 0, ni = 0, bj = 0;
#pragma unroll
        for (int c = 0; c < E_CPW; ++c) {
            inm[c] = 0; headm[c] = 0;
            if (c_lo + c >= c_hi) continue;
            const int base = (c_lo + c) * 64, i = base + lane;
            while (bj + 1 < nblk && i >= s_blk[bj].end) ++bj;
            const int lb = s_blk[bj].lb, lbm = max(lb, 0);
            const bool in = i < nst && i >= lbm && i < s_blk[bj].end && !(rfl[c] & MC_F_MODEL_N);
            const unsigned long long m = __ballot(in), below = m & lt;
            const int pl = 63 - __clzll(below | 1ull);
            int prow = base + pl, ppos = __shfl(rp[c], pl);
            if (!below) { prow = carry_row; ppos = carry_pos; }
            const bool alone = prow < lbm, head = in && (alone || ppos != rp[c]);
            if (head && alone && lb < 0) cutm |= 1u << c;
            const unsigned long long hm = __ballot(head);
            inm[c] = m; headm[c] = hm;
            nh += __popcll(hm); ni += __popcll(m);
            if (m) {
                const int top = 63 - __clzll(m);
                carry_row = base + top;
                carry_pos = __shfl(rp[c], top);
            }
        }
        if (lane == 0) { s_wheads[wave] = nh; s_wins[wave] = ni; }
        lds_barrier();
        ER_STAMP(2);
        int hbase = 0, ibase = 0, n_in = 0;
#pragma unroll
        for (int w = 0; w < E_THREADS / 64; ++w) {
            const int a = s_wheads[w], b2 = s_wins[w];
            if (w < wave) { hbase += a; ibase += b2; }
            n_runs += a; n_in += b2;
        }
        // ---- runs numbered in row order; the rows in runs packed run after run ----
#pragma unroll
        for (int c = 0; c < E_CPW; ++c) {
            if (c_lo + c >= c_hi) continue;
            const int i = (c_lo + c) * 64 + lane;
            const int rid = hbase + __popcll(headm[c] & le) - 1, at = ibase + __popcll(inm[c] & lt);
            if (i < nst) s_rid[i] = (uint16_t)max(rid, 0);
            if ((inm[c] >> lane) & 1ull) s_dc[at] = rd[c];
            if ((headm[c] >> lane) & 1ull) {
                s_rrow[rid] = (uint16_t)(i | (((cutm >> c) & 1u) ? (RUN_UNUSABLE << E_RF_SHIFT) : 0));
                s_rpos[rid] = rp[c];
                s_rc0[rid] = (uint16_t)at;
            }
            hbase += __popcll(headm[c]); ibase += __popcll(inm[c]);
        }
        if (tid == 0) s_rc0[n_runs] = (uint16_t)n_in;
        lds_barrier();
        ER_STAMP(3);
        if (w0_mine) ctx_end(X0);
        // ---- the mean of every run: its rows in file order, NumPy's pairwise order (np.mean, :186; values fl(d / 1e4), :286) ----
        for (int R = tid; R < n_runs; R += E_THREADS) {
            double mean = 0.0;
            uint32_t rf = 0;
            const int c0 = s_rc0[R], n = (int)s_rc0[R + 1] - c0;
            const int d0 = s_dc[c0], d1 = s_dc[c0 + 1], d2 = s_dc[c0 + 2], d3 = s_dc[c0 + 3];     // (there is room behind the last row)
            if (n > 128) { rf = RUN_UNUSABLE; mean = 0.0; }          // NumPy's pairwise recursion proper: the row-by-row kernel
            else if (n >= 8) {
                const int n8 = n - (n % 8);
                double r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = 0.0;
                for (int j = 0; j < n8; j += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) r[u] += div1e4(s_dc[c0 + j + u]);
                }
                double acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (int j = n8; j < n; ++j) acc += div1e4(s_dc[c0 + j]);
                mean = (0.0 + acc) / (double)n;
            } else {
                // (every second run is one event: fl(d / 1e4) over 1 -- no division, and narrow by construction)
                double acc = -0.0 + div1e4(d0);
                if (n > 1) acc += div1e4(d1);
                if (n > 2) acc += div1e4(d2);
                if (n > 3) acc += div1e4(d3);
                for (int j = 4; j < n; ++j) acc += div1e4(s_dc[c0 + j]);
                mean = 0.0 + acc;
                if (n > 1) mean = mean / (double)n;
            }
            if (n > 1 && n <= 128) {
                int32_t as_int;
                if (!slot_is_narrow(mean, &as_int)) rf |= RUN_WIDE;
            }
            s_mean[R] = mean;
            if (rf) s_rrow[R] |= (uint16_t)(rf << E_RF_SHIFT);
        }
    }
    lds_barrier();
    ER_STAMP(4);
    // ---- the windows whose last row lies in the piece (the tile's payloads are in file order: a contiguous stretch) ----
    for (int w = w_lo + tid; w < w_hi; w += E_THREADS) {
        const int64_t q = first_rec + w;
        if (q >= n_rec) break;
        const Payload P = w == w_lo + tid ? P0 : tile_payload(A, tile, w);
        if (P.r < s0 || P.r >= s1) continue;
        const int m = P.m;
        const bool rev = P.flags & PF_REV;
        WinCtx X = X0;
        if (w != w_lo + tid) { ctx_begin(P, X); ctx_end(X); }      // (a piece with more windows than the workgroup has threads)
        const int bj = X.bj;
        if (P.flags & PF_EXTRA) {                   // the one-event '+' window of a palindromic first site row (R5)
            for (int s2 = 0; s2 < k; ++s2) A.O.feats[q * k + s2] = 0.0;
            A.O.wmask[q] = 0;
            A.O.site_pos[q] = m;
            A.O.site_seg[q] = T.nb_seg_begin[P.nb];
            A.O.close_row[q] = P.close_row;
            A.O.info[q] = MC_I_TOO_MANY | ((!(P.flags & PF_CLOSE_NS) && (A.desc[P.nb].xflags & 1)) ? MC_I_MULTI : 0u);
            A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
            continue;
        }
        bool rare = !usable;
        uint32_t have = 0, wide = 0;
        if (!rare) {
            const RunBlock &B = s_blk[bj];
            if ((P.flags & PF_STRAY) && m - B.stray_q >= 0 && m - B.stray_q < k) rare = true;     // (the stray event is first in its slot)
            const int lb = B.lb;                           // (< 0: the block's tested rows begin before the staged rows)
            const int R = s_rid[(int)(P.r - h0)];
            for (int t = 0; t < k && !rare; ++t) {
                const int Rt = R - t;
                if (Rt < 0) { if (lb < 0) rare = true; break; }         // (the window reaches behind the rows in front)
                const int rr = s_rrow[Rt];
                if ((rr & ((1 << E_RF_SHIFT) - 1)) < max(lb, 0)) break;           // a run of the block before
                const int qpos = s_rpos[Rt];
                if (qpos < m - k + 1) break;
                const int rf = rr >> E_RF_SHIFT;
                if (rf & RUN_UNUSABLE) { rare = true; break; }
                const int slot = m - qpos;
                if (slot < 0) continue;                    // (the run of the closing row itself, behind the site)
                A.O.feats[q * k + (rev ? slot : k - 1 - slot)] = s_mean[Rt];         // :187-188 (a window that turns out rare is written again)
                have |= 1u << slot;
                if (rf & RUN_WIDE) wide |= 1u << slot;
            }
        }
        if (rare) {                                 // (the row-by-row kernel looks its windows up in the ordered list)
            sorted[q] = P;
            A.rare_list[atomicAdd(&A.cnt->n_rare, 1u)] = q;
            continue;
        }
        const uint32_t kbits = (1u << k) - 1u, empties = ~have & kbits;
        const bool too_many = __popc(empties) > A.skip_thresh;
        uint32_t info = rev ? MC_I_REV : 0u, wmask = 0;
        for (uint32_t z = too_many ? kbits : empties; z; z &= z - 1u) {
            const int s = __ffs(z) - 1;
            A.O.feats[q * k + (rev ? s : k - 1 - s)] = 0.0;
        }
        if (too_many) info |= MC_I_TOO_MANY;
        else {
            // (slot s is feature s on the reverse strand, k - 1 - s on the forward one)
            wmask = rev ? wide : __brev(wide) >> (32 - k);
            info |= rev ? empties : __brev(empties) >> (32 - k);   // feature dst came from an empty slot (:186)
            // context[k], the character after the 'M', picks the sub-model (:197)
            if (X.edge) info |= MC_I_EDGE;                          // the 2k-1 context leaves the contig: Python slicing decides
            else {
                const unsigned char ch = ((X.word >> (X.at & 31)) & 1u) ? 'M' : (rev ? comp_char(X.base) : X.base);
                info |= ((uint32_t)ch) << MC_I_NEXT_SHIFT;
            }
        }
        if (P.flags & PF_MULTI) info |= MC_I_MULTI;         // the closing row shifted the window (:242-248)
        A.O.wmask[q] = (uint8_t)wmask;
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = X.seg;
        A.O.close_row[q] = P.close_row;
        A.O.info[q] = info;
        A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
    }
    ER_STAMP(5);
}

// Records with a slot of more than 128 events (NumPy's pairwise recursion proper): recomputed here, one
// thread per such record, so that k1_emit carries neither the stack nor the registers for it.
__device__ double big_pairwise(RowSrc &S, int64_t &cur, int64_t n) {
    // emulate  f(n) = n <= 128 ? leaf(n) : f(n2) + f(n - n2),  n2 = n/2 rounded down to a multiple of 8
    int64_t fsize[40];
    int fstage[40];      // 0 = not started, 1 = left half pending, 2 = right half pending
    double fleft[40];
    int fp = 1;
    fsize[0] = n;
    fstage[0] = 0;
    double ret = 0.0;
    while (fp > 0) {
        const int top = fp - 1;
        int64_t n2 = fsize[top] / 2;
        n2 -= n2 % 8;
        if (fstage[top] == 0) {
            if (fsize[top] <= 128) {
                ret = leaf_sum(S, cur, (int)fsize[top]);
                --fp;
            } else {
                fstage[top] = 1;
                fsize[fp] = n2; fstage[fp] = 0; ++fp;
            }
        } else if (fstage[top] == 1) {
            fleft[top] = ret;
            fstage[top] = 2;
            fsize[fp] = fsize[top] - n2; fstage[fp] = 0; ++fp;
        } else {
            ret = fleft[top] + ret;
            --fp;
        }
    }
    return ret;
}

__device__ __noinline__ void bigfix_record(const K1Args &A, int64_t j) {
    const DevRecords &O = A.O;
    const uint32_t info = O.info[j];
    if (!(info & MC_I_BIG)) return;
    const DevTable &T = A.T;
    const int k = A.k;
    const int m = O.site_pos[j];
    const bool rev = info & MC_I_REV;
    // name block of the record = the one its (single) segment starts
    const int seg = O.site_seg[j];
    int lo = 0, hi = T.n_nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (T.nb_seg_begin[mid] <= seg) lo = mid; else hi = mid - 1;
    }
    const NbDesc d = A.desc[lo];
    RowSrc S{T.pos, T.evmu, T.flags, false, 0.0};
    // last row of the window: the last unfiltered row of the block before the closing row
    int64_t r = min(O.close_row[j], d.row_end) - 1;
    const int64_t lb = max(d.row_begin, d.first());
    while (r >= lb && (T.flags[r] & MC_F_MODEL_N)) --r;
    int64_t cnt[MC_MAX_K] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t ws = r;
    for (int64_t rr = r; rr >= lb; --rr) {
        if (T.flags[rr] & MC_F_MODEL_N) continue;
        const int p = T.pos[rr];
        if (p < m - k + 1) break;
        if (p > m) continue;        // (see emit_record)
        cnt[m - p] += 1;
        ws = rr;
    }
    int stray_slot = -1;
    if (d.stray_q != NO_STRAY && m - d.stray_q >= 0 && m - d.stray_q < k) {
        stray_slot = m - d.stray_q;
        cnt[stray_slot] += 1;
    }
    int64_t cur = ws;
    for (int s = k - 1; s >= 0; --s) {
        const int dst = rev ? s : k - 1 - s;
        double f = 0.0;
        if (s == stray_slot) { S.stray_pending = true; S.stray_val = (double)d.stray_d / 10000.0; }
        if (cnt[s] > 0) f = (0.0 + big_pairwise(S, cur, cnt[s])) / (double)cnt[s];
        O.feats[j * k + dst] = f;
    }
    O.info[j] = info & ~MC_I_BIG;
    O.wmask[j] = 0xFF;
}

__global__ void k1_bigfix(K1Args A, int64_t n) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j < n) bigfix_record(A, j);
}

// Pipelined passes: the windows k1_emit left to the row-by-row walk (their number is on the device only), each finished
// by one thread, including the full pairwise recursion if a slot turns out to hold more than 128 events.
__global__ void k1_rare_dev(K1Args A, const Payload *__restrict__ sorted, const int64_t *__restrict__ rare_list) {
    if (A.cnt->overflow) return;
    const int64_t n_rare = (int64_t)A.cnt->n_rare;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rare; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = rare_list[i];
        const Payload P = sorted[q];
        const NbDesc d = A.desc[P.nb];
        RowSrc S{A.T.pos, A.T.evmu, A.T.flags, false, 0.0};
        emit_record(A, S, d, P.nb, P.r, P.m, q);
        bigfix_record(A, q);
        if (A.chunk_cnt && !(A.O.info[q] & MC_I_TOO_MANY)) {       // (the packing's counts: k1_emit left this record out)
            int n_wide = 0;
            for (int f = 0; f < A.k; ++f) {
                int32_t d32;
                n_wide += slot_is_narrow(A.O.feats[q * A.k + f], &d32) ? 0 : 1;
            }
            count_for_packing(A.chunk_cnt, q, min((int64_t)A.cnt->n_records, A.O.capacity), true, n_wide);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The literal path: name blocks the window rule does not cover (same read name in several blocks, positions going
// backwards, strand changes inside a read, a site at contig position 0, reads spanning contigs).  Runs of such blocks
// are executed row by row exactly as the reference's loop does (extract_contexts.py:147-291), one GPU thread per run;
// they are rare, so this path is written for exactness, not speed.  Its records are merged with the fast path's by
// closing row (= flush order).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t XR_POS0 = 4;      // NbDesc.xflags: irregular because of a site at contig position 0 (falsy mpos)

// After k0_classify: widen the irregular set so that every run starts and ends in a state the fast path knows.
__global__ void k0_extend(DevTable T, NbDesc *__restrict__ desc, const int64_t *__restrict__ nb_f0, int entry_read,
                          Counters *cnt, unsigned long long pass_no) {
    const int b = (int)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (b >= T.n_nb || cnt->irregular_pass != pass_no) return;
    const NbDesc d = desc[b];
    if (d.mode != MODE_IRREGULAR) return;
    // (a) a block that sees name == last_read continues the state of the block that set last_read: take everything
    //     from that block on (the blocks in between have no site row but may hold rows that flush and reset)
    if (T.nb_repeat[b] || entry_read >= 0) {
        int j = b - 1;
        while (j >= 0 && nb_f0[j] < 0) --j;
        const int last_read = j >= 0 ? T.nb_read[j] : entry_read;
        if (last_read == d.read && j >= 0)
            for (int i = j; i < b; ++i) desc[i].mode = MODE_IRREGULAR;
    }
    // (b) a falsy mpos (site at position 0) can leave events in the slots while no window is open: they survive until
    //     the next window is flushed, i.e. through the next block that has a site row
    if (d.xflags & XR_POS0) {
        for (int i = b + 1; i < T.n_nb; ++i) {
            const bool site_block = nb_f0[i] >= 0 && !desc[i].filtered;
            desc[i].mode = MODE_IRREGULAR;
            if (site_block) break;
        }
    }
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
__device__ double pairwise_arr(const double *a, int64_t n) {
    int64_t fbeg[40], fsize[40];
    int fstage[40];
    double fleft[40];
    int fp = 1;
    fbeg[0] = 0; fsize[0] = n; fstage[0] = 0;
    double ret = 0.0;
    while (fp > 0) {
        const int top = fp - 1;
        int64_t n2 = fsize[top] / 2;
        n2 -= n2 % 8;
        if (fstage[top] == 0) {
            const int64_t m = fsize[top];
            const double *p = a + fbeg[top];
            if (m < 8) {
                double res = -0.0;
                for (int64_t i = 0; i < m; ++i) res += p[i];
                ret = res;
                --fp;
            } else if (m <= 128) {
                double r[8];
                int64_t i;
                for (i = 0; i < 8; ++i) r[i] = p[i];
                for (i = 8; i < m - (m % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r[j] += p[i + j];
                double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < m; ++i) res += p[i];
                ret = res;
                --fp;
            } else {
                fstage[top] = 1;
                fbeg[fp] = fbeg[top]; fsize[fp] = n2; fstage[fp] = 0; ++fp;
            }
        } else if (fstage[top] == 1) {
            fleft[top] = ret;
            fstage[top] = 2;
            fbeg[fp] = fbeg[top] + n2; fsize[fp] = fsize[top] - n2; fstage[fp] = 0; ++fp;
        } else {
            ret = fleft[top] + ret;
            --fp;
        }
    }
    return ret;
}

__global__ void k_literal(LitArgs A) {
    const DevTable &T = A.T;
    const int b0 = (int)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (b0 >= T.n_nb) return;
    if (A.desc[b0].mode != MODE_IRREGULAR || (b0 > 0 && A.desc[b0 - 1].mode == MODE_IRREGULAR)) {
        if (!A.write) { A.run_cnt[b0] = 0; A.run_rows[b0] = 0; }
        return;
    }
    const int k = A.k;
    // ---- machine state (:113-119) ----
    int last_read;
    {
        int j = b0 - 1;
        while (j >= 0 && A.nb_f0[j] < 0) --j;
        last_read = j >= 0 ? T.nb_read[j] : A.entry_read;
    }
    int64_t first_idx = A.entry_first_idx;
    bool has_mpos = false;
    int64_t mpos = 0;
    int last_rev = 0, last_seg = -1;
    int64_t nslot[MC_MAX_K] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sid[MC_MAX_K] = {0, 1, 2, 3, 4, 5, 6, 7};        // logical slot -> physical array
    int64_t run_rows = 0, out = 0, cap = 0;
    double *slots = nullptr;
    if (A.write) {
        const int g = b0 / GROUP;
        long long ro = A.rows_local[b0], co = A.cnt_local[b0];
        for (int i = 0; i < g; ++i) { ro += A.rows_group[i]; co += A.cnt_group[i]; }
        cap = A.run_rows[b0];
        slots = A.scratch + (size_t)ro * MC_MAX_K;
        out = co;
    }
    auto truthy = [&]() { return has_mpos && mpos != 0; };
    auto flush = [&](int64_t close_row, bool multi) {                                   // :179-239
        if (A.write) {
            const int64_t j = out;
            int nskip = 0;
            for (int i = 0; i < k; ++i) nskip += (nslot[i] == 0);
            uint32_t info = last_rev ? MC_I_REV : 0u;
            const int contig = T.seg_contig[last_seg];
            const int64_t L = A.R.contig_len[contig];
            for (int i = 0; i < k; ++i) A.L.feats[j * k + i] = 0.0;
            if (nskip <= A.skip_thresh) {
                for (int i = 0; i < k; ++i) {
                    const int dst = last_rev ? i : k - 1 - i;
                    if (nslot[i] == 0) info |= 1u << dst;
                    else A.L.feats[j * k + dst] = (0.0 + pairwise_arr(slots + (size_t)sid[i] * cap, nslot[i])) / (double)nslot[i];
                }
                if (mpos - k + 1 < 0 || mpos + k > L || mpos < 1 || mpos + 1 >= L) {
                    info |= MC_I_EDGE;
                } else {
                    const uint8_t *seq = A.R.seq + A.R.seq_off[contig];
                    unsigned char ch;
                    if (!last_rev) ch = bit_at(A.R.mf + A.R.word_off[contig], mpos + 1) ? 'M' : seq[mpos + 1];
                    else ch = bit_at(A.R.mr + A.R.word_off[contig], mpos - 1) ? 'M' : comp_char(seq[mpos - 1]);
                    info |= ((uint32_t)ch) << MC_I_NEXT_SHIFT;
                }
            } else {
                info |= MC_I_TOO_MANY;
            }
            if (multi) info |= MC_I_MULTI;
            A.L.site_pos[j] = (int32_t)mpos;
            A.L.site_seg[j] = last_seg;
            A.L.close_row[j] = close_row;
            A.L.info[j] = info;
            A.L.wmask[j] = 0xFF;
            A.L.prob[j] = __longlong_as_double(0x7ff8000000000000LL);
        }
        ++out;
    };
    auto clear_slots = [&]() { for (int i = 0; i < k; ++i) nslot[i] = 0; };

    int b = b0;
    for (; b < T.n_nb && A.desc[b].mode == MODE_IRREGULAR; ++b) {
        for (int seg = T.nb_seg_begin[b]; seg < T.nb_seg_begin[b + 1]; ++seg) {
            const int name = T.seg_read[seg], contig = T.seg_contig[seg];
            const bool filtered = A.qual[name] < A.qual_thresh;
            const int64_t L = A.R.contig_len[contig];
            const uint32_t *mf = A.R.mf + A.R.word_off[contig], *mr = A.R.mr + A.R.word_off[contig];
            for (int64_t r = T.seg_begin[seg]; r < T.seg_begin[seg + 1]; ++r) {
                ++run_rows;
                const int64_t idx = T.idx[r];
                const uint32_t fl = T.flags[r];
                if (name != last_read) first_idx = idx;                                  // :161-162
                if (filtered || (fl & MC_F_MODEL_N)) continue;                           // :167-168
                int rev;
                if ((name != last_read && (fl & MC_F_KMER_EQ)) || (name == last_read && idx > first_idx)) rev = 0;
                else rev = 1;                                                            // :169-174
                const int64_t pos = T.pos[r];
                const int off = first_m(rev ? mr : mf, L, pos, k);                       // :176
                if (truthy() && ((pos >= mpos + 1 && name == last_read) || name != last_read)) {   // :179
                    const bool reset = off < 0 || name != last_read || pos > mpos + A.skip_thresh + 1;
                    flush(r, !reset && off != 0);
                    if (reset) {                                                         // :242-245
                        clear_slots();
                        has_mpos = false;
                    } else {                                                             // :246-256
                        const int64_t old = mpos;
                        mpos = pos + off;
                        const int s = (int)(mpos - old < k ? mpos - old : k);
                        int psid[MC_MAX_K];
                        int64_t pn[MC_MAX_K];
                        for (int i = 0; i < k; ++i) { psid[i] = sid[i]; pn[i] = nslot[i]; }
                        for (int i = 0; i < k; ++i) {
                            const int src = (i - s + k) % k;
                            sid[i] = psid[src];
                            nslot[i] = i < s ? 0 : pn[src];
                        }
                    }
                }
                if (off >= 0) {                                                          // :269-287
                    if (truthy()) {
                        if (name != last_read) { has_mpos = false; clear_slots(); }
                        else if (rev != last_rev) has_mpos = false;                      // slots kept (:276-277)
                    }
                    if (!truthy()) { has_mpos = true; mpos = pos + off; }
                    last_read = name;
                    last_rev = rev;
                    last_seg = seg;
                    if (A.write) { const int2 e = T.evmu[r]; slots[(size_t)sid[off] * cap + nslot[off]] = (double)(e.x - e.y) / 10000.0; }
                    nslot[off] += 1;
                } else if (truthy()) {                                                   // :289-291
                    has_mpos = false;
                    clear_slots();
                }
            }
        }
    }
    // the run is over: its open window is closed by the next unfiltered row of the file (another read's)
    if (truthy()) {
        int64_t close_row = -1;
        int64_t rr = T.nb_row_begin[b];
        int bb = b;
        while (rr < T.n_rows) {
            while (bb + 1 < T.n_nb && T.nb_row_begin[bb + 1] <= rr) ++bb;
            if (A.desc[bb].filtered) { rr = T.nb_row_begin[bb + 1]; continue; }
            if (!(T.flags[rr] & MC_F_MODEL_N)) { close_row = rr; break; }
            ++rr;
        }
        if (close_row < 0 && A.tail_contig >= 0) close_row = T.n_rows;
        if (close_row >= 0) flush(close_row, false);
    }
    if (!A.write) {
        A.run_cnt[b0] = (int32_t)out;
        A.run_rows[b0] = (int32_t)run_rows;
    }
}

// fast records O (sorted by closing row) + literal records L (sorted) -> M (sorted); closing rows are distinct
__global__ void k_merge(DevRecords O, int64_t n_o, DevRecords L, int64_t n_l, DevRecords M, int k) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_o + n_l) return;
    const bool from_l = i >= n_o;
    const int64_t j = from_l ? i - n_o : i;
    const DevRecords &S = from_l ? L : O;
    const DevRecords &X = from_l ? O : L;
    const int64_t nx = from_l ? n_o : n_l;
    const int64_t key = S.close_row[j];
    int64_t lo = 0, hi = nx;
    while (lo < hi) {                                       // records of the other list that come first
        const int64_t mid = (lo + hi) >> 1;
        const int64_t xk = X.close_row[mid];
        if (xk < key || (xk == key && from_l)) lo = mid + 1; else hi = mid;
    }
    const int64_t d = j + lo;
    for (int f = 0; f < k; ++f) M.feats[d * k + f] = S.feats[j * k + f];
    M.site_pos[d] = S.site_pos[j];
    M.site_seg[d] = S.site_seg[j];
    M.close_row[d] = S.close_row[j];
    M.info[d] = S.info[j];
    M.wmask[d] = S.wmask[j];
    M.prob[d] = S.prob[j];
}

// ---------------------------------------------------------------------------------------------------
// K2: batched MLP forward, fp64 (predict_proba, :199).  One lane per record; the weights are scalar operands.
// ---------------------------------------------------------------------------------------------------
// tanh(x) = sign(x) (1 - 2 / (e^{2|x|} + 1)) and 1 / (1 + e^{-z}) = 1 - 1 / (e^{z} + 1) from one exponential each and a
// reciprocal that FOUR of them share, written out: the library's exp and the IEEE division cost ~100 instructions per tanh,
// and the classifier is 100 tanh per call -- this is 25.  Absolute error < 1e-15, far inside the 1e-9 the probabilities are
// held to.
//
// e^{2t} + 1 for t >= 0:  t = n ln2/2 + r, |r| <= ln2/4;  e^{2r} = 1 + 2r + r^2 G(r) with G of degree 9 -- the interpolant of
// (e^x - 1 - x) / x^2 at the Chebyshev nodes of [-ln2/2, ln2/2] (computed with 80 digits; relative error of e^x with the
// coefficients rounded to double: 1.6e-17; the Taylor polynomial needs two more terms), its coefficients scaled by powers
// of two for the argument x = 2r -- and e^{2t} = 2^n e^{2r}.  n = rint(2t / ln2) without v_rndne / v_cvt / v_ldexp
// (quarter-rate fp64 instructions): adding 1.5 * 2^52 leaves the integer in the low mantissa bits, and 2^n is built from it
// with one integer instruction.  e^0 = 1 exactly (tanh(0) = 0).
//
// The constants live in VGPRs on purpose (VgprConst): k2_mlp keeps the weights of four hidden units in SGPRs (72 of the
// ~100 there are), and constants the compiler put there as well were spilled to VGPR lanes and read back inside the loop.
struct VgprConst {
    double v;
    __device__ __forceinline__ explicit VgprConst(double x) : v(x) { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ operator double() const { return v; }
};
struct ExpConsts {
    VgprConst two_log2e{2.8853900817779268}, magic{6755399441055744.0}, half_ln2_hi{-0.3465735901845619},
        half_ln2_lo{-9.541074646352939e-11};
    VgprConst t_max4{87.5};        // e^{2t} <= e^175: the product of four such (e^{2t} + 1) stays finite; tanh(87.5) = 1 in double
    VgprConst g9{5.1405589494805136e-05}, g8{0.0002828297056809958}, g7{0.0014109321451518497}, g6{0.00634918945176432},
        g5{0.02539682542470863}, g4{0.08888888907016779}, g3{0.26666666666656197}, g2{0.6666666666659861},
        g1{1.3333333333333335}, g0{2.0000000000000004};
};

__device__ __forceinline__ double exp2t_plus1(double t, const ExpConsts &C) {      // (0 <= t <= 350)
    const double tt = fma(t, C.two_log2e, C.magic);
    const double n = tt - C.magic;
    double r = fma(n, C.half_ln2_hi, t);                    // ln2/2 in two pieces
    r = fma(n, C.half_ln2_lo, r);
    double p = C.g9;
    p = fma(p, r, C.g8);
    p = fma(p, r, C.g7);
    p = fma(p, r, C.g6);
    p = fma(p, r, C.g5);
    p = fma(p, r, C.g4);
    p = fma(p, r, C.g3);
    p = fma(p, r, C.g2);
    p = fma(p, r, C.g1);
    p = fma(p, r, C.g0);
    p = fma(p, r, 2.0);
    p = fma(p, r, 1.0);
    const int ni = __double2loint(tt);                      // n: 0 .. 1010
    return fma(p, __hiloint2double((ni + 1023) << 20, 0), 1.0);     // p 2^n + 1
}

// 1 / d for d >= 1: the hardware's reciprocal estimate and two Newton steps (relative error ~1e-16; no scaling needed, d is
// never small, huge d gives 0)
__device__ __forceinline__ double recip_ge1(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double tanh_1exp(double x, const ExpConsts &C) {
    const double q = recip_ge1(exp2t_plus1(fmin(fabs(x), C.t_max4), C));
    return copysign(fma(-2.0, q, 1.0), x);
}

// four at a time, step by step side by side (four independent chains in flight: the Horner scheme alone is a dependent
// sequence of twelve), and one reciprocal, of the product of the four denominators (each <= e^175 + 1)
__device__ __forceinline__ void tanh_4(double &x0, double &x1, double &x2, double &x3, const ExpConsts &C) {
    double t[4] = {fmin(fabs(x0), C.t_max4), fmin(fabs(x1), C.t_max4), fmin(fabs(x2), C.t_max4), fmin(fabs(x3), C.t_max4)};
    double tt[4], r[4], p[4], d[4];
#define MC_EACH for (int c = 0; c < 4; ++c)
#pragma unroll
    MC_EACH tt[c] = fma(t[c], C.two_log2e, C.magic);
#pragma unroll
    MC_EACH r[c] = fma(tt[c] - C.magic, C.half_ln2_hi, t[c]);
#pragma unroll
    MC_EACH r[c] = fma(tt[c] - C.magic, C.half_ln2_lo, r[c]);
#pragma unroll
    MC_EACH p[c] = fma(C.g9, r[c], C.g8);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g7);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g6);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g5);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g4);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g3);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g2);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g1);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], C.g0);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], 2.0);
#pragma unroll
    MC_EACH p[c] = fma(p[c], r[c], 1.0);
#pragma unroll
    MC_EACH d[c] = fma(p[c], __hiloint2double((__double2loint(tt[c]) + 1023) << 20, 0), 1.0);
#undef MC_EACH
    const double p01 = d[0] * d[1], p23 = d[2] * d[3];
    const double rall = recip_ge1(p01 * p23);
    const double r01 = rall * p23, r23 = rall * p01;        // 1 / (d0 d1), 1 / (d2 d3)
    x0 = copysign(fma(-2.0, r01 * d[1], 1.0), x0);
    x1 = copysign(fma(-2.0, r01 * d[0], 1.0), x1);
    x2 = copysign(fma(-2.0, r23 * d[3], 1.0), x2);
    x3 = copysign(fma(-2.0, r23 * d[2], 1.0), x3);
}

// 1 / (1 + e^{-z}) = (1 + tanh(z / 2)) / 2
__device__ __forceinline__ double logistic(double z, const ExpConsts &C) {
    const double q = recip_ge1(exp2t_plus1(fmin(0.5 * fabs(z), 350.0), C));   // 1 / (e^{|z|} + 1)
    return z >= 0.0 ? 1.0 - q : q;
}

// One lane per record, one hidden unit after the other inside the lane, the weights as SCALAR operands: the records a wave
// takes belong to one sub-model, so W1[:, j], b1[j], W2[j] are the same for its 64 lanes -- they come through the scalar
// cache into SGPRs (nine s_load'ed doubles per hidden unit) and the vector pipe issues nothing but the arithmetic:
// 7 + ~30 + 1 fp64 instructions per hidden unit and record, no LDS reads, no address arithmetic, no butterfly.
//
// A workgroup takes a contiguous stretch of the records (the pass's records divided evenly over the workgroups, K2B at a
// time) and
//   A. finds the records that are scored at all (skipped records and records whose context leaves the contig are not) and
//      lists them sub-model by sub-model in LDS, every sub-model's list padded to whole groups of 64; the read quality
//      (a chain of three dependent loads per record) is fetched here, for all records at once;
//   B. wave w computes quarter (w & 3) of the hidden units for groups (w >> 2), (w >> 2) + n_waves / 4, ...: the four
//      SIMDs of the CU carry the same load whatever the number of groups, and a workgroup of 1024 records (~11 groups of
//      the headline workload) keeps all of them busy.  Partial sums go to LDS;
//   C. the quarters are added in a fixed order (the result does not depend on which wave ran when) and the logistic
//      function gives the probability.
// The earlier version (eight lanes per record, pairs of records per lane group, weights in LDS: 126 LDS reads and ~1200
// VALU instructions per step of 16 records, 3.1 uneven waves per SIMD) took 53 us for the headline pass.
constexpr int K2B = 1024;                       // records per workgroup iteration
#ifndef MC_K2_THREADS
#define MC_K2_THREADS 1024
#endif
constexpr int K2_THREADS = MC_K2_THREADS;
constexpr int K2_WAVES = K2_THREADS / 64;
constexpr int K2_MAXM = 8;                      // sub-models (mc_ctx_set_mlp refuses more)
constexpr int K2_SLOTS = K2B + K2_MAXM * 64;    // list entries: every sub-model's part is padded to a multiple of 64
constexpr int K2_SUB = K2B / 64;                // 64-record pieces of a stretch: the unit of the list's prefix sums
static_assert(K2_WAVES >= 4 && K2_WAVES % 4 == 0 && K2B % K2_THREADS == 0, "four unit quarters; whole records per thread");
static_assert(K2_SUB * K2_MAXM <= K2_THREADS && K2_MAXM * 64 <= K2_THREADS, "one thread per (piece, sub-model) / per pad entry");

#ifdef MC_K2_TRACE      // (variant build for tools/k2_trace.py: 100 MHz time stamps of every wave's phases)
__device__ unsigned long long g_k2_trace[1024 * 16 * 16];
#define K2_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024) g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + (i)] = wall_clock64(); g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + 8 + (i)] = clock64(); } while (0)
#else
#define K2_STAMP(i) do { } while (0)
#endif


#define MC_SCALAR_MEM __attribute__((address_space(4)))    // constant address space: uniform loads from it are s_load

// NI_T: the number of inputs when it is known at compile time (7 for the reference's models: the loops over the inputs
// unroll exactly), 0: any.  The dot products use fma: nothing here has to reproduce a CPU sum bit for bit (the probabilities
// are held to 1e-9 against the oracle, 1e-12 against scikit-learn's known answers).
template <int NI_T>
__global__ __launch_bounds__(K2_THREADS) void k2_mlp(DevMlp M, const double *__restrict__ feats, int k,
                                                     const int32_t *__restrict__ site_seg, const int32_t *__restrict__ seg_read,
                                                     const double *__restrict__ qual, const uint32_t *__restrict__ info,
                                                     const uint8_t *__restrict__ submodel_in, int64_t n,
                                                     double *__restrict__ prob, const unsigned long long *__restrict__ n_dev,
                                                     const unsigned int *__restrict__ overflow) {
    if (overflow && *overflow) return;          // (pipelined pass with record buffers too small: it is repeated)
    if (n_dev) n = min(n, (int64_t)*n_dev);     // the count is on the device only (pipelined passes): n is the capacity
    constexpr int NX = NI_T ? NI_T : MC_MAX_K + 1;
    const int H = M.n_hidden, NI = NI_T ? NI_T : M.n_in, S = NI + 2, NM = min(M.n_models, K2_MAXM);
    __shared__ uint16_t s_list[K2_SLOTS];       // record (offset in the stretch) of every list entry; 0xFFFF: padding
    __shared__ double s_q[K2B];                 // read quality of the stretch's records
    __shared__ double s_part[4][K2_SLOTS];      // partial output sums of the four unit quarters
    __shared__ int s_cnt[K2_SUB][K2_MAXM], s_before[K2_SUB][K2_MAXM], s_tot[K2_MAXM], s_gmodel[K2_SLOTS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the compiler has to know that this is uniform: scalar loads)
    const unsigned long long below = (1ull << lane) - 1ull;
    const ExpConsts C;
    K2_STAMP(0);
    // Which quarter of the hidden units a wave takes: the one of the SIMD it runs on, so that the four SIMDs of the CU carry
    // a quarter of the arithmetic each whatever the number of groups (the waves of one SIMD share its groups).  The waves
    // register before the first barrier of the first stretch.  (If the workgroup's waves did not land on all four SIMDs:
    // by wave number.)
    __shared__ uint32_t s_simd_of_wave[K2_WAVES / 4];       // a byte per wave
    const int simd = (int)__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4) & 3;      // HW_ID[5:4]
    int quarter = 0, g_first = 0, g_step = 1;
    bool placed = false;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    K2_STAMP(1);
    for (int64_t base = lo; base < hi; base += K2B) {
        // ---- A: the lists
        // (the read quality is a chain of three dependent loads -- segment, read, quality: it starts with the first load of
        // the stretch and is only waited for when the lists are done)
        int mi[K2B / K2_THREADS], rank[K2B / K2_THREADS];
        double qv[K2B / K2_THREADS];
#pragma unroll
        for (int i = 0; i < K2B / K2_THREADS; ++i) {
            const int off = i * K2_THREADS + tid;
            const int64_t r = base + off;
            mi[i] = 255;                            // sub-model of record r (255: not scored here)
            qv[i] = 0.0;
            if (r < hi) {
                if (submodel_in) mi[i] = submodel_in[r];
                else {
                    const uint32_t inf = info[r];
                    const int32_t seg = site_seg[r];
                    if (!(inf & (MC_I_TOO_MANY | MC_I_EDGE))) {
                        mi[i] = M.sub_of_char[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
                        qv[i] = qual[seg_read[seg]];
                    }
                }
            }
            rank[i] = 0;
            for (int m = 0; m < NM; ++m) {          // (a key outside the models is the KeyError path, :218: the host decides)
                const unsigned long long bal = __ballot(mi[i] == m);
                if (mi[i] == m) rank[i] = __popcll(bal & below);
                if (lane == 0) s_cnt[off >> 6][m] = __popcll(bal);
            }
        }
        if (!placed && lane == 0) reinterpret_cast<uint8_t *>(s_simd_of_wave)[wave] = (uint8_t)simd;
        K2_STAMP(2);
        __syncthreads();
        if (!placed) {
            uint32_t per_simd = 0;                  // a byte per SIMD: its waves
            int slot = 0;                           // waves of this wave's SIMD with a smaller number
            for (int w4 = 0; w4 < K2_WAVES / 4; ++w4) {
                const uint32_t four = __builtin_amdgcn_readfirstlane(s_simd_of_wave[w4]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int sd = (four >> (8 * b)) & 3;
                    per_simd += 1u << (8 * sd);
                    slot += (w4 * 4 + b < wave && sd == simd) ? 1 : 0;
                }
            }
            const bool by_simd = (per_simd & 0xFFu) && (per_simd & 0xFF00u) && (per_simd & 0xFF0000u) && (per_simd & 0xFF000000u);
            quarter = by_simd ? simd : wave & 3;
            g_first = by_simd ? slot : wave >> 2;
            g_step = by_simd ? (int)((per_simd >> (8 * simd)) & 0xFFu) : K2_WAVES / 4;
            placed = true;
        }
        if (tid < K2_SUB * K2_MAXM) {               // thread (piece c, sub-model m): the sub-model's records in pieces before c
            const int c = tid / K2_MAXM, m = tid % K2_MAXM;
            int all = 0, before = 0;
            if (m < NM)
                for (int c2 = 0; c2 < K2_SUB; ++c2) { const int v = s_cnt[c2][m]; all += v; before += c2 < c ? v : 0; }
            s_before[c][m] = before;
            if (c == 0) s_tot[m] = all;
        }
        __syncthreads();
        int n_groups = 0;
        {
            int start[K2_MAXM];                     // first list entry of every sub-model
#pragma unroll
            for (int m = 0; m < K2_MAXM; ++m) { start[m] = n_groups * 64; n_groups += (s_tot[m] + 63) >> 6; }
#pragma unroll
            for (int i = 0; i < K2B / K2_THREADS; ++i) {
                const int off = i * K2_THREADS + tid;
                if (mi[i] < NM) {
                    int st = 0;
#pragma unroll
                    for (int m = 0; m < K2_MAXM; ++m) st = mi[i] == m ? start[m] : st;
                    s_list[st + s_before[off >> 6][mi[i]] + rank[i]] = (uint16_t)off;
                }
            }
            if (tid < K2_MAXM * 64) {               // padding of sub-model tid / 64, and the sub-model of its groups
                const int m = tid >> 6;
                int st = 0;
#pragma unroll
                for (int m2 = 0; m2 < K2_MAXM; ++m2) st = m == m2 ? start[m2] : st;
                const int tot = s_tot[m], g = (tot + 63) >> 6;
                if (tot + lane < g * 64) s_list[st + tot + lane] = 0xFFFF;
                if (lane < g) s_gmodel[(st >> 6) + lane] = m;
            }
        }
#pragma unroll
        for (int i = 0; i < K2B / K2_THREADS; ++i) s_q[i * K2_THREADS + tid] = qv[i];
        __syncthreads();
        K2_STAMP(3);
        // ---- B: a quarter of the hidden units for every fourth (eighth ...) group
        const int u0 = quarter * H / 4, u1 = (quarter + 1) * H / 4;
        for (int g = g_first; g < n_groups; g += g_step) {
            const int mdl = __builtin_amdgcn_readfirstlane(s_gmodel[g]);
            const int e = s_list[g * 64 + lane];
            const int off = e == 0xFFFF ? s_list[g * 64] : e;       // (padding lanes compute the group's first record again)
            const int64_t r = base + off;
            double x[NX];
            if (submodel_in) {                       // plain batched call: X rows of n_in values
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = i < NI ? feats[r * NI + i] : 0.0;
            } else {                                 // flush records: k slot means + read quality (:189-193)
                const double q = s_q[off];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = i < k ? feats[r * k + i] : (i == k ? q : 0.0);
            }
            const MC_SCALAR_MEM double *wu = (const MC_SCALAR_MEM double *)M.wu + ((size_t)mdl * H + u0) * S;
            double z = 0.0;
            int u = u0;
            for (; u + 4 <= u1; u += 4, wu += 4 * S) {          // four independent chains: the fp64 tanh is a long dependent sequence
                double a0 = x[0] * wu[0], a1 = x[0] * wu[S], a2 = x[0] * wu[2 * S], a3 = x[0] * wu[3 * S];
#pragma unroll
                for (int i = 1; i < NX; ++i)
                    if (i < NI) {
                        a0 = fma(x[i], wu[i], a0);
                        a1 = fma(x[i], wu[S + i], a1);
                        a2 = fma(x[i], wu[2 * S + i], a2);
                        a3 = fma(x[i], wu[3 * S + i], a3);
                    }
                a0 += wu[NI]; a1 += wu[S + NI]; a2 += wu[2 * S + NI]; a3 += wu[3 * S + NI];     // (a second scalar operand in the first fma would cost two moves)
                tanh_4(a0, a1, a2, a3, C);
                z = fma(a0, wu[NI + 1], z);
                z = fma(a1, wu[S + NI + 1], z);
                z = fma(a2, wu[2 * S + NI + 1], z);
                z = fma(a3, wu[3 * S + NI + 1], z);
            }
            for (; u < u1; ++u, wu += S) {
                double a0 = x[0] * wu[0];
#pragma unroll
                for (int i = 1; i < NX; ++i)
                    if (i < NI) a0 = fma(x[i], wu[i], a0);
                z = fma(tanh_1exp(a0 + wu[NI], C), wu[NI + 1], z);
            }
            s_part[quarter][g * 64 + lane] = z;
        }
        K2_STAMP(4);
        __syncthreads();
        K2_STAMP(5);
        // ---- C: the output unit
        for (int t = tid; t < n_groups * 64; t += K2_THREADS) {
            const int e = s_list[t];
            if (e == 0xFFFF) continue;
            const double z = ((s_part[0][t] + s_part[1][t]) + s_part[2][t]) + s_part[3][t];
            prob[base + e] = logistic(z + M.b2[s_gmodel[t >> 6]], C);
        }
        // (no barrier here: what the next stretch writes before its first barrier -- s_q, s_cnt -- was last read before the
        // barrier above)
        K2_STAMP(6);
#ifdef MC_K2_TRACE
        if (lane == 0 && blockIdx.x < 1024)
            g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + 7] = (unsigned)simd | (unsigned)quarter << 4 | (unsigned)g_first << 8 | (unsigned)g_step << 16 | (unsigned long long)n_groups << 24;
#endif
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: random-forest predict_proba (classifier RF, train_model.py:39-45; call site :199).  One lane per record: inputs cast
// to float32 (scikit-learn's DTYPE), each tree walked with `x[feature] <= threshold` to a leaf, p1 = v1/(v0+v1) per tree
// (predict_proba normalises the leaf values), summed over the trees in order and divided by their number.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k3_forest(DevForest F, const double *__restrict__ feats, int k,
                                                const int32_t *__restrict__ site_seg, const int32_t *__restrict__ seg_read,
                                                const double *__restrict__ qual, const uint32_t *__restrict__ info,
                                                const uint8_t *__restrict__ submodel_in, int64_t n,
                                                double *__restrict__ prob, const unsigned long long *__restrict__ n_dev,
                                                const unsigned int *__restrict__ overflow) {
    __shared__ double s_x[64][MC_MAX_K + 2];
    if (overflow && *overflow) return;          // (pipelined pass with record buffers too small: it is repeated)
    if (n_dev) n = min(n, (int64_t)*n_dev);     // the count is on the device only (pipelined passes): n is the capacity
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n) return;
    double *x = s_x[threadIdx.x];
    const int NI = F.n_in;
    int mi;
    if (submodel_in) {
        mi = submodel_in[r];
        for (int i = 0; i < NI; ++i) x[i] = (double)(float)feats[r * NI + i];
    } else {
        const uint32_t inf = info[r];
        if (inf & (MC_I_TOO_MANY | MC_I_EDGE)) return;
        mi = F.sub_of_char[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
        for (int i = 0; i < k; ++i) x[i] = (double)(float)feats[r * k + i];
        x[k] = (double)(float)qual[seg_read[site_seg[r]]];
    }
    if (mi >= F.n_models) return;
    const int t0 = F.model_tree_off[mi], t1 = F.model_tree_off[mi + 1];
    double sum = 0.0;
    for (int t = t0; t < t1; ++t) {
        int node = F.tree_node_off[t];
        int l;
        while ((l = F.left[node]) >= 0) node = (x[F.feature[node]] <= F.threshold[node]) ? l : F.right[node];
        const double v0 = F.value[2 * (size_t)node], v1 = F.value[2 * (size_t)node + 1];
        double norm = (-0.0 + v0) + v1;
        if (norm == 0.0) norm = 1.0;
        sum += v1 / norm;
    }
    prob[r] = sum / (double)(t1 - t0);
}

// ---------------------------------------------------------------------------------------------------
// K3': the closed-form classifiers -- logistic regression (-c LR) and Gaussian naive Bayes (-c NBC), train_model.py:55-60;
// call site :199.  One lane per record, fp64, the sums in index order (scikit-learn: a BLAS dot / numpy sums over seven
// terms: agreement to ~1e-16 relative, pinned at 1e-12 against captured predict_proba).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k3_simple(DevSimple S, const double *__restrict__ feats, int k,
                                                const int32_t *__restrict__ site_seg, const int32_t *__restrict__ seg_read,
                                                const double *__restrict__ qual, const uint32_t *__restrict__ info,
                                                const uint8_t *__restrict__ submodel_in, int64_t n,
                                                double *__restrict__ prob, const unsigned long long *__restrict__ n_dev,
                                                const unsigned int *__restrict__ overflow) {
    if (overflow && *overflow) return;
    if (n_dev) n = min(n, (int64_t)*n_dev);
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n) return;
    double x[MC_MAX_K + 1];
    const int NI = S.n_in;
    int mi;
    if (submodel_in) {
        mi = submodel_in[r];
        for (int i = 0; i < MC_MAX_K + 1; ++i) x[i] = i < NI ? feats[r * NI + i] : 0.0;
    } else {
        const uint32_t inf = info[r];
        if (inf & (MC_I_TOO_MANY | MC_I_EDGE)) return;
        mi = S.sub_of_char[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
        for (int i = 0; i < MC_MAX_K + 1; ++i) x[i] = i < k ? feats[r * k + i] : 0.0;
        const double q = qual[seg_read[site_seg[r]]];
        for (int i = 0; i < MC_MAX_K + 1; ++i) if (i == k) x[i] = q;
    }
    if (mi >= S.n_models) return;
    const double *P = S.params + (size_t)mi * S.stride;
    if (S.kind == MC_CLF_LOGISTIC) {
        double d = 0.0;
        for (int i = 0; i < MC_MAX_K + 1; ++i) if (i < NI) d += x[i] * P[i];
        d += P[NI];
        // scipy.special.expit: 1 / (1 + exp(-d)), the large-|d| ends as it writes them
        prob[r] = d >= 0.0 ? 1.0 / (1.0 + exp(-d)) : exp(d) / (1.0 + exp(d));
    } else {
        // GaussianNB._joint_log_likelihood: log prior - 0.5 * sum(log(2 pi var)) - 0.5 * sum((x - theta)^2 / var)
        double jll[2];
        for (int cls = 0; cls < 2; ++cls) {
            const double *theta = P + (size_t)cls * 2 * NI, *var = theta + NI;
            double a = 0.0, b2 = 0.0;
            for (int i = 0; i < MC_MAX_K + 1; ++i)
                if (i < NI) {
                    a += log(2.0 * 3.14159265358979323846 * var[i]);
                    const double t = x[i] - theta[i];
                    b2 += (t * t) / var[i];
                }
            jll[cls] = P[4 * (size_t)NI + cls] + (-0.5 * a) - 0.5 * b2;
        }
        // exp(jll1 - logsumexp(jll0, jll1))
        const double mx = fmax(jll[0], jll[1]);
        const double lse = mx + log(exp(jll[0] - mx) + exp(jll[1] - mx));
        prob[r] = exp(jll[1] - lse);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Pipelined passes: every pass in flight has its own counters, strand-resolve output and record set; the host reads
// the counters on the copy stream and then moves exactly n records with the DMA engines while the next pass computes.
// (A kernel that stores the records straight into pinned host memory reaches the same 54 GB/s, but every kernel of the
// next pass that ENDS while it runs waits for it: the end-of-kernel cache write-back queues behind its PCIe writes --
// measured with rocprofv3, see DESIGN.md.  DMA copies do not go through the shader caches.)
// ---------------------------------------------------------------------------------------------------
// What a pipelined pass hands to the DMA engine is ONE block: four narrow columns of all n flush records (closing row --
// 32 bits wide for tables below 2^31 - 1 rows --, site, segment, info) and, behind them, the slot means and the probability
// of the records that are calls -- compacted: a record with MC_I_TOO_MANY is only counted by the host (:239), nothing reads
// its means, and at 6 % skips it is every third record.  The row of record j in the compacted part is the number of
// records before it without MC_I_TOO_MANY: the host derives it where it needs it (mc_calls_view) -- the copy-out was what
// bounded a pass (PCIe, 55 GB/s), so bytes dropped here were time (16 instead of 24 narrow bytes per record: 12.8 -> 11.2 MB
// per pass of the headline workload; the slot means as 32-bit integers where they can be, see pack_tail: 8.75 MB, and the
// kernels of the ctx stream are the bound).  Two small kernels: per-chunk counts of kept records, then every
// workgroup sums the counts before its chunk and packs the chunk.  The pass's counters go to pinned host memory from here
// as well (a 96-byte store over PCIe): the host reads them after hipEventSynchronize(ev_done) and enqueues the transfer
// at once, without a read-back on the copy stream in between.
// (512 x 512: beside the next pass's scan the two kernels wait for memory most of the time, and twice the lanes have twice the
// loads in flight -- k_pack 96 -> 46 us there, and the scan it runs beside 106 -> 96 us; pipelined pass with 256 x 256: 0.2013 ms,
// 448 or 512: 0.193-0.195, 576: 0.203, 640: 0.207, 768: 0.215, 1024: 0.231; 256 x 512, 384 x 384: 0.195-0.197)
// (PACK_WGS, PACK_THREADS: defined with the emit kernels, which count per packing chunk)

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

// chunk_cnt[PACK_PAD * b] = kept records of chunk b, chunk_cnt[PACK_PAD * b + 1] = their wide slots
__global__ __launch_bounds__(PACK_THREADS) void k_pack_count(DevRecords O, const Counters *__restrict__ cnt, int k,
                                                             unsigned long long *__restrict__ chunk_cnt) {
    __shared__ unsigned int s_wave[2][PACK_THREADS / 64];
    const int64_t n = cnt->overflow ? 0 : min((int64_t)cnt->n_records, O.capacity);
    const int64_t per = (n + PACK_WGS - 1) / PACK_WGS;
    const int64_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    unsigned int kept = 0, wide = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PACK_THREADS) {
        if (O.info[i] & MC_I_TOO_MANY) continue;
        kept += 1u;
        const unsigned wm = O.wmask[i];                      // (k1_emit's note; 0xFF: a record of the rare paths, looked at here)
        if (wm != 0xFFu) wide += (unsigned)__popc(wm);
        else
            for (int f = 0; f < k; ++f) {
                int32_t d;
                wide += slot_is_narrow(O.feats[i * k + f], &d) ? 0u : 1u;
            }
    }
    for (int o = 32; o > 0; o >>= 1) { kept += __shfl_xor(kept, o); wide += __shfl_xor(wide, o); }
    if ((threadIdx.x & 63) == 0) { s_wave[0][threadIdx.x >> 6] = kept; s_wave[1][threadIdx.x >> 6] = wide; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0, w = 0;
        for (int j = 0; j < PACK_THREADS / 64; ++j) { t += s_wave[0][j]; w += s_wave[1][j]; }
        chunk_cnt[PACK_PAD * blockIdx.x] = t;
        chunk_cnt[PACK_PAD * blockIdx.x + 1] = w;
    }
}

__global__ __launch_bounds__(PACK_THREADS) void k_pack(DevRecords O, const Counters *__restrict__ cnt,
                                                       const unsigned long long *__restrict__ chunk_cnt,
                                                       unsigned char *__restrict__ out, int k, int close32,
                                                       Counters *__restrict__ host_status) {
    static_assert(PACK_WGS <= PACK_THREADS && PACK_THREADS % 64 == 0, "one chunk count per thread");
    __shared__ unsigned long long s_sum[4][PACK_THREADS / 64];
    __shared__ unsigned int s_wave[2][PACK_THREADS / 64];
    __shared__ double s_feats[PACK_THREADS * MC_MAX_K];     // the strip's slot means, loaded with consecutive lanes on consecutive words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // kept records (and their wide slots) before this chunk, and in all chunks
    unsigned long long v = tid < PACK_WGS ? chunk_cnt[PACK_PAD * tid] : 0ull, before = tid < (int)blockIdx.x ? v : 0ull;
    unsigned long long w = tid < PACK_WGS ? chunk_cnt[PACK_PAD * tid + 1] : 0ull, wbefore = tid < (int)blockIdx.x ? w : 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o); before += __shfl_xor(before, o);
        w += __shfl_xor(w, o); wbefore += __shfl_xor(wbefore, o);
    }
    if (lane == 0) { s_sum[0][wave] = v; s_sum[1][wave] = before; s_sum[2][wave] = w; s_sum[3][wave] = wbefore; }
    __syncthreads();
    unsigned long long total = 0, base = 0, total_wide = 0, wbase = 0;
    for (int j = 0; j < PACK_THREADS / 64; ++j) { total += s_sum[0][j]; base += s_sum[1][j]; total_wide += s_sum[2][j]; wbase += s_sum[3][j]; }
    constexpr unsigned head_words = offsetof(Counters, end_of_head) / 4;      // everything the host looks at
    constexpr int kept_word = (int)(offsetof(Counters, n_kept) / 4);          // (n_kept and n_wide: two words each, set below)
    static_assert(offsetof(Counters, n_wide) == offsetof(Counters, n_kept) + 8, "n_kept, n_wide side by side");
    if (blockIdx.x == 0) {
        if (tid < (int)head_words && (tid < kept_word || tid >= kept_word + 4))
            reinterpret_cast<volatile unsigned int *>(host_status)[tid] = reinterpret_cast<const unsigned int *>(cnt)[tid];
        if (tid == 0) {
            *reinterpret_cast<volatile unsigned long long *>(&host_status->n_kept) = total;
            *reinterpret_cast<volatile unsigned long long *>(&host_status->n_wide) = total_wide;
        }
    }
    if (cnt->overflow) return;                                           // (the host runs such a pass again, synchronously)
    const int64_t n = min((int64_t)cnt->n_records, O.capacity);
    const int64_t per = (n + PACK_WGS - 1) / PACK_WGS;
    const int64_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    const PackLayout L = pack_layout(n, close32);
    const PackTail T = pack_tail(L.feats, (size_t)total, k, (size_t)total_wide);
    int64_t *o_close = reinterpret_cast<int64_t *>(out);
    int32_t *o_close32 = reinterpret_cast<int32_t *>(out);
    int32_t *o_pos = reinterpret_cast<int32_t *>(out + L.pos);
    int32_t *o_seg = reinterpret_cast<int32_t *>(out + L.seg);
    uint32_t *o_info = reinterpret_cast<uint32_t *>(out + L.info);
    int32_t *o_lo = reinterpret_cast<int32_t *>(out + T.lo32);
    double *o_prob = reinterpret_cast<double *>(out + T.prob);
    uint8_t *o_mask = out + T.wmask;
    uint32_t *o_hi = reinterpret_cast<uint32_t *>(out + T.hi32);
    for (int64_t s = lo; s < hi; s += PACK_THREADS) {
        const int64_t i = s + tid;
        const bool valid = i < hi;
        const uint32_t info = valid ? O.info[i] : MC_I_TOO_MANY;
        const bool keep = !(info & MC_I_TOO_MANY);
        {
            const int64_t n_here = min((int64_t)PACK_THREADS, hi - s) * k;
            for (int64_t j = tid; j < n_here; j += PACK_THREADS) s_feats[j] = O.feats[s * k + j];
        }
        __syncthreads();
        // the record's slot means: 32-bit integers where they are fl(d / 1e4), both halves where they are not
        int32_t lo32[MC_MAX_K];
        uint32_t hi32[MC_MAX_K];
        unsigned int wmask = 0, n_w = 0;
        if (keep) {
            const unsigned noted = O.wmask[i];
            for (int f = 0; f < k; ++f) {
                const double x = s_feats[tid * k + f];
                int32_t d = 0;
                const bool narrow = noted != 0xFFu ? !((noted >> f) & 1u) : slot_is_narrow(x, &d);
                if (narrow) lo32[f] = noted != 0xFFu ? (int32_t)rint(x * 1e4) : d;
                else {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
                    lo32[f] = (int32_t)(uint32_t)bits;
                    hi32[n_w++] = (uint32_t)(bits >> 32);
                    wmask |= 1u << f;
                }
            }
        }
        // places: rank among the strip's kept records; wide slots before this record's
        const unsigned long long kmask = __ballot(keep);
        unsigned int w_incl = n_w;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int y = __shfl_up(w_incl, o);
            if (lane >= o) w_incl += y;
        }
        if (lane == 63) s_wave[1][wave] = w_incl;
        if (lane == 0) s_wave[0][wave] = (unsigned int)__popcll(kmask);
        __syncthreads();
        unsigned int in_strip = (unsigned int)__popcll(kmask & ((1ull << lane) - 1ull)), strip = 0, w_off = w_incl - n_w, w_strip = 0;
        for (int j = 0; j < PACK_THREADS / 64; ++j) {
            if (j < wave) { in_strip += s_wave[0][j]; w_off += s_wave[1][j]; }
            strip += s_wave[0][j];
            w_strip += s_wave[1][j];
        }
        if (valid) {
            if (close32) o_close32[i] = (int32_t)O.close_row[i];
            else o_close[i] = O.close_row[i];
            o_pos[i] = O.site_pos[i];
            o_seg[i] = O.site_seg[i];
            o_info[i] = info;
            if (keep) {
                const unsigned long long row = base + in_strip;
                o_prob[row] = O.prob[i];
                o_mask[row] = (uint8_t)wmask;
                for (int f = 0; f < k; ++f) o_lo[row * k + f] = lo32[f];
                for (unsigned int j = 0; j < n_w; ++j) o_hi[wbase + w_off + j] = hi32[j];
            }
        }
        __syncthreads();
        base += strip;
        wbase += w_strip;
    }
}

// ===================================================================================================
// host side
// ===================================================================================================
constexpr int MC_PASSES_IN_FLIGHT = 4;   // one being copied out, one computing, two queued (the host enqueues while it copies)

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
        int slot = -1;             // table slot the pass scans
        unsigned long long pass_no = 0;   // what Counters.irregular_pass holds if the pass classified a block irregular
        const double *qual = nullptr;   // read qualities it was enqueued with
        int32_t n_qual = 0;
        std::vector<void *> dev_allocs, k0_allocs;
    } ab[MC_PASSES_IN_FLIGHT];
    hipStream_t side_stream = nullptr;   // classifier and packing of the pipelined passes
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
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k1_emit, 256, 0) == hipSuccess && occ > 0) c->emit_wgs = occ;
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
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    for (auto &b : c->ab)
        for (hipEvent_t e : {b.ev_k0_start, b.ev_k0_end, b.ev_scan_start, b.ev_scan_end, b.ev_emit_end, b.ev_k2_start, b.ev_k2_end, b.ev_done, b.ev_copied})
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
        dev_alloc(P, &c->group_sum, (size_t)(nt / GROUP + 2)) || dev_alloc(P, &c->tile_cnt, (size_t)nt + 1) || dev_alloc(P, &c->tile_half, (size_t)nt + 1))
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
        S.tmpl_ref = -1;          // (the name-block templates too: k_nb_template is part of what a table costs when it is scanned once)
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
    HIP_TRY(hipStreamSynchronize(c->stream));
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
    UP(M.sub_of_char, sub_of_char, 256, c->mlp_allocs);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mc_ctx_set_forest(mc_ctx *c, int32_t n_models, int32_t n_in, const int32_t *model_tree_off,
                                 const int32_t *tree_node_off, const int32_t *left, const int32_t *right,
                                 const int32_t *feature, const double *threshold, const double *value,
                                 const uint8_t *sub_of_char) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
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
    HIP_TRY(hipStreamSynchronize(c->stream));
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

// the classifier of the context -- MLP (k2_mlp), forest (k3_forest) or one of the closed forms (k3_simple) -- over n records
// (n_dev: the count is on the device, n is the capacity)
static void launch_k2(mc_ctx *c, unsigned grid, hipStream_t st, const double *feats, int k, const int32_t *site_seg,
                      const int32_t *seg_read, const double *qual, const uint32_t *info, const uint8_t *submodel_in, int64_t n,
                      double *prob, const unsigned long long *n_dev, const unsigned int *overflow);
static unsigned k2_grid(const mc_ctx *c, int64_t n);
static int classifier_inputs(const mc_ctx *c) {
    return c->F.left ? c->F.n_in : (c->Sc.params ? c->Sc.n_in : (c->M.W1 ? c->M.n_in : 0));
}
static void launch_classifier(mc_ctx *c, hipStream_t st, const double *feats, int k, const int32_t *site_seg, const int32_t *seg_read,
                              const double *qual, const uint32_t *info, const uint8_t *submodel_in, int64_t n, double *prob,
                              const unsigned long long *n_dev, const unsigned int *overflow) {
    if (n <= 0) return;
    if (c->F.left)             // (one lane per record; with the count on the device a workgroup beyond it ends at once)
        hipLaunchKernelGGL(k3_forest, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, c->F, feats, k, site_seg, seg_read, qual, info,
                           submodel_in, n, prob, n_dev, overflow);
    else if (c->Sc.params)
        hipLaunchKernelGGL(k3_simple, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, c->Sc, feats, k, site_seg, seg_read, qual, info,
                           submodel_in, n, prob, n_dev, overflow);
    else
        launch_k2(c, k2_grid(c, n), st, feats, k, site_seg, seg_read, qual, info, submodel_in, n, prob, n_dev, overflow);
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

// k2_mlp workgroups: the records are divided evenly over them (the kernel does that with the count on the device); one
// per CU, fewer for a handful of records
#ifndef MC_K2_WG_PER_CU
#define MC_K2_WG_PER_CU 1
#endif
static unsigned k2_grid(const mc_ctx *c, int64_t n) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)c->n_cu * MC_K2_WG_PER_CU));
}

// the classifier kernel: the 7-input instance (k = 6, the reference's models) or the general one
static void launch_k2(mc_ctx *c, unsigned grid, hipStream_t st, const double *feats, int k, const int32_t *site_seg,
                      const int32_t *seg_read, const double *qual, const uint32_t *info, const uint8_t *submodel_in, int64_t n,
                      double *prob, const unsigned long long *n_dev, const unsigned int *overflow) {
    if (c->M.n_in == 7)
        hipLaunchKernelGGL(k2_mlp<7>, dim3(grid), dim3(K2_THREADS), 0, st, c->M, feats, k, site_seg, seg_read, qual,
                           info, submodel_in, n, prob, n_dev, overflow);
    else
        hipLaunchKernelGGL(k2_mlp<0>, dim3(grid), dim3(K2_THREADS), 0, st, c->M, feats, k, site_seg, seg_read, qual,
                           info, submodel_in, n, prob, n_dev, overflow);
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
    hipLaunchKernelGGL(k_literal, dim3(g), dim3(64), 0, c->stream, LA);
    hipLaunchKernelGGL(k1_group_scan, dim3((unsigned)n_groups), dim3(GROUP), 0, c->stream, (const int32_t *)run_cnt,
                       (int64_t)T.n_nb, cnt_local, cnt_group);
    hipLaunchKernelGGL(k1_group_scan, dim3((unsigned)n_groups), dim3(GROUP), 0, c->stream, (const int32_t *)run_rows,
                       (int64_t)T.n_nb, rows_local, rows_group);
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
    hipLaunchKernelGGL(k_literal, dim3(g), dim3(64), 0, c->stream, LA);
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((n_fast + n_lit + 255) / 256)), dim3(256), 0, c->stream, c->O, n_fast, L, n_lit, M, k);
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
                hipLaunchKernelGGL(k_summarize, dim3((unsigned)((S.T.n_rows / 4 + 256) / 256)), dim3(256), 0, st, S.T);
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
    if (c->cur >= 0 && c->slots[c->cur].tmpl_ref != c->ref_version) {      // once per (table, reference)
        hipLaunchKernelGGL(k_nb_template, dim3((unsigned)((T.n_nb + 255) / 256)), dim3(256), 0, st, T, c->R);
        c->slots[c->cur].tmpl_ref = c->ref_version;
    }
    // (four waves per name block unless the reads are short: see k0_first_site)
    if (MC_K0_WAVES > 1 && T.n_rows >= (int64_t)T.n_nb * 2048)
        hipLaunchKernelGGL(k0_first_site<MC_K0_WAVES>, dim3((unsigned)T.n_nb), dim3(64 * MC_K0_WAVES), 0, st, T, c->R,
                           c->qual, prm->qual_thresh, k, K.desc, K.nb_f0, cnt, lookback ? 0 : 1,
                           prm->skip_thresh, pass_no, plan.first ? 1 : 0);
    else
        hipLaunchKernelGGL(k0_first_site<1>, dim3((unsigned)(((int64_t)T.n_nb * 64 + 255) / 256)), dim3(256), 0, st, T, c->R,
                           c->qual, prm->qual_thresh, k, K.desc, K.nb_f0, cnt, lookback ? 0 : 1,
                           prm->skip_thresh, pass_no, plan.first ? 1 : 0);
    if (lookback)
        hipLaunchKernelGGL(k0_classify, dim3((unsigned)((T.n_nb + 255) / 256)), dim3(256), 0, st, T, c->R,
                           K.desc, K.nb_f0, prm->entry_read, k, prm->skip_thresh, cnt, pass_no);
    if (extend)
        hipLaunchKernelGGL(k0_extend, dim3((unsigned)((T.n_nb + 255) / 256)), dim3(256), 0, st, T, K.desc,
                           (const int64_t *)K.nb_f0, prm->entry_read, cnt, pass_no);
    return 0;
}

// K1 (scan, order, emit) of one pass into the record set O on stream st; ev_scan_end is recorded after the scan.
static int enqueue_k1(mc_ctx *c, const mc_params *prm, const K0Set &K, Counters *cnt, const DevRecords &O, hipStream_t st,
                      hipEvent_t ev_scan_end, K1Args *out_args, Payload *sorted, int64_t *rare_list, unsigned long long pass_no,
                      const PassPlan &plan, hipEvent_t ev_emit_end = nullptr, unsigned long long *chunk_cnt = nullptr) {
    const DevTable &T = c->T;
    K1Args A;
    A.T = T; A.R = c->R; A.desc = K.desc; A.tile_chunk = c->tile_chunk; A.payload = c->payload;
    A.payload_cap = c->payload_cap; A.tile_cnt = c->tile_cnt; A.tile_half = c->tile_half;
    A.tile_local = c->tile_local; A.group_sum = c->group_sum; A.O = O; A.cnt = cnt; A.k = prm->k;
    A.skip_thresh = prm->skip_thresh; A.tail_contig = prm->tail_contig; A.rare_list = rare_list;
    A.pass_no = pass_no;
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
    // one wave per tile; the instance with the small candidate list unless marked positions are dense (a one-base motif)
    const dim3 grid((unsigned)T.n_tiles);
    constexpr int CG_DENSE = CHUNK / 8 + 2;      // (every unit of a chunk; one cut by the boundary of its two blocks is listed twice)
    if (dense) {
        if (plan.scan_mode == SCAN_VALIDATE) hipLaunchKernelGGL((k1_scan<CG_DENSE, SCAN_VALIDATE>), grid, dim3(64), 0, st, A);
        else hipLaunchKernelGGL((k1_scan<CG_DENSE, SCAN_STREAM>), grid, dim3(64), 0, st, A);
    } else {
        if (plan.scan_mode == SCAN_VALIDATE) hipLaunchKernelGGL((k1_scan<64, SCAN_VALIDATE>), grid, dim3(64), 0, st, A);
        else if (plan.scan_mode == SCAN_STREAM) hipLaunchKernelGGL((k1_scan<64, SCAN_STREAM>), grid, dim3(64), 0, st, A);
        else hipLaunchKernelGGL((k1_scan<64, SCAN_SUMMARY>), grid, dim3(64), 0, st, A);
    }
    if (ev_scan_end) HIP_TRY(hipEventRecord(ev_scan_end, st));
    hipLaunchKernelGGL(k1_group_scan, dim3((unsigned)((T.n_tiles + GROUP - 1) / GROUP)), dim3(GROUP), 0, st,
                       (const int32_t *)c->tile_cnt, T.n_tiles, c->tile_local, c->group_sum);
    // (dense references: a workgroup per tile, the mean of every position once, see k1_emit_runs -- which takes the payloads where
    // the scan left them: no gather)
    hipLaunchKernelGGL(k1_list, dim3((unsigned)((T.n_tiles * 8 + 255) / 256)), dim3(256), 0, st, A, sorted, runs ? 0 : 1);
    // (ev_emit_end rides on the emit's own dispatch packet: a hipEventRecord behind it is a barrier packet of its own and
    // costs the queue 5-9 us)
    const dim3 emit_grid((unsigned)std::min<int64_t>((O.capacity * EG + 255) / 256, (int64_t)c->n_cu * c->emit_wgs));
    if (ev_emit_end && MC_EVENTS_ON_KERNELS) {
        if (runs) hipExtLaunchKernelGGL(k1_emit_runs, dim3((unsigned)(T.n_tiles * (TILE / ET))), dim3(E_THREADS), 0, st, nullptr, ev_emit_end, 0, A, sorted);
        else hipExtLaunchKernelGGL(k1_emit, emit_grid, dim3(256), 0, st, nullptr, ev_emit_end, 0, A, (const Payload *)sorted);
    } else {
        if (runs) hipLaunchKernelGGL(k1_emit_runs, dim3((unsigned)(T.n_tiles * (TILE / ET))), dim3(E_THREADS), 0, st, A, sorted);
        else hipLaunchKernelGGL(k1_emit, emit_grid, dim3(256), 0, st, A, (const Payload *)sorted);
        if (ev_emit_end) HIP_TRY(hipEventRecord(ev_emit_end, st));
    }
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
            hipLaunchKernelGGL(k1_rare, dim3((h.n_rare + 63) / 64), dim3(64), 0, c->stream, A, (const Payload *)c->payload_sorted,
                               (const int64_t *)c->rare_list, (int64_t)h.n_rare);
        }
        if (h.n_big && n > 0) hipLaunchKernelGGL(k1_bigfix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, A, n);
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
        b.cap = b.n_nb = 0; b.k = 0; b.used = false; b.copying = false;
    }
    c->ab_head = c->ab_tail = c->ab_count = 0;
}

static int pinned(void **host, size_t bytes) {
    HIP_TRY(hipHostMalloc(host, std::max<size_t>(bytes, 256), hipHostMallocDefault));
    return 0;
}

static int ensure_async_buf(mc_ctx *c, mc_ctx::AsyncBuf &b, int64_t cap, int k) {
    const DevTable &T = c->T;
    if (!b.ev_done) {
        // events between kernels of this GPU (timing, the side stream's wait for the emit) need no system-scope fence -- without
        // it a record costs the queue ~5 us instead of ~9; the two the host waits for before it reads pinned memory (ev_done,
        // ev_copied) keep the default
        const unsigned dev_flags = hipEventDisableSystemFence;
        for (hipEvent_t *e : {&b.ev_k0_start, &b.ev_k0_end, &b.ev_scan_start, &b.ev_scan_end, &b.ev_emit_end, &b.ev_k2_start, &b.ev_k2_end})
            HIP_TRY(hipEventCreateWithFlags(e, dev_flags));
        for (hipEvent_t *e : {&b.ev_done, &b.ev_copied})
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
    if (b.cap >= cap && b.k == k) return 0;
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
    const size_t pack_bytes = (size_t)cap * (20 + ((size_t)k + 1) * 8 + 1) + 128;       // (every slot mean 64 bits wide at worst, a mask byte per call)
    if (dev_alloc(b.dev_allocs, &b.pack, pack_bytes) || dev_alloc(b.dev_allocs, &b.chunk_cnt, (size_t)PACK_PAD * PACK_WGS)) return -10;
    if (dev_alloc(b.dev_allocs, &b.sorted, (size_t)cap) || dev_alloc(b.dev_allocs, &b.rare, (size_t)cap)) return -10;
    if (b.pack_host) { (void)hipHostFree(b.pack_host); b.pack_host = nullptr; }
    if (pinned((void **)&b.pack_host, pack_bytes)) return -10;
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
    hipLaunchKernelGGL(k1_rare_dev, dim3(64), dim3(64), 0, st, A, (const Payload *)b.sorted, (const int64_t *)b.rare);
    if (b.timed || !MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_k2_start, st));
    if (b.prm.score)
        launch_classifier(c, st, b.O.feats, b.k, b.O.site_seg, T.seg_read, c->qual, b.O.info, (const uint8_t *)nullptr, b.cap, b.O.prob,
                          (const unsigned long long *)&b.cnt->n_records, (const unsigned int *)&b.cnt->overflow);
    if (b.timed || !MC_EVENTS_ON_KERNELS) HIP_TRY(hipEventRecord(b.ev_k2_end, st));
    return 0;
}

// Packing of a pass whose classifier has been enqueued (ev_k2_end recorded): what mc_wait_records_begin copies out.
// count: the chunk counts are not there yet (the emit counts them as it writes the records, except k1_emit_runs)
static int enqueue_pack(mc_ctx *c, mc_ctx::AsyncBuf &b, bool count) {
    hipStream_t s2 = c->side_stream;
    if (count) hipLaunchKernelGGL(k_pack_count, dim3(PACK_WGS), dim3(PACK_THREADS), 0, s2, b.O, (const Counters *)b.cnt, b.k, b.chunk_cnt);
    if (MC_EVENTS_ON_KERNELS)
        hipExtLaunchKernelGGL(k_pack, dim3(PACK_WGS), dim3(PACK_THREADS), 0, s2, nullptr, b.ev_done, 0, b.O, (const Counters *)b.cnt,
                              (const unsigned long long *)b.chunk_cnt, b.pack, b.k, b.close32 ? 1 : 0, b.st_dev);
    else {
        hipLaunchKernelGGL(k_pack, dim3(PACK_WGS), dim3(PACK_THREADS), 0, s2, b.O, (const Counters *)b.cnt,
                           (const unsigned long long *)b.chunk_cnt, b.pack, b.k, b.close32 ? 1 : 0, b.st_dev);
        HIP_TRY(hipEventRecord(b.ev_done, s2));
    }
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
        b.slot = -1;
        c->ab_head = (c->ab_head + 1) % MC_PASSES_IN_FLIGHT;
        c->ab_count += 1;
        return 0;
    }
    const int64_t cap = std::max<int64_t>(guess_capacity(c), c->Omain.capacity);
    if (int rc = ensure_scratch(c, T.n_nb, T.n_tiles)) return rc;
    if (int rc = ensure_records(c, cap, k)) return rc;          // the scratch all passes share (payloads, lists)
    if (int rc = ensure_async_buf(c, b, cap, k)) return rc;
    for (auto &other : c->ab)               // all record sets at once: no (pinned) allocation later, in the middle of a stream
        if (!other.used && (other.cap < cap || other.n_nb < T.n_nb)) { if (int rc = ensure_async_buf(c, other, cap, k)) return rc; }
    // K0 (strand resolve) and K1 (scan, ordering, emit) of a pass on the ctx stream, back to back with the next pass: nothing
    // on the scan's path waits for another queue.  K2 (classifier) and the packing on the side stream, behind the pass's
    // emit: they run beside K0 of the next pass (small latency-bound kernels) and the first microseconds of its scan.
    // Measured on the 10^8-row table (rocprofv3 timelines), passes per second relative to this layout:
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
    if (int rc = enqueue_k1(c, prm, b.K, b.cnt, b.O, st, nullptr, &A, b.sorted, b.rare, b.pass_no, plan, b.ev_emit_end, b.chunk_cnt)) return rc;
    if (int rc = enqueue_k2(c, b, A)) return rc;
    b.close32 = T.n_rows < INT32_MAX;           // (a closing row can be n_rows itself: the next shard's first row)
    if (int rc = enqueue_pack(c, b, A.chunk_cnt == nullptr)) return rc;
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
        if (out_bytes <= COPY_BY_KERNEL_MAX && c->side_stream) {
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
    if (special && getenv("MCALLER_VERBOSE"))
        fprintf(stderr, "mcaller_hip: pass re-run synchronously (overflow %u, irregular %u, big %u, rare %u, records %llu)\n",
                st.overflow, (unsigned)(st.irregular_pass == b.pass_no), st.n_big, st.n_rare, st.n_records);
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
        HIP_TRY(hipEventElapsedTime(&t_scan, b.ev_scan_start, b.ev_emit_end));     // scan + ordering + emit, one span
        t_emit = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t_k2, b.ev_k2_start, b.ev_k2_end));
        c->times[0] = t_k0; c->times[1] = t_scan; c->times[2] = t_emit; c->times[3] = t_k2;
        c->times[4] = t_k0 + t_scan + t_emit + t_k2;
    }
    c->O = b.O;                    // what mc_site_counts reduces: the records of the pass just handed out
    c->last_n = n;
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

#ifdef MC_ER_TRACE
extern "C" int mc_debug_er_trace(unsigned long long *out, int64_t n_words) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_er_trace), (size_t)n_words * 8));
    return 0;
}
#endif

#ifdef MC_K2_TRACE
extern "C" int mc_debug_k2_trace(unsigned long long *out, int64_t n_words) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_trace), (size_t)n_words * 8));
    return 0;
}
#endif

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
    const int64_t ns = c->R.n_sites, n = c->last_n;
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
