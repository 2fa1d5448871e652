// mc_emit.hip: the window emit (k1_emit; dense references: k1_emit_runs) and the row-by-row kernels behind it (k1_rare, k1_bigfix) -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers: mc_dev.h; the map of the
// kernels: mc_stream.hip.
#include "mc_dev.h"
#include "mc_rows.h"

namespace {

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
static_assert(EG >= MC_MAX_K, "one lane per slot");

// Windows longer than WROWS rows (a handful per 10^8 rows, if any): k1_emit lists them, k1_rare walks them row by row,
// one thread each, after the host has seen the count.

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
#ifdef MC_EM_TRACE      // (variant build: 100 MHz time stamps of the first two rounds of the first wave of 1024 workgroups)
__device__ unsigned long long g_em_trace[1024 * 2 * 8];
#define EM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && em_round < 2) g_em_trace[(blockIdx.x * 2 + em_round) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define EM_STAMP(i) do { } while (0)
#endif
#ifndef MC_EMIT_WAVES
#define MC_EMIT_WAVES 6             // (80 registers, nothing spilled: a reload from scratch waits for every load in flight)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MC_EMIT_WAVES, MC_EMIT_WAVES))) void k1_emit(K1Args A, const Payload *__restrict__ sorted) {
    const DevTable &T = A.T;
    const int lane = threadIdx.x & 63;
    [[maybe_unused]] int em_round = 0;
    EM_STAMP(7);
    if (A.cnt->overflow) return;       // the record buffers were too small: k1_list left payloads unwritten, the pass is repeated
    const int64_t n_rec = min((int64_t)A.cnt->n_records, A.O.capacity);
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
    // (tw: the wave's first lane, the same in all lanes -- scalar registers)
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int64_t tw = blockIdx.x * (int64_t)blockDim.x + wave_in_wg * 64; tw / EG < n_rec; tw += stride) {
    // (what a round can work out again in an instruction is worked out again: kept across the loop these values -- and the
    // constant of the probability column -- are what the compiler spills, and a reload from scratch waits for every load in flight)
    int lane_now = lane;
    uint32_t nan_hi = 0x7ff80000u;
    asm volatile("" : "+v"(lane_now), "+v"(nan_hi));
    const int64_t t = tw + lane_now;
    EM_STAMP(0);
    const int s = lane_now & (EG - 1);
    const int gsh = lane_now & ~(EG - 1);                    // first lane of my group
    const double no_prob = __longlong_as_double((long long)((unsigned long long)nan_hi << 32));
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
    // ---- everything the payload addresses, in ONE round trip: what every window needs of the name block's descriptor (28 of its
    // 64 bytes, the eight lanes of a group the same ones), the segment of the block, and the rows before the window's last row -- lane l of the group looks
    // at rows r-l, r-l-8, r-l-16, r-l-24: four independent loads of the position and of the flag byte, eight consecutive rows
    // per load instruction and group (the columns have FRONT rows of padding in front: no clamping).  No test between the loads
    // and the line that keeps them together: a load behind a test is waited for before the next one is sent (a lane without a
    // window reads what its stand-in payload points at: block 0, row 0) ----
    const NbDesc *dp = A.desc + P.nb;
    static_assert(offsetof(NbDesc, row_begin) == 0 && offsetof(NbDesc, mask_off) == 16 && offsetof(NbDesc, first_delta) == 24 && offsetof(NbDesc, contig_len) == 28,
                  "the descriptor's second 16 bytes: mask_off, first_delta, contig_len");
    int2 d_rb = *reinterpret_cast<const int2 *>(&dp->row_begin);
    int4 d_1 = reinterpret_cast<const int4 *>(dp)[1];
    int32_t d_sdel = dp->seq_delta;
    int32_t seg_of = T.nb_seg_begin[P.nb];
    int pj[4];
    uint32_t fj[4];
    {
        const int32_t *pp = T.pos + (r - s);
        const uint8_t *fp = T.flags + (r - s);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pj[e] = pp[-8 * e];
            fj[e] = fp[-8 * e];
        }
    }
    asm volatile("" : "+v"(d_rb.x), "+v"(d_rb.y), "+v"(d_1.x), "+v"(d_1.y), "+v"(d_1.z), "+v"(d_1.w), "+v"(d_sdel),
                      "+v"(seg_of), "+v"(pj[0]), "+v"(pj[1]), "+v"(pj[2]), "+v"(pj[3]), "+v"(fj[0]), "+v"(fj[1]), "+v"(fj[2]), "+v"(fj[3]));
    EM_STAMP(1);
    const int64_t d_row_begin = ((int64_t)(uint32_t)d_rb.x) | ((int64_t)d_rb.y << 32), d_mask_off = ((int64_t)(uint32_t)d_1.x) | ((int64_t)d_1.y << 32);
    const int64_t d_first = d_1.z < 0 ? -1 : d_row_begin + d_1.z;         // (NbDesc::first())
    const int64_t Lc = d_1.w;
    if (live && !window && s == 0) {            // the one-event '+' window of a palindromic first site row (R5)
        for (int s2 = 0; s2 < k; ++s2) A.O.feats[q * k + s2] = 0.0;
        A.O.wmask[q] = 0;
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = seg_of;
        A.O.close_row[q] = P.close_row;
        A.O.info[q] = MC_I_TOO_MANY | ((!(P.flags & PF_CLOSE_NS) && (dp->xflags & 1)) ? MC_I_MULTI : 0u);
        A.O.prob[q] = no_prob;
    }
    // ---- which of those rows belong to which slot?  A row is in the window iff it is unfiltered, not before the block's first
    // tested row, and its k-mer offset m - pos is one of 0..k-1; positions are non-decreasing in a regular block, so the first
    // unfiltered row with pos < m-k+1 (or the block's start) ends the window.  One window in a hundred is longer than 32 rows:
    // the groups that saw no end look at rows 32..63 in a second step; a window longer than 64 rows goes to the row-by-row kernel
    // (k1_rare).  (Fewer rows looked at = fewer DRAM lines per window: the kernel's time is the number of scattered lines it
    // touches.) ----
    uint32_t W = 0xFFFFFFFFu;                   // my eight rows' slots, four bits each (15: not in the window)
    bool stop_any = false;
    int back = 0;
    // rows r-l-8e, e = E0 .. E0+3 -> their nibbles of W
    auto sort_rows = [&](const int E0) {
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
        back = (int)min(r - max(d_row_begin, d_first), (int64_t)1 << 20) - s;
        sort_rows(0);
    }
    bool covered = ((__ballot(stop_any) >> gsh) & 0xFFull) != 0ull;
    if (__ballot(window && !covered)) {             // (one round in twelve)
        if (window && !covered) {
            const int32_t *pp = T.pos + (r - s);
            const uint8_t *fp = T.flags + (r - s);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pj[e] = pp[-8 * (4 + e)];
                fj[e] = fp[-8 * (4 + e)];
            }
            sort_rows(4);
        }
        covered = ((__ballot(stop_any) >> gsh) & 0xFFull) != 0ull;
    }
    const bool fast = window && covered;
    if (window && !covered && s == 0) A.rare_list[atomicAdd(&A.cnt->n_rare, 1u)] = q;
    EM_STAMP(2);
    // ---- what the info word needs from the reference: context[k], the character after the 'M', picks the sub-model (:197) --
    // the mask word and the base at m + 1 (m - 1 on the reverse strand), one load each behind the descriptor; they travel beside
    // the (event, model) loads below and are looked at when the slot means are done.  All lanes ask (the eight of a group the
    // same addresses); a context that leaves the contig is Python slicing's business (MC_I_EDGE): its lanes ask for position 0 ----
    const bool rev0 = P.flags & PF_REV;
    const bool edge = m - k + 1 < 0 || (int64_t)m + k > Lc || m < 1 || m + 1 >= Lc;
    const int at = edge ? 0 : (rev0 ? m - 1 : m + 1);
    uint32_t ctx_word = ((rev0 ? A.R.mr : A.R.mf) + d_mask_off)[at >> 5];
    // (a reference whose two layouts lie too far apart for the descriptor's 32 bits: the base is fetched at the end, two loads)
    const bool seq_near = d_sdel != NO_SEQ_DELTA;
    uint32_t ctx_base = A.R.seq[(seq_near ? 32 * d_mask_off + d_sdel : (int64_t)0) + at];
    uint32_t info_out = 0u;
    bool info_ctx = false;
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
    // (a window left to the row-by-row walk, in a pass whose copy-out is packed: a call has a row in the packed block and the
    // rows are counted HERE -- will it be a call?  The walk's own count of the empty slots, made now; its slot means travel wide,
    // all k of them)
    if (A.chunk_cnt && window && !covered && s == 0 && !window_too_many(A, P.nb, r, m)) { kept_rec = true; wide_bit = (1u << k) - 1u; }
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
            info_ctx = true;
        }
        if (P.flags & PF_MULTI) info |= MC_I_MULTI;         // the closing row shifted the window (:242-248)
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = seg_of;
        A.O.close_row[q] = P.close_row;
        A.O.prob[q] = no_prob;
        info_out = info;
        kept_rec = !too_many;
    }
    }
    EM_STAMP(3);
    // (the mask word and the base are waited for HERE, by everybody: a use behind a test, and the compiler moves their loads
    // behind the test as well -- behind the slot means instead of beside them)
    asm volatile("" : "+v"(ctx_word), "+v"(ctx_base));
    EM_STAMP(4);
    if (fast && s == 0) {
        if (info_ctx) {
            if (edge) info_out |= MC_I_EDGE;             // the 2k-1 context leaves the contig: Python slicing decides
            else {
                if (!seq_near) ctx_base = A.R.seq[A.R.seq_off[dp->contig] + at];
                const unsigned char ch = ((ctx_word >> (at & 31)) & 1u) ? 'M' : (rev0 ? comp_char((unsigned char)ctx_base) : (unsigned char)ctx_base);
                info_out |= ((uint32_t)ch) << MC_I_NEXT_SHIFT;
            }
        }
        A.O.info[q] = info_out;
    }
    count_wave_for_packing(A.chunk_cnt, n_rec, kept_rec, q, wide_bit, k);       // (the packing's counts, k_pack)
    EM_STAMP(5);
    ++em_round;
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
#define MC_ER_WAVES 7
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
    int16_t end, lb;                // lb: first row that is in a run (-1: before the staged rows; >= the staged rows: none)
    int id, contig, contig_len, stray_q;
    uint32_t xflags;
    int64_t mask_off;
};
static_assert(sizeof(RunBlock) == 32, "RunBlock layout");

// (the barriers of k1_emit_runs order LDS traffic only: __syncthreads() would also wait for every global load in flight -- the
// rows a wave keeps in registers, what a window needs from the reference -- although nobody shares those)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int E_CHUNKS = ER / 64;                          // the staged rows in chunks of 64: a wave owns 5 or 4 consecutive ones
constexpr int E_CPW = 5;
static_assert(ER % 64 == 0 && E_THREADS == 256 && (E_CHUNKS + 3) / 4 <= E_CPW, "the chunks are split over 4 waves, at most E_CPW each");
static_assert(ER < (1 << 12), "s_rrow keeps RUN_* above the row");
constexpr int E_RF_SHIFT = 12;

__global__ __launch_bounds__(E_THREADS) __attribute__((amdgpu_waves_per_eu(MC_ER_WAVES, MC_ER_WAVES))) void k1_emit_runs(K1Args A, Payload *__restrict__ sorted) {
    __shared__ uint8_t s_rid[ET];               // run at or before the row (the piece's rows: where windows end), counted from ...
    __shared__ int16_t s_cb[E_CHUNKS];          // ... the last run that begins before the row's chunk of 64 (-1: none)
    __shared__ int32_t s_dc[ER + 8];            // (event - model) of the rows in runs, run after run
    __shared__ double s_mean[ER];
    __shared__ uint16_t s_rpos[ER];             // the run's position, its low 16 bits: windows look at differences between neighbours (RUN_UNUSABLE: a jump they cannot tell)
    __shared__ uint16_t s_rrow[ER];             // first row of the run (staged index) | RUN_* << 12
    __shared__ uint16_t s_rc0[ER + 2];          // where the run's rows begin in s_dc; one more: where the last run's end
    __shared__ RunBlock s_blk[E_MAXB];
    __shared__ int s_nblk, s_wheads[E_THREADS / 64], s_wins[E_THREADS / 64];
    const DevTable &T = A.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PIECES = TILE / ET;
    const int64_t tile = blockIdx.x / PIECES;
    static_assert(ET == CHUNK && PIECES == 2, "a piece is a chunk of the scan: its windows are the tile's first tile_half, or the rest");
    const int piece = blockIdx.x % PIECES;
    const int64_t s0 = tile * TILE + (int64_t)piece * ET, s1 = min(s0 + (int64_t)ET, T.n_rows);
    if (s0 >= T.n_rows) return;
    const int k = A.k;
    const int64_t h0 = max(s0 - (int64_t)EH, (int64_t)0);
    const int nst = (int)(s1 - h0), i_piece = (int)(s0 - h0);
    ER_STAMP(0);
    // ---- the rows: a wave owns consecutive chunks of 64 (lane = row in the chunk) and keeps them in registers; with them the
    // chunk in front of its first one (which row in a run came last before the wave's rows).  Their addresses need nothing but
    // the block index, so they set out FIRST and all at once, whatever the piece turns out to hold: a lane beyond the staged rows
    // reads the last staged row and drops it (a load behind a test is waited for before the next one is sent) ----
    const int c_lo = (wave * E_CHUNKS + 3) >> 2, c_hi = ((wave + 1) * E_CHUNKS + 3) >> 2;
    int32_t rp[E_CPW], rd[E_CPW];
    uint32_t rfl[E_CPW];
    int2 re[E_CPW];
#pragma unroll
    for (int c = 0; c < E_CPW; ++c) {
        const int64_t j = h0 + min((c_lo + c) * 64 + lane, nst - 1);
        re[c] = T.evmu[j];
        rp[c] = T.pos[j];
        rfl[c] = T.flags[j];
    }
    const int64_t jpre = h0 + min(max((c_lo - 1) * 64 + lane, 0), nst - 1);
    int32_t pre_p = T.pos[jpre];
    uint32_t pre_f = T.flags[jpre];
    // ---- what the piece holds (scalars, one trip), then the name blocks and where this thread's first window lies (one more,
    // beside the rows).  No test stands between the loads and KEEP_TOGETHER: the compiler moves a load behind a branch that can
    // leave (one round trip each, then), and splits a descriptor into the part the first test needs and the rest ----
    const unsigned overflow = A.cnt->overflow;   // the record buffers were too small: k1_list left payloads unwritten, the pass is repeated
    const int t_half = A.tile_half[tile], t_cnt = A.tile_cnt[tile];
    const int64_t n_rec = min((int64_t)A.cnt->n_records, A.O.capacity);
    const int64_t first_rec = A.tile_first[tile];
    const int bfrom = T.tile_nb[(h0 >= tile * TILE || tile == 0) ? tile : tile - 1];
    const int w_lo = piece ? t_half : 0, w_hi = piece ? t_cnt : t_half;
    const bool live = !overflow && w_lo < w_hi;
    // (the tile's payloads are in file order, where the scan left them: the first PT in the tile's own slots, the rest in chunks)
    const int w0 = w_lo + tid;
    const bool have_p0 = live && w0 < w_hi && first_rec + w0 < n_rec;
    const int j0 = have_p0 ? w0 : 0, jc0 = max(j0 - PT, 0);
    long long cb0 = A.tile_chunk[tile * NCHUNK + (jc0 >> A.chunk_shift)];
    // the name blocks that overlap the staged rows: the first wave looks at the 64 blocks from the one the tile before began in
    // (the staged rows begin at most EH rows in front of this tile), all at once; the whole descriptor, 64 bytes in four loads (a
    // lane behind the last block reads the last one and drops it)
    const int b = bfrom + lane;
    if (wave == 0) {
        const int4 *dq = (const int4 *)(A.desc + min(b, T.n_nb - 1));
        int4 q0 = dq[0], q1 = dq[1], q2 = dq[2], q3 = dq[3];
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w),
                          "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w), "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));
        bool over = false, ends_early = false;
        RunBlock rb;
        NbDesc D;
        ((int4 *)&D)[0] = q0; ((int4 *)&D)[1] = q1; ((int4 *)&D)[2] = q2; ((int4 *)&D)[3] = q3;
        if (b < T.n_nb) {
            const NbDesc *dp = &D;
            const int64_t rbeg = dp->row_begin, rend = dp->row_end;
            over = rbeg < s1 && rend > h0;
            ends_early = rend < s1 && b + 1 < T.n_nb;          // (the block behind this one begins before the piece ends)
            if (over) {
                rb.end = (int16_t)min(rend - h0, (int64_t)nst);
                // first row that belongs to a run: the block's first tested row (rows in front of it, and blocks that are not
                // regular, are in no run)
                rb.lb = (int16_t)(dp->mode == MODE_REGULAR ? min(max(max(rbeg, dp->first()) - h0, (int64_t)-1), (int64_t)nst) : (int64_t)nst);
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
    asm volatile("" : "+v"(re[0].x), "+v"(re[0].y), "+v"(rp[0]), "+v"(rfl[0]), "+v"(re[1].x), "+v"(re[1].y), "+v"(rp[1]), "+v"(rfl[1]),
                      "+v"(re[2].x), "+v"(re[2].y), "+v"(rp[2]), "+v"(rfl[2]), "+v"(re[3].x), "+v"(re[3].y), "+v"(rp[3]), "+v"(rfl[3]),
                      "+v"(re[4].x), "+v"(re[4].y), "+v"(rp[4]), "+v"(rfl[4]), "+v"(pre_p), "+v"(pre_f), "+v"(cb0));
    static_assert(E_CPW == 5, "KEEP_TOGETHER lists the rows of five chunks");
    // this thread's first window sets out now and is long there when the run table stands
    Payload P0 = A.payload[j0 >= PT ? cb0 + (jc0 & ((1 << A.chunk_shift) - 1)) : tile * PT + j0];
    if (!live) return;                  // (nothing closes in the piece, or the pass is repeated with more room)
    if (!have_p0) { P0.r = -1; P0.m = 0; P0.flags = 0; P0.nb = 0; P0.close_row = 0; P0.close_pos = 0; }
#pragma unroll
    for (int c = 0; c < E_CPW; ++c) {
        const bool staged = c_lo + c < c_hi && (c_lo + c) * 64 + lane < nst;
        rd[c] = staged ? re[c].x - re[c].y : 0;
        rp[c] = staged ? rp[c] : 0;
        rfl[c] = staged ? rfl[c] : (uint32_t)MC_F_MODEL_N;
    }
    if (!(c_lo > 0 && (c_lo - 1) * 64 + lane < nst)) { pre_p = 0; pre_f = MC_F_MODEL_N; }
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
        uint32_t cutm = 0;              // bit c: the lane's row of chunk c begins a run that windows cannot use: its first rows may lie in front of the staged ones
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
            // (... or that lies too far from the run before it for sixteen bits of position)
#ifdef MC_ER_NO_JUMP_CHECK      // (variant build: the mutant tests/test_gpu_parity.py::test_one_base_motif_with_jumps_in_the_positions must catch)
            if (head && alone && lb < 0) cutm |= 1u << c;
#else
            if (head && (alone ? lb < 0 : (uint32_t)(rp[c] - ppos) >= 30000u)) cutm |= 1u << c;
#endif
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
            if (i < nst && i >= i_piece) s_rid[i - i_piece] = (uint8_t)__popcll(headm[c] & le);
            if (lane == 0) s_cb[c_lo + c] = (int16_t)(hbase - 1);
            if ((inm[c] >> lane) & 1ull) s_dc[at] = rd[c];
            if ((headm[c] >> lane) & 1ull) {
                s_rrow[rid] = (uint16_t)(i | (((cutm >> c) & 1u) ? (RUN_UNUSABLE << E_RF_SHIFT) : 0));
                s_rpos[rid] = (uint16_t)rp[c];
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
            const int R = max((int)s_cb[(int)(P.r - h0) >> 6] + (int)s_rid[(int)(P.r - s0)], 0);
            for (int t = 0; t < k && !rare; ++t) {
                const int Rt = R - t;
                if (Rt < 0) { if (lb < 0) rare = true; break; }         // (the window reaches behind the rows in front)
                const int rr = s_rrow[Rt];
                if ((rr & ((1 << E_RF_SHIFT) - 1)) < max(lb, 0)) break;           // a run of the block before
                const int slot = (int16_t)((uint16_t)m - s_rpos[Rt]);       // m - (the run's position): neighbours lie < 30000 apart
                if (slot > k - 1) break;
                const int rf = rr >> E_RF_SHIFT;
                if (rf & RUN_UNUSABLE) { rare = true; break; }
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
        finish_rare_record(A, sorted, rare_list[i], A.chunk_cnt != nullptr);
    }
}


}  // namespace

int mc_emit_occupancy(void) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k1_emit, 256, 0) != hipSuccess) { (void)hipGetLastError(); occ = 0; }
    return occ;
}

// (`stop` rides on the emit's own dispatch packet: a hipEventRecord behind it is a barrier packet of its own and costs the
// queue 5-9 us)
void mc_launch_emit(const K1Args &A, const Payload *sorted, unsigned grid, hipStream_t st, hipEvent_t stop, hipEvent_t start) {
    if (stop) hipExtLaunchKernelGGL(k1_emit, dim3(grid), dim3(256), 0, st, start, stop, 0, A, sorted);
    else hipLaunchKernelGGL(k1_emit, dim3(grid), dim3(256), 0, st, A, sorted);
}

// (dense references: a workgroup per piece of ET rows, the mean of every position once -- it takes the payloads where the scan
// left them)
void mc_launch_emit_runs(const K1Args &A, Payload *sorted, hipStream_t st, hipEvent_t stop) {
    const dim3 grid((unsigned)(A.T.n_tiles * (TILE / ET)));
    if (stop) hipExtLaunchKernelGGL(k1_emit_runs, grid, dim3(E_THREADS), 0, st, nullptr, stop, 0, A, sorted);
    else hipLaunchKernelGGL(k1_emit_runs, grid, dim3(E_THREADS), 0, st, A, sorted);
}

void mc_launch_rare(const K1Args &A, const Payload *sorted, const int64_t *rare_list, int64_t n_rare, hipStream_t st) {
    hipLaunchKernelGGL(k1_rare, dim3((unsigned)((n_rare + 63) / 64)), dim3(64), 0, st, A, sorted, rare_list, n_rare);
}

void mc_launch_rare_dev(const K1Args &A, const Payload *sorted, const int64_t *rare_list, hipStream_t st) {
    hipLaunchKernelGGL(k1_rare_dev, dim3(64), dim3(64), 0, st, A, sorted, rare_list);
}

void mc_launch_bigfix(const K1Args &A, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k1_bigfix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, n);
}

#ifdef MC_EM_TRACE
extern "C" int mc_debug_em_trace(unsigned long long *out, int64_t n_words) {
    if (n_words > 1024 * 2 * 8) return -12;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_em_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
#endif
#ifdef MC_ER_TRACE
extern "C" int mc_debug_er_trace(unsigned long long *out, int64_t n_words) {
    if (n_words > 1024 * 8) n_words = 1024 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_er_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
#endif
