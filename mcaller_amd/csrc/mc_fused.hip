// mc_fused.hip: k1_fused -- scan, ordering and emit of a pass over a DENSE reference (a one-base motif: a window closes every
// eleven rows) in ONE kernel -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers:
// mc_dev.h; the map of the kernels: mc_stream.hip.  Replaces, for pipelined passes, k1_scan<130> + k1_group_scan + k1_list +
// k1_emit_runs (mc_scan.hip, mc_emit.hip): the reference's flush / shift path, extract_contexts.py:179-256.
//
// Where the pair spends its instructions is the scan's question "is this row the last row of a window?", asked of every ROW
// (57 vector instructions a row) -- but the rows of one position share their site, so only the last row of a RUN of one position can
// be a window's last row, and the run table is what k1_emit_runs builds anyway.  Here the question is asked per run, by the
// workgroup that holds the run table:
//   * a workgroup takes a PIECE of FT rows with FH rows in front, stages positions, flag bytes and (event - model), cuts the
//     unfiltered rows of every regular name block into runs, gives every run its mean (NumPy's pairwise order) -- as
//     k1_emit_runs does -- and, new, its SITE: the first marked position of the run's k-mer (two words of the block's strand
//     mask, one batched trip per thread);
//   * the window of run R is closed by the HEAD of the next run of its block iff that head lies beyond R's site (:179): closing
//     run R + 1 = "closer".  A piece owns the windows whose CLOSING row lies in it -- everything a closer needs lies behind it,
//     in the rows in front, so nothing is looked at beyond the piece, and the order of the closing rows is the order in which the
//     reference flushes (:179-239 run at the closing row);
//   * the closers of a piece are counted per thread over consecutive runs, numbered by one scan across the workgroup, listed in
//     LDS, and handed out one window per thread: record = piece x cap + rank.  FIXED ROOM per piece instead of an ordering across
//     pieces (a look-back serialised on the slowest workgroup, a count pass costs what the scan costs: NOTES.md sections 10, 12):
//     the slots a piece does not fill are HOLES (MC_I_HOLE | MC_I_TOO_MANY), which the classifier kernels, k_site_counts and
//     k_pack_count skip like any record that is not a call, and k_pack compacts away: the host never sees one.  A piece with more
//     windows than room marks the pass (Counters.overflow): it is repeated by the pair, with twice the room next time;
//   * what is not the head of a run closing the run before it -- the first unfiltered row of a name block (it closes the last
//     window of the read before, whatever its position), the head of a block's first run (it closes the one-event '+' window of
//     a palindromic first site row, R5), the end of the shard -- is a handful of SPECIAL closers per table: one thread each,
//     from global memory (closed_by), their windows finished by the row-by-row kernel (k1_rare_dev) like k1_emit's long windows;
//     so is everything the run table cannot answer (a run that begins in front of the staged rows, more than 128 events in a
//     run, the stray event of R5); a piece with more name blocks than the table holds is walked a row per lane;
//   * on a table's first pass the piece's rows are validated as the scan validates them (every row against the row before it
//     in its name block: note_validation), event indices staged with the rows.
#include "mc_dev.h"
#include "mc_rows.h"

namespace {

#ifndef MC_FH
#define MC_FH 64
#endif
#ifndef MC_FD_WAVES
#define MC_FD_WAVES 6
#endif
constexpr int F_THREADS = 256;
constexpr int FR = 4 * F_THREADS;   // rows staged per piece: four consecutive ones per thread
constexpr int FH = MC_FH;           // ... of which in front of the piece (what the piece's first windows reach back into)
constexpr int FT = FR - FH;         // rows per piece
#ifndef MC_F_MAXB
#define MC_F_MAXB 16
#endif
constexpr int F_MAXB = MC_F_MAXB;   // name blocks per staged range
constexpr int F_ROW_BITS = 11;
constexpr uint32_t RF_WIDE = 1, RF_UNUSABLE = 2, RF_ALONE = 4;      // flags of a run, above its first row in s_rrow
constexpr int F_MAXSPEC = 2 * F_MAXB + 1;
constexpr int F_HEAVY = FR / 4;     // runs of several events a wave can list: a wave takes every fourth 64 of the (at most FR) runs
static_assert(FH % 4 == 0 && FH >= 16 && FH < FR / 2, "whole groups of four rows in front of a piece");
static_assert(FR < (1 << F_ROW_BITS), "s_rrow keeps the run's flags above its row");

#ifndef MC_FD_STOP
#define MC_FD_STOP 0    // (variant builds, tools/fused_probe.py: n > 0 ends the kernel behind its n-th phase -- what the phases cost)
#endif
// (what the phases so far left in LDS is read back and stored: a store nobody reads is not a store the compiler has to keep)
#define FD_STOP_AFTER(n) do { if (MC_FD_STOP == (n)) { A.O.feats[q0 * k + tid] = s_mean[tid] + (double)(s_ro[tid] + s_rrow[tid] + s_rpos[tid] + s_rc0[tid] + s_dc[tid] + s_cnt[tid & 15] + s_bfirst[tid & 15]); return; } } while (0)
#ifdef MC_FD_TRACE      // (variant build: 100 MHz time stamps of the phases of 1024 workgroups in the middle of the grid)
__device__ unsigned long long g_fd_trace[1024 * 10];
#define FD_STAMP(i) do { if (tid == 0 && blockIdx.x >= 40000 && blockIdx.x < 41024) g_fd_trace[(blockIdx.x - 40000) * 10 + (i)] = wall_clock64(); } while (0)
#else
#define FD_STAMP(i) do { } while (0)
#endif

struct FBlock {                     // a name block that overlaps the staged rows (staged indices), and what its windows need of it
    int16_t end, lb;                // lb: first row that is in a run (-1: before the staged rows; >= the staged rows: none)
    int16_t begin;                  // the block's first row (-1: before the staged rows)
    uint8_t rev, xflags;
    int id, contig, contig_len, stray_q, seq_delta, seg, extra_mpos;
    uint32_t vf;
    int64_t mask_off;
};
static_assert(sizeof(FBlock) == 48, "FBlock layout");

struct FSpec {                      // a special closer's window: finished by k1_rare_dev (kind 1) or written as it is (kind 2: R5)
    int64_t r, cr;                  // last row of the window, closing row
    int32_t m, nb;
    uint8_t kind, ns;               // (the record goes in front of the window its block's first run closes: s_bfirst)
    uint8_t counts;                 // kind 1: what the window adds to the piece's calls (the walk's prediction, made beside the means: 0 or 1)
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The validation of a first pass (mc_scan.hip has the same two for the scan): flags a row shows that the classification of the
// pass did not rest on go into the table's flags; a block taken for regular marks the pass.
__device__ __forceinline__ void f_note_validation(const K1Args &A, int nb_abs, uint32_t seen) {
    const NbDesc *dp = A.desc + nb_abs;
    if (!(seen & ~dp->vf)) return;
    atomicOr(&A.T.nb_vflags[nb_abs], seen);
    if (dp->mode == MODE_REGULAR) {
        *reinterpret_cast<volatile unsigned int *>(&A.cnt->violation) = 1u;
        *reinterpret_cast<volatile unsigned long long *>(&A.cnt->irregular_pass) = A.pass_no;
    }
}
__device__ __forceinline__ uint32_t f_row_vflags(int p, int x, int prev_p, int prev_x, bool has_pred) {
    uint32_t f = p == 0 ? V_POS0 : 0u;
    if (has_pred) {
        if (p < prev_p) f |= V_POS_DEC;
        f |= x > prev_x ? V_IDX_INC : (x < prev_x ? V_IDX_DEC : V_IDX_EQ);
    }
    return f;
}

// Which window does the unfiltered row c (in name block bc; c == n_rows, bc == n_nb: the first row of the next shard) close?
// The one of the unfiltered row before it (:179): everything from global memory, one thread, rare.  c_pos: the row's position
// (looked at only when the two rows share their block).  kind 0: none; 1: the window of site m whose last row is r (block nb);
// 2: the one-event '+' window of nb's palindromic first site row (R5).
struct Closed { int kind; int64_t r; int m, nb; bool ns; };
// (inlined at its three call sites: a call would take the kernel's arguments by reference -- they would be copied to scratch at
// the kernel's entry and every A.x afterwards would be a scratch load)
__device__ __forceinline__ Closed closed_by(const K1Args &A, int64_t c, int bc, int c_pos) {
    const DevTable &T = A.T;
    Closed out;
    out.kind = 0; out.r = 0; out.m = 0; out.nb = 0; out.ns = false;
    int bb = min(bc, T.n_nb - 1);
    int64_t rr = c - 1;
    while (rr >= 0) {                               // the unfiltered row before c
        while (bb > 0 && T.nb_row_begin[bb] > rr) --bb;
        if (A.desc[bb].filtered) { rr = T.nb_row_begin[bb] - 1; continue; }     // (the read whole)
        if (!(T.flags[rr] & MC_F_MODEL_N)) break;
        --rr;
    }
    if (rr < 0) return out;
    const NbDesc d = A.desc[bb];
    if (d.mode != MODE_REGULAR) return out;         // (an irregular block: the pass is finished by the literal path anyway)
    out.ns = bb != bc;
    out.r = rr; out.nb = bb;
    if (rr == d.extra_row()) {                      // R5: closed by whatever unfiltered row comes next
        out.kind = 2; out.m = d.extra_mpos;
        return out;
    }
    if (rr < max(d.row_begin, d.first())) return out;
    const uint32_t *bits = (d.rev ? A.R.mr : A.R.mf) + d.mask_off;
    const int o = first_m(bits, d.contig_len, T.pos[rr], A.k);
    if (o < 0) return out;
    out.m = T.pos[rr] + o;
    if (!out.ns && c_pos <= out.m) return out;
    out.kind = 1;
    return out;
}

__device__ __forceinline__ void write_extra(const K1Args &A, int64_t q, int m, int seg, int64_t cr, bool multi) {
    for (int s2 = 0; s2 < A.k; ++s2) A.O.feats[q * A.k + s2] = 0.0;
    A.O.wmask[q] = 0;
    A.O.site_pos[q] = m;
    A.O.site_seg[q] = seg;
    A.O.close_row[q] = cr;
    A.O.info[q] = MC_I_TOO_MANY | (multi ? MC_I_MULTI : 0u);
    A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
}

// a window left to the row-by-row kernel: it looks its payload up in the ordered list (k1_rare_dev: nb, r, m)
__device__ __forceinline__ void leave_to_rare(const K1Args &A, Payload *__restrict__ sorted, int64_t q, int64_t r, int m, int nb, int64_t cr) {
    Payload P;
    P.r = r; P.close_row = cr; P.m = m; P.close_pos = 0; P.flags = 0; P.nb = nb;
    sorted[q] = P;
    A.rare_list[atomicAdd(&A.cnt->n_rare, 1u)] = q;
}

// A window left to the row-by-row walk: will its record be a call?  (The walk's own count of the empty slots, made now: a call has
// a row in the packed copy-out and the rows are counted per piece, here.)  -> what it adds to the piece's counts: 1 | k << 16, or 0
__device__ __forceinline__ int rare_counts(const K1Args &A, int nb, int64_t r, int m) {
    if (!A.piece_kw) return 0;
    return window_too_many(A, nb, r, m) ? 0 : (1 | (A.k << 16));
}

// the piece's calls and their wide slot means (what every thread counted of its windows) -> A.piece_kw[piece]; all threads call.
// No barrier: a wave adds what it counted to the sum in LDS, then marks its arrival; the wave that arrives last finds every sum in
// (a wave's LDS operations execute in order) and stores the word.
__device__ __forceinline__ void store_piece_counts(const K1Args &A, int64_t piece_no, int n_win, int mine, int tid, int *s_kw, int *s_arrived) {
    if (!A.piece_kw) return;
    if (__ballot(mine != 0)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    }
    if ((tid & 63) == 0) {
        if (mine) atomicAdd(s_kw, mine);
        if (atomicAdd(s_arrived, 1) == F_THREADS / 64 - 1) {
            const int kw = atomicAdd(s_kw, 0);
            // (one word per piece: records | calls << 9 | wide slot means << 18 -- the side kernel is for rooms below 512 slots, so they fit)
            A.piece_kw[piece_no] = n_win | ((kw & 0xFFFF) << 9) | ((kw >> 16) << 18);
        }
    }
}

// exclusive prefix of `mine` over the workgroup's threads (and the total); s_w: F_THREADS / 64 words of LDS
__device__ __forceinline__ int wg_exclusive_scan(int mine, int lane, int wave, int *s_w, int &total) {
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    lds_barrier();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < F_THREADS / 64; ++w) {
        const int v = s_w[w];
        if (w < wave) off += v;
        total += v;
    }
    return off + incl - mine;
}

template <bool VALIDATE>
__global__ __launch_bounds__(F_THREADS) __attribute__((amdgpu_waves_per_eu(MC_FD_WAVES, MC_FD_WAVES)))
void k1_fused(K1Args A, Payload *__restrict__ sorted, int cap) {
    __shared__ int32_t s_dc[FR + 8];            // (event - model) of the rows in runs, run after run; later: the list of closers
    __shared__ double s_mean[FR];
    __shared__ int32_t s_rpos[FR];              // the run's position
    __shared__ uint16_t s_rrow[FR + 1];         // first row of the run (staged index) | RF_* << F_ROW_BITS
    __shared__ uint16_t s_rc0[FR + 2];          // where the run's rows begin in s_dc; one more: where the last run's end
    __shared__ uint8_t s_ro[FR + 1];            // the run's site: offset of the first 'M' of its k-mer + 1 (0: none) | bit 7: the position behind the site is marked too
    __shared__ FBlock s_blk[F_MAXB];
    __shared__ int16_t s_bfirst[F_MAXB + 1];    // first run of the block (the runs of later blocks behind it); -1 until known
    __shared__ FSpec s_spec[F_MAXSPEC];
    __shared__ __attribute__((aligned(16))) uint8_t s_cnt[(FR + 63) / 64 + 2];    // closed windows of every 64 consecutive runs
    __shared__ uint16_t s_heavy[4 * F_HEAVY];        // per wave: its runs of more than one event
    __shared__ int s_nblk, s_anyspec, s_wheads[F_THREADS / 64], s_wins[F_THREADS / 64], s_scan[F_THREADS / 64];
    __shared__ int s_arrived;                   // waves that have added their counts to s_kw
    __shared__ int s_kw;                        // the piece's calls (records without MC_I_TOO_MANY) | their wide slot means << 16: the packing's counts
    __shared__ unsigned s_over;                 // Counters.overflow as the workgroup's first wave saw it: ONE answer for all waves
    __shared__ int2 s_wlast[F_THREADS / 64];    // a wave's last row in a run: (row, position), row -1: none
    __shared__ int4 s_wfirst[F_THREADS / 64];   // ... its first one: (row, position, first row of its block that is in a run); row -1: none
    const DevTable &T = A.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t piece_no = blockIdx.x;
    const int64_t s0 = piece_no * (int64_t)FT, s1 = min(s0 + (int64_t)FT, T.n_rows);
    if (piece_no == 0 && tid == 0) A.cnt->n_records = (unsigned long long)gridDim.x * (unsigned long long)cap;      // every slot of every piece: holes are records that are not calls
    const int k = A.k;
    const int64_t q0 = piece_no * (int64_t)cap;
    if (s0 >= T.n_rows) {                        // (a piece behind the table: all holes)
        for (int w = tid; w < cap; w += F_THREADS) A.O.info[q0 + w] = MC_I_HOLE | MC_I_TOO_MANY;
        if (tid == 0) { A.piece_cnt[piece_no] = 0; if (A.piece_kw) A.piece_kw[piece_no] = 0; }
        return;
    }
    const int64_t h0 = max(s0 - (int64_t)FH, (int64_t)0);
    const int nst = (int)(s1 - h0), i_piece = (int)(s0 - h0);
    FD_STAMP(0);
    // ---- the rows: a thread owns FOUR CONSECUTIVE rows of the 1024 staged ones and keeps them in registers -- one 16-byte load
    // of the positions, two of the (event, model) pairs, the four flag bytes as a word (the event indices on a first pass): the
    // row before a row is the thread's own three times out of four, so "does this row begin a run" costs a comparison, not a
    // ballot, a count of leading zeros and a shuffle (with a lane per row the heads and the numbering were a third of the kernel).
    // The addresses need nothing but the block index: they set out FIRST (a thread beyond the staged rows reads the last staged
    // group and drops it: a load behind a test is a round trip of its own) ----
    const int i0 = 4 * tid;
    const int64_t jg = h0 + min(i0, (nst - 1) & ~3);            // (whole groups stay inside the columns' padding)
    int4 p4 = *reinterpret_cast<const int4 *>(T.pos + jg);
    int4 ea = *reinterpret_cast<const int4 *>(T.evmu + jg), eb = *reinterpret_cast<const int4 *>(T.evmu + jg + 2);
    uint32_t f4 = *reinterpret_cast<const uint32_t *>(T.flags + jg);
    int4 x4 = make_int4(0, 0, 0, 0);
    int qp0 = 0, qx0 = 0;                        // (a first pass: the row before the thread's first one, for the wave's first lane)
    if (VALIDATE) {
        x4 = *reinterpret_cast<const int4 *>(T.idx + jg);
        const int64_t jq = h0 + min(max(i0 - 1, 0), nst - 1);
        qp0 = T.pos[jq];
        qx0 = T.idx[jq];
    }
    // (a piece ran out of room: the pass is repeated, nobody reads what the others write.  Read by the first wave only and
    // handed to the others through LDS: other workgroups set it while this one runs, and waves that saw different values would
    // part ways in front of the barriers)
    const unsigned overflow = wave == 0 ? *reinterpret_cast<volatile unsigned int *>(&A.cnt->overflow) : 0u;
    const int bfrom = T.tile_nb[h0 / TILE];
    // the name blocks that overlap the staged rows: the first wave looks at the 64 blocks from the one the staged rows' tile
    // began in, all at once; the whole descriptor, 64 bytes in four loads, and the block's segment
    const int b = bfrom + lane;
    if (wave == 0) {
        const int bc = min(b, T.n_nb - 1);
        const int4 *dq = (const int4 *)(A.desc + bc);
        int4 d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
        int seg = T.nb_seg_begin[bc];
        asm volatile("" : "+v"(d0.x), "+v"(d0.y), "+v"(d0.z), "+v"(d0.w), "+v"(d1.x), "+v"(d1.y), "+v"(d1.z), "+v"(d1.w),
                          "+v"(d2.x), "+v"(d2.y), "+v"(d2.z), "+v"(d2.w), "+v"(d3.x), "+v"(d3.y), "+v"(d3.z), "+v"(d3.w), "+v"(seg));
        bool over = false, ends_early = false;
        FBlock fb;
        NbDesc D;
        ((int4 *)&D)[0] = d0; ((int4 *)&D)[1] = d1; ((int4 *)&D)[2] = d2; ((int4 *)&D)[3] = d3;
        if (b < T.n_nb) {
            const int64_t rbeg = D.row_begin, rend = D.row_end;
            over = rbeg < s1 && rend > h0;
            ends_early = rend < s1 && b + 1 < T.n_nb;           // (the block behind this one begins before the piece ends)
            if (over) {
                fb.end = (int16_t)min(rend - h0, (int64_t)nst);
                fb.begin = (int16_t)max(rbeg - h0, (int64_t)-1);
                // first row that belongs to a run: the block's first tested row (rows in front of it, and blocks that are not
                // regular, are in no run)
                fb.lb = (int16_t)(D.mode == MODE_REGULAR ? min(max(max(rbeg, D.first()) - h0, (int64_t)-1), (int64_t)nst) : (int64_t)nst);
                fb.rev = D.rev;
                fb.xflags = (uint8_t)(D.xflags | (D.filtered ? 0x80u : 0u) | (D.mode == MODE_REGULAR ? 0x40u : 0u));
                fb.id = b;
                fb.contig = D.contig;
                fb.contig_len = D.contig_len;
                fb.stray_q = D.stray_q;
                fb.seq_delta = D.seq_delta;
                fb.seg = seg;
                fb.extra_mpos = D.extra_mpos;
                fb.vf = D.vf;
                fb.mask_off = D.mask_off;
            }
        }
        const unsigned long long bal = __ballot(over);
        const int n = __popcll(bal), at = __popcll(bal & ((1ull << lane) - 1ull));
        const bool more_behind = (__ballot(ends_early) >> 63) & 1ull;
        if (over && at < F_MAXB) s_blk[at] = fb;
        if (lane <= F_MAXB) s_bfirst[lane] = -1;
        if (lane < (FR + 63) / 64 + 2) s_cnt[lane] = 0;
        if (lane == 0) { s_nblk = more_behind ? F_MAXB + 1 : n; s_anyspec = 0; s_over = overflow; s_kw = 0; s_arrived = 0; }
    }
    asm volatile("" : "+v"(p4.x), "+v"(p4.y), "+v"(p4.z), "+v"(p4.w), "+v"(ea.x), "+v"(ea.y), "+v"(ea.z), "+v"(ea.w), "+v"(eb.x), "+v"(eb.y), "+v"(eb.z), "+v"(eb.w),
                      "+v"(f4), "+v"(x4.x), "+v"(x4.y), "+v"(x4.z), "+v"(x4.w), "+v"(qp0), "+v"(qx0));
    int rp[4] = {p4.x, p4.y, p4.z, p4.w}, rd[4] = {ea.x - ea.y, ea.z - ea.w, eb.x - eb.y, eb.z - eb.w};
    [[maybe_unused]] const int rx[4] = {x4.x, x4.y, x4.z, x4.w};
    uint32_t nbits = 0;                          // bit e: the thread's row e is filtered ('N' model k-mer) or not staged
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool staged = i0 + e < nst;
        if (!staged || ((f4 >> (8 * e)) & MC_F_MODEL_N)) nbits |= 1u << e;
        if (!staged) { rp[e] = 0; rd[e] = 0; }
    }
    lds_barrier();
    // A first pass goes on even so: its rows have to be VALIDATED by this pass -- the repeat is planned as a later pass over a
    // validated table (TableSlot.passes) and classifies on the flags this pass leaves complete
    if (!VALIDATE && s_over) return;
    FD_STAMP(1);
    FD_STOP_AFTER(1);
    const int nblk = s_nblk;
    const bool usable = nblk <= F_MAXB;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool at_eof = s1 == T.n_rows && A.tail_contig >= 0;      // the first row of the next shard closes this shard's last window (R6, R8)

    if (!usable) {
        // ---- more name blocks than the table holds (reads of a few dozen rows): every row of the piece is looked at as a closer,
        // from global memory; a thread takes four consecutive rows.  Exact, slow, rare ----
        Closed cl[5];
        int mine = 0;
        int bc = bfrom;
#pragma unroll 1
        for (int e = 0; e < 5; ++e) {
            cl[e].kind = 0;
            int64_t c = s0 + 4 * tid + e;
            if (e == 4) { if (!(tid == F_THREADS - 1 && at_eof)) continue; c = T.n_rows; }
            else if (c >= s1) continue;
            int cpos = 0;
            if (c < T.n_rows) {
                while (bc + 1 < T.n_nb && T.nb_row_begin[bc + 1] <= c) ++bc;
                if (VALIDATE) {
                    const bool has_pred = c > T.nb_row_begin[bc];
                    const uint32_t f = f_row_vflags(T.pos[c], T.idx[c], has_pred ? T.pos[c - 1] : 0, has_pred ? T.idx[c - 1] : 0, has_pred);
                    f_note_validation(A, bc, f);
                }
                if (A.desc[bc].filtered || (T.flags[c] & MC_F_MODEL_N)) continue;
                cpos = T.pos[c];
            }
            cl[e] = closed_by(A, c, c < T.n_rows ? bc : T.n_nb, cpos);
            mine += cl[e].kind != 0;
        }
        int n_win;
        int rank = wg_exclusive_scan(mine, lane, wave, s_scan, n_win);
        if (n_win > cap) { if (tid == 0) atomicOr(&A.cnt->overflow, 1u); return; }
        int kw = 0;
#pragma unroll 1
        for (int e = 0; e < 5; ++e) {
            if (!cl[e].kind) continue;
            const int64_t q = q0 + rank++;
            const int64_t c = e == 4 ? T.n_rows : s0 + 4 * tid + e;
            if (cl[e].kind == 2) write_extra(A, q, cl[e].m, T.nb_seg_begin[cl[e].nb], c, !cl[e].ns && A.desc[cl[e].nb].extra_multi());
            else { leave_to_rare(A, sorted, q, cl[e].r, cl[e].m, cl[e].nb, c); kw += rare_counts(A, cl[e].nb, cl[e].r, cl[e].m); }
        }
        for (int w = n_win + tid; w < cap; w += F_THREADS) A.O.info[q0 + w] = MC_I_HOLE | MC_I_TOO_MANY;
        if (tid == 0) A.piece_cnt[piece_no] = n_win;
        store_piece_counts(A, piece_no, n_win, kw, tid, &s_kw, &s_arrived);
        return;
    }

    int n_runs = 0;
    {
        // ---- the thread's rows: which are in runs, which begin one (the row before it in its block that is in a run lies at
        // another position, or there is none); a first pass: every row of the piece against the row before it ----
        int bj = 0;
        while (bj + 1 < nblk && i0 >= s_blk[bj].end) ++bj;
        const int bj0 = bj;
        uint32_t inb = 0, lbneg = 0;             // bit e: row e is in a run / its block's tested rows begin in front of the staged ones
        int lbm[4];
        int lrow = -1, lpos = 0;                 // the thread's last row in a run
        [[maybe_unused]] int qp = 0, qx = 0;
        if (VALIDATE) {
            // the row before the thread's first: the previous lane's last (wave_shr:1 -- a DPP move), the wave's first lane: loaded
            qp = __builtin_amdgcn_update_dpp(0, rp[3], 0x138, 0xF, 0xF, false);
            qx = __builtin_amdgcn_update_dpp(0, rx[3], 0x138, 0xF, 0xF, false);
            if (lane == 0) { qp = qp0; qx = qx0; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            while (bj + 1 < nblk && i >= s_blk[bj].end) ++bj;
            const int lb = s_blk[bj].lb, bend = s_blk[bj].end;
            lbm[e] = max(lb, 0);
            if (lb < 0) lbneg |= 1u << e;
            if (VALIDATE) {
                const int bbeg = s_blk[bj].begin;
                const bool mine = i >= i_piece && i < nst && i < bend && i >= bbeg;
                const uint32_t f = mine ? f_row_vflags(rp[e], rx[e], e ? rp[e ? e - 1 : 0] : qp, e ? rx[e ? e - 1 : 0] : qx, i > bbeg) : 0u;
                if (f & ~s_blk[bj].vf) f_note_validation(A, s_blk[bj].id, f);
            }
            const bool in = !((nbits >> e) & 1u) && i >= lbm[e] && i < bend;
            if (in) { inb |= 1u << e; lrow = i; lpos = rp[e]; }
        }
        // the row in a run before the thread's first one: the last one of the nearest lane below that has any; the wave's lowest
        // lane with a row in a run finds it in the waves before (s_wlast, after the barrier)
        const unsigned long long hasm = __ballot(lrow >= 0), below = hasm & lt;
        const int src = 63 - __clzll(below | 1ull);
        int prow = __shfl(lrow, src), ppos = __shfl(lpos, src);
        const bool wave_first = !below && inb != 0u;    // (this thread holds the wave's first row in a run)
        if (!below) { prow = -1; ppos = 0; }
        // heads, alone (the block's first run of the staged rows), cut (... whose first rows may lie in front of the staged ones)
        uint32_t headb = 0, aloneb = 0;
        int efirst = -1;                         // the thread's first row in a run
        {
            int pr = prow, pp = ppos;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!((inb >> e) & 1u)) continue;
                if (efirst < 0) efirst = e;
                const bool alone = pr < lbm[e];
                if (alone || pp != rp[e]) headb |= 1u << e;
                if (alone) aloneb |= 1u << e;
                pr = i0 + e; pp = rp[e];
            }
        }
        // (the wave's first row in a run: decided below, from what the waves before publish -- not counted yet)
        if (wave_first) { headb &= ~(1u << efirst); aloneb &= ~(1u << efirst); }
        // what a wave publishes: its heads and rows in runs, its last row in a run, its first one with its block's first tested row
        {
            int nh = __popc(headb), ni = __popc(inb);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { nh += __shfl_xor(nh, o); ni += __shfl_xor(ni, o); }
            if (hasm) {
                const int top = 63 - __clzll(hasm), bot = __ffsll((unsigned long long)hasm) - 1;
                const int lr = __shfl(lrow, top), lp = __shfl(lpos, top);
                const int fe = __shfl(efirst, bot);
                const int fr = 4 * bot + fe + 64 * 4 * wave, fpv = __shfl(efirst == 0 ? rp[0] : efirst == 1 ? rp[1] : efirst == 2 ? rp[2] : rp[3], bot);
                const int fl = __shfl(efirst == 0 ? lbm[0] : efirst == 1 ? lbm[1] : efirst == 2 ? lbm[2] : lbm[3], bot);
                if (lane == 0) { s_wlast[wave] = make_int2(lr, lp); s_wfirst[wave] = make_int4(fr, fpv, fl, 0); }
            } else if (lane == 0) { s_wlast[wave] = make_int2(-1, 0); s_wfirst[wave] = make_int4(-1, 0, 0, 0); }
            if (lane == 0) { s_wheads[wave] = nh; s_wins[wave] = ni; }
        }
        lds_barrier();
        FD_STAMP(2);
        FD_STOP_AFTER(2);
        // every wave decides the first row in a run of the waves up to itself (is it a head?  alone?) and adds up what lies before it
        int hbase = 0, ibase = 0, n_in = 0;
        {
            int pr = -1, pp = 0;
#pragma unroll
            for (int w = 0; w < F_THREADS / 64; ++w) {
                const int4 wf = s_wfirst[w];
                const int2 wl = s_wlast[w];
                int first_is_head = 0;
                if (wf.x >= 0) {
                    const bool alone = pr < wf.z;
                    first_is_head = (alone || pp != wf.y) ? 1 : 0;
                    if (w == wave && wave_first) {
                        if (first_is_head) headb |= 1u << efirst;
                        if (alone) aloneb |= 1u << efirst;
                    }
                    pr = wl.x; pp = wl.y;
                }
                const int a = s_wheads[w] + first_is_head, b2 = s_wins[w];
                if (w < wave) { hbase += a; ibase += b2; }
                n_runs += a; n_in += b2;
            }
        }
        // ---- runs numbered in row order; the rows in runs packed run after run: what the lanes below hold, then the thread's own ----
        {
            const int mine = __popc(headb) | (__popc(inb) << 16);
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const int excl = incl - mine;
            hbase += excl & 0xFFFF;
            ibase += excl >> 16;
        }
        bj = bj0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!((inb >> e) & 1u)) continue;
            const int i = i0 + e;
            s_dc[ibase] = rd[e];
            if ((headb >> e) & 1u) {
                const bool alone = (aloneb >> e) & 1u;
                const uint32_t fl = ((alone && ((lbneg >> e) & 1u)) ? RF_UNUSABLE : 0u) | (alone ? RF_ALONE : 0u);
                s_rrow[hbase] = (uint16_t)(i | (fl << F_ROW_BITS));
                s_rpos[hbase] = rp[e];
                s_rc0[hbase] = (uint16_t)ibase;
                if (alone) {                    // (the block's first run)
                    while (bj + 1 < nblk && i >= s_blk[bj].end) ++bj;
                    s_bfirst[bj] = (int16_t)hbase;
                }
                ++hbase;
            }
            ++ibase;
        }
        if (tid == 0) { s_rc0[n_runs] = (uint16_t)n_in; s_rrow[n_runs] = (uint16_t)nst; s_ro[n_runs] = 0; }
        lds_barrier();
        FD_STAMP(3);
        FD_STOP_AFTER(3);
    }
    // ---- the mean of every run (its rows in file order, NumPy's pairwise order: np.mean, :186; values fl(d / 1e4), :286) and its
    // site: the first 'M' of meth_ref[p : p + k] (:176, :270) from two words of the block's strand mask -- the words of all runs
    // of a thread set out before the first mean is begun ----
    // (a thread's runs are R = tid, tid + 256, ...: the words of the run after the current one are on their way while its mean
    // is added up -- a loop, not five copies of it: the instruction cache holds the kernel)
    uint32_t wlo = 0u, whi = 0u;
    // (nineteen pieces in twenty lie inside ONE name block: what a run needs of its block is then the same for every run -- taken
    // once, not looked up per run: the row of the run, the search for its block and four fields of it, twice per run)
    const bool one_block = nblk == 1;
    const int b0_len = s_blk[0].contig_len, b0_rev = s_blk[0].rev;
    const uint32_t *const b0_bits = (b0_rev ? A.R.mr : A.R.mf) + s_blk[0].mask_off;
    auto site_words = [&](int R, uint32_t &lo, uint32_t &hi) {
        lo = hi = 0u;
        if (R < n_runs) {
            int clen = b0_len;
            const uint32_t *bits = b0_bits;
            if (!one_block) {
                const int row = s_rrow[R] & ((1 << F_ROW_BITS) - 1);
                int bj = 0;
                while (bj + 1 < nblk && row >= s_blk[bj].end) ++bj;
                const FBlock &B = s_blk[bj];
                clen = B.contig_len;
                bits = (B.rev ? A.R.mr : A.R.mf) + B.mask_off;
            }
            const int p = s_rpos[R];
            const int pw = ((p >= 0 && p < clen) ? max(p - 1, 0) : 0) >> 5;     // (two zero words lie behind every contig's mask)
            const uint32_t *g = bits + pw;
#ifdef MC_FD_FAKE_WORDS          // (variant build, timing only: what the two words' trip costs the phase)
            lo = (uint32_t)pw * 2654435761u; hi = ~lo; (void)g;
#else
            lo = g[0];
            hi = g[1];
#endif
        }
    };
    site_words(tid, wlo, whi);
    int n_heavy = 0;                            // the wave's runs of several events so far (the same in all lanes)
    // the special closers of the piece: what the first unfiltered row of a name block, the head of its first run, and the end
    // of the shard close -- one thread per block, beside the means of the others
    if (tid <= nblk) {
        const int j = tid;
        FSpec e1, e2;
        e1.kind = e2.kind = 0;
        e1.ns = e2.ns = 0; e1.r = e2.r = e1.cr = e2.cr = 0; e1.m = e2.m = e1.nb = e2.nb = 0; e1.counts = e2.counts = 0;
        if (j < nblk) {
            const FBlock &B = s_blk[j];
            if (!(B.xflags & 0x80u) && B.end > i_piece) {            // (not filtered; overlaps the piece)
                const int fr = s_bfirst[j];
                const int frow = fr >= 0 ? (int)(s_rrow[fr] & ((1 << F_ROW_BITS) - 1)) : nst;
                // the block's first unfiltered row.  (A run of the block that begins in front of the piece: the row lies in front
                // of the piece too, and belongs to the piece before -- the usual case, nothing is loaded for it)
                if (!(frow < i_piece)) {
                    int64_t c = T.nb_row_begin[B.id];
                    const int64_t cend = h0 + B.end;
                    while (c < cend && (T.flags[c] & MC_F_MODEL_N)) ++c;
                    if (c >= s0 && c < cend && c < s1) {
                        const Closed cl = closed_by(A, c, B.id, 0);
                        if (cl.kind) { e1.kind = (uint8_t)cl.kind; e1.r = cl.r; e1.cr = c; e1.m = cl.m; e1.nb = cl.nb; e1.ns = 1; }
                    } else if (c < s0 && B.lb < 0 && fr >= 0) {
                        // The block's tested rows begin in front of the staged rows and nothing but 'N' rows of it is staged in front
                        // of its first run (a read across a gap of the model: FH and more filtered rows in a row): that run's head
                        // closes the window of the unfiltered row before the gap (:179 skips the 'N' rows), which the run table
                        // does not hold.  (The '+' window of R5 in that place is e2's, below)
                        const int64_t ch = h0 + frow;
                        const Closed cl = closed_by(A, ch, B.id, s_rpos[fr]);
                        if (cl.kind == 1) { e1.kind = 1; e1.r = cl.r; e1.cr = ch; e1.m = cl.m; e1.nb = cl.nb; e1.ns = 0; }
                    }
                }
                // the head of its first run closes the '+' window of a palindromic first site row (R5; that row is first - 1):
                // if nothing but 'N' rows lies between the two
                if ((B.xflags & 2u) && (B.xflags & 0x40u) && fr >= 0 && frow >= i_piece) {
                    const NbDesc *dp = A.desc + B.id;
                    const int64_t xr = dp->row_begin + dp->first_delta - 1;
                    int64_t r = h0 + frow - 1;
                    while (r > xr && (T.flags[r] & MC_F_MODEL_N)) --r;
                    if (r == xr) { e2.kind = 2; e2.r = xr; e2.cr = h0 + frow; e2.m = B.extra_mpos; e2.nb = B.id; e2.ns = 0; }
                }
            }
        } else if (at_eof) {
            const Closed cl = closed_by(A, T.n_rows, T.n_nb, 0);
            if (cl.kind) { e1.kind = (uint8_t)cl.kind; e1.r = cl.r; e1.cr = T.n_rows; e1.m = cl.m; e1.nb = cl.nb; e1.ns = 1; }
        }
        // (will the window the walk finishes be a call?  Asked HERE, beside the others' means: in the windows phase the walk's round
        // trips would be the workgroup's)
        if (e1.kind == 1) e1.counts = rare_counts(A, e1.nb, e1.r, e1.m) ? 1 : 0;
        s_spec[2 * j] = e1;
        if (j < nblk) s_spec[2 * j + 1] = e2;
        if (e1.kind | e2.kind) s_anyspec = 1;
    }
#pragma unroll 1
    for (int R = tid; R < n_runs; R += F_THREADS) {
        uint32_t nlo, nhi;
        site_words(R + F_THREADS, nlo, nhi);
        // (every second run is ONE event: its mean is fl(d / 1e4), narrow by construction -- done here; the runs of several events are
        // listed per wave and taken densely below: their sums, the division and the narrow test are most of what a run costs, and
        // a wave would go through them for the half of its lanes that have none)
#ifdef MC_FD_NO_MEAN          // (variant build, timing only)
        const int c0 = 0, n = 1;
#else
        const int c0 = s_rc0[R], n = (int)s_rc0[R + 1] - c0;
        if (n == 1) s_mean[R] = div1e4(s_dc[c0]);
#endif
        {
            const unsigned long long hm = __ballot(n > 1);
            if (n > 1) s_heavy[wave * F_HEAVY + n_heavy + __popcll(hm & lt)] = (uint16_t)R;
            n_heavy += __popcll(hm);
        }
        // the site
        int L = b0_len, brev = b0_rev;
        if (!one_block) {
            const int row = s_rrow[R] & ((1 << F_ROW_BITS) - 1);
            int bj = 0;
            while (bj + 1 < nblk && row >= s_blk[bj].end) ++bj;
            L = s_blk[bj].contig_len; brev = s_blk[bj].rev;
        }
        const int p = s_rpos[R];
        uint32_t code = 0;
#ifdef MC_FD_NO_SITE          // (variant build, timing only)
        code = (uint32_t)(p & 7); (void)L; (void)brev;
        if (false) {
#else
        if (p >= 0 && p < L) {
#endif
            const int sh = p - ((max(p - 1, 0) >> 5) << 5);                 // bit of position p in whi:wlo (0 .. 32)
            const uint64_t W = (((uint64_t)whi << 32) | wlo) >> sh;
            const uint32_t bits = (uint32_t)W & ((1u << k) - 1u);
            if (bits) {
                const int o = __ffs(bits) - 1;
                // the position behind the site in the read's direction: m + 1, on the reverse strand m - 1 (context[k], :197)
                const int at_bit = brev ? sh + o - 1 : sh + o + 1;
                const uint32_t next = at_bit >= 0 ? (uint32_t)(((((uint64_t)whi << 32) | wlo) >> at_bit) & 1ull) : 0u;
                code = (uint32_t)(o + 1) | (next << 7);
            }
        }
        // is the run's window closed in this piece?  By the head of the next run, if that is a run of the same block (it has a
        // run before it), lies in the piece, and beyond the site (:179)
        bool closed = false;
        if ((code & 15u) && R + 1 < n_runs) {
            const uint32_t r1 = s_rrow[R + 1];
            closed = !((r1 >> F_ROW_BITS) & RF_ALONE) && (int)(r1 & ((1 << F_ROW_BITS) - 1)) >= i_piece && s_rpos[R + 1] - p > (int)(code & 15u) - 1;
        }
        s_ro[R] = (uint8_t)(code | (closed ? 0x40u : 0u));
        // (the closed windows of the wave's 64 consecutive runs: one count per wave and turn of the loop)
        const unsigned long long cm = __ballot(closed);
        if (lane == 0) s_cnt[R >> 6] = (uint8_t)__popcll(cm);
        wlo = nlo; whi = nhi;
    }
    n_heavy = __shfl(n_heavy, 0);               // (the wave's first lane took every turn of the loop)
    // ---- the runs of several events, the wave's own, densely: the mean in NumPy's pairwise order (np.mean, :186; values fl(d / 1e4),
    // :286), and whether it travels as an integer ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // (a wave's LDS operations execute in order: its own list)
#pragma unroll 1
    for (int i = lane; i < n_heavy; i += 64) {
        const int R = s_heavy[wave * F_HEAVY + i];
        double mean = 0.0;
        uint32_t rf = 0;
        const int c0 = s_rc0[R], n = (int)s_rc0[R + 1] - c0;
        if (n > 128) { rf = RF_UNUSABLE; mean = 0.0; }          // NumPy's pairwise recursion proper: the row-by-row kernel
        else if (n >= 8) {
            const int n8 = n - (n % 8);
            double r[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) r[v] = 0.0;
            for (int j = 0; j < n8; j += 8) {
#pragma unroll
                for (int v = 0; v < 8; ++v) r[v] += div1e4(s_dc[c0 + j + v]);
            }
            double acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (int j = n8; j < n; ++j) acc += div1e4(s_dc[c0 + j]);
            mean = (0.0 + acc) / (double)n;
        } else {
            const int d0 = s_dc[c0], d1 = s_dc[c0 + 1], d2 = s_dc[c0 + 2], d3 = s_dc[c0 + 3];     // (there is room behind the last row)
            double acc = (-0.0 + div1e4(d0)) + div1e4(d1);
            if (n > 2) acc += div1e4(d2);
            if (n > 3) acc += div1e4(d3);
            for (int j = 4; j < n; ++j) acc += div1e4(s_dc[c0 + j]);
            mean = (0.0 + acc) / (double)n;
        }
        if (n <= 128) {
            int32_t as_int;
            if (!slot_is_narrow(mean, &as_int)) rf |= RF_WIDE;
        }
        s_mean[R] = mean;
        if (rf) s_rrow[R] |= (uint16_t)(rf << F_ROW_BITS);
    }
    FD_STAMP(4);
    lds_barrier();
    FD_STAMP(5);
    FD_STOP_AFTER(4);
    // ---- the closers of the piece, in row order (the order in which the reference flushes): every 64 consecutive runs have
    // counted their closed windows; a wave numbers its runs' windows with one ballot per turn and lists the closing runs (the
    // events' room is free now) ----
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_dc);
    int n_win = 0;
    {
        int before = 0;                             // closed windows of the runs in front of this wave's 64 of the turn
        const int n64 = (n_runs + 63) >> 6;
        // (the counts of all groups in one trip: a read per turn of the loop is an LDS round trip per turn)
        static_assert((FR + 63) / 64 <= 16, "sixteen counts: four words");
        const uint4 cw = *reinterpret_cast<const uint4 *>(s_cnt);
        for (int g = 0; g < n64; ++g) {
            const uint32_t w4 = g < 4 ? cw.x : g < 8 ? cw.y : g < 12 ? cw.z : cw.w;
            const int c = (int)((w4 >> (8 * (g & 3))) & 0xFFu);
            if ((g & 3) == wave) {
                const int R = g * 64 + lane;
                const bool closed = R < n_runs && (s_ro[R] & 0x40u);
                const unsigned long long cm = __ballot(closed);
                if (closed) s_list[before + __popcll(cm & lt)] = (uint16_t)(R + 1);
            }
            before += c;
        }
        n_win = before;
    }
    // What the special closers close goes in front of the window their block's first run would close (a block without a run:
    // the next block's): merged into the list by one thread, from the back -- one piece in twenty has any
    if (s_anyspec) {
        lds_barrier();
        if (tid == 0) {
            int next = n_runs;
            for (int j = nblk - 1; j >= 0; --j) {
                if (s_bfirst[j] < 0) s_bfirst[j] = (int16_t)next;
                next = s_bfirst[j];
            }
            s_bfirst[nblk] = (int16_t)n_runs;
            int n_s = 0;
            for (int e = 0; e < 2 * nblk + 1; ++e) n_s += s_spec[e].kind ? 1 : 0;
            int i = n_win - 1, out = n_win + n_s - 1;
            for (int e = 2 * nblk; e >= 0; --e) {
                if (!s_spec[e].kind) continue;
                const int at = s_bfirst[e >> 1];
                while (i >= 0 && (int)s_list[i] >= at) s_list[out--] = s_list[i--];
                s_list[out--] = (uint16_t)(0x8000u | (unsigned)e);
            }
            s_scan[0] = n_win + n_s;
        }
        lds_barrier();
        n_win = s_scan[0];
    }
    if (n_win > cap) { if (tid == 0) atomicOr(&A.cnt->overflow, 1u); return; }
    lds_barrier();
    FD_STAMP(6);
    FD_STOP_AFTER(5);
    // ---- a window per thread ----
    const uint32_t kbits = (1u << k) - 1u;
    const FBlock B0w = s_blk[0];
    int kw = 0;                                     // the thread's calls | their wide slot means << 16
    for (int w = tid; w < n_win; w += F_THREADS) {
        const int64_t q = q0 + w;
        const unsigned ent = s_list[w];
        if (ent & 0x8000u) {
            const FSpec &S = s_spec[ent & 0x7FFFu];
            if (S.kind == 2) write_extra(A, q, S.m, T.nb_seg_begin[S.nb], S.cr, !S.ns && A.desc[S.nb].extra_multi());
            else { leave_to_rare(A, sorted, q, S.r, S.m, S.nb, S.cr); kw += S.counts ? (1 | (k << 16)) : 0; }
            continue;
        }
        const int Rc = (int)ent, R = Rc - 1;
        const int hrow = s_rrow[Rc] & ((1 << F_ROW_BITS) - 1);
        FBlock B = B0w;                             // (one name block in the piece: its fields were taken once)
        if (!one_block) {
            int bj = 0;
            while (bj + 1 < nblk && hrow >= s_blk[bj].end) ++bj;
            B = s_blk[bj];
        }
        const uint32_t code = s_ro[R];
        const int o = (int)(code & 15u) - 1, m = s_rpos[R] + o;
        const bool rev = B.rev;
        // context[k], the character behind the 'M' (:197): marked too, or the base (complemented on the reverse strand)
        const int64_t L = B.contig_len;
        const bool edge = m - k + 1 < 0 || (int64_t)m + k > L || m < 1 || m + 1 >= L;
        const int at = edge ? 0 : (rev ? m - 1 : m + 1);
        const int64_t seq_base = B.seq_delta != NO_SEQ_DELTA ? 32 * B.mask_off + B.seq_delta : A.R.seq_off[B.contig];
#ifdef MC_FD_FAKE_CHAR            // (variant build, timing only: what the base's trip costs the phase)
        uint32_t base_ch = (uint32_t)(seq_base + at) & 0x5Fu;
#else
        uint32_t base_ch = A.R.seq[seq_base + at];
#endif
        bool rare = (B.stray_q != NO_STRAY) && m - B.stray_q >= 0 && m - B.stray_q < k;        // (the stray event of R5 is first in its slot)
        uint32_t have = 0, wide = 0;
        const int lb = B.lb;
        for (int t = 0; t < k && !rare; ++t) {
            const int Rt = R - t;
            if (Rt < 0) { if (lb < 0) rare = true; break; }         // (the window reaches behind the rows in front)
            const uint32_t rr = s_rrow[Rt];
            if ((int)(rr & ((1 << F_ROW_BITS) - 1)) < max(lb, 0)) break;            // a run of the block before
            const int slot = m - s_rpos[Rt];
            if (slot > k - 1) break;
            const uint32_t rf = rr >> F_ROW_BITS;
            if (rf & RF_UNUSABLE) { rare = true; break; }
            if (slot < 0) continue;
#ifdef MC_FD_NO_FEATS             // (variant build, timing only: the slot means' stores)
            if (s_mean[Rt] == 12345.678) A.O.feats[q * k + (rev ? slot : k - 1 - slot)] = s_mean[Rt];
#else
            A.O.feats[q * k + (rev ? slot : k - 1 - slot)] = s_mean[Rt];            // :187-188 (a window that turns out rare is written again)
#endif
            have |= 1u << slot;
            if (rf & RF_WIDE) wide |= 1u << slot;
        }
        const int64_t cr = h0 + hrow;
        if (rare) {                                 // (the row-by-row kernel: the window's last row is the unfiltered row before the closer)
            int64_t r = cr - 1;
            while (r > 0 && (T.flags[r] & MC_F_MODEL_N)) --r;
            leave_to_rare(A, sorted, q, r, m, B.id, cr);
            kw += rare_counts(A, B.id, r, m);
            continue;
        }
        const uint32_t empties = ~have & kbits;
        const bool too_many = __popc(empties) > A.skip_thresh;
        uint32_t info = rev ? MC_I_REV : 0u, wmask = 0;
        for (uint32_t z = too_many ? kbits : empties; z; z &= z - 1u) {
            const int s = __ffs(z) - 1;
            A.O.feats[q * k + (rev ? s : k - 1 - s)] = 0.0;
        }
        if (too_many) info |= MC_I_TOO_MANY;
        else {
            kw += 1 + (__popc(wide) << 16);
            wmask = rev ? wide : __brev(wide) >> (32 - k);
            info |= rev ? empties : __brev(empties) >> (32 - k);   // feature dst came from an empty slot (:186)
            if (edge) info |= MC_I_EDGE;                           // the 2k-1 context leaves the contig: Python slicing decides
            else {
                const unsigned char ch = (code & 0x80u) ? 'M' : (rev ? comp_char((unsigned char)base_ch) : (unsigned char)base_ch);
                info |= ((uint32_t)ch) << MC_I_NEXT_SHIFT;
            }
        }
        // the closing row shifts the window when it continues the chain with kmer[0] != 'M' (:242-248)
        if (s_rpos[Rc] <= m + A.skip_thresh + 1 && (s_ro[Rc] & 15u) > 1u) info |= MC_I_MULTI;
        A.O.wmask[q] = (uint8_t)wmask;
        A.O.site_pos[q] = m;
        A.O.site_seg[q] = B.seg;
        A.O.close_row[q] = cr;
        A.O.info[q] = info;
        A.O.prob[q] = __longlong_as_double(0x7ff8000000000000LL);
    }
    FD_STAMP(7);
    if (tid == 0) A.piece_cnt[piece_no] = n_win;                 // (what the classifier makes its stretches of)
    for (int w = n_win + tid; w < cap; w += F_THREADS) A.O.info[q0 + w] = MC_I_HOLE | MC_I_TOO_MANY;
    store_piece_counts(A, piece_no, n_win, kw, tid, &s_kw, &s_arrived);
    FD_STAMP(8);
}

}  // namespace

// room per piece for a reference with `density` marked positions per position and strand: what guess_capacity() assumes per
// row (a window closes about once per marked site a read covers, ~0.52 positions per row), twice over, in whole sixteens
int mc_fused_room(double density) {
    const int want = (int)(FT * density * 0.52 * 2.0) + 48;
    return std::min(FT + 2 * F_MAXB + 2, (want + 15) & ~15);
}
int mc_fused_room_max(void) { return FT + 2 * F_MAXB + 2; }
int64_t mc_fused_pieces(const DevTable &T) { return (T.n_rows + FT - 1) / FT; }

void mc_launch_fused(const K1Args &A, Payload *sorted, int cap, bool validate, hipStream_t st, hipEvent_t stop) {
    const dim3 grid((unsigned)mc_fused_pieces(A.T));
    if (validate) {
        if (stop) hipExtLaunchKernelGGL(k1_fused<true>, grid, dim3(F_THREADS), 0, st, nullptr, stop, 0, A, sorted, cap);
        else hipLaunchKernelGGL(k1_fused<true>, grid, dim3(F_THREADS), 0, st, A, sorted, cap);
    } else {
        if (stop) hipExtLaunchKernelGGL(k1_fused<false>, grid, dim3(F_THREADS), 0, st, nullptr, stop, 0, A, sorted, cap);
        else hipLaunchKernelGGL(k1_fused<false>, grid, dim3(F_THREADS), 0, st, A, sorted, cap);
    }
}

#ifdef MC_FD_TRACE
extern "C" int mc_debug_fd_trace(unsigned long long *out, int64_t n_words) {
    if (n_words > 1024 * 10) n_words = 1024 * 10;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fd_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
#endif
