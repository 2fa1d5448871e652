// mc_scan.hip: the scan (k1_scan, k_summarize) and the ordering of its payloads (k1_group_scan, k1_list) -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers: mc_dev.h; the map of the
// kernels: mc_stream.hip.
#include "mc_dev.h"

namespace {

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
    if (!gather && l == 0) A.tile_first[tile] = first;
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


}  // namespace

void mc_launch_summarize(const DevTable &T, hipStream_t st) {
    hipLaunchKernelGGL(k_summarize, dim3((unsigned)((T.n_rows / 4 + 256) / 256)), dim3(256), 0, st, T);
}

// one wave per tile; the instance with the small candidate list unless marked positions are dense (a one-base motif)
void mc_launch_scan(const K1Args &A, bool dense, int scan_mode, hipStream_t st) {
    const dim3 grid((unsigned)A.T.n_tiles);
    constexpr int CG_DENSE = CHUNK / 8 + 2;      // (every unit of a chunk; one cut by the boundary of its two blocks is listed twice)
    if (dense) {
        if (scan_mode == SCAN_VALIDATE) hipLaunchKernelGGL((k1_scan<CG_DENSE, SCAN_VALIDATE>), grid, dim3(64), 0, st, A);
        else hipLaunchKernelGGL((k1_scan<CG_DENSE, SCAN_STREAM>), grid, dim3(64), 0, st, A);
    } else {
        if (scan_mode == SCAN_VALIDATE) hipLaunchKernelGGL((k1_scan<64, SCAN_VALIDATE>), grid, dim3(64), 0, st, A);
        else if (scan_mode == SCAN_STREAM) hipLaunchKernelGGL((k1_scan<64, SCAN_STREAM>), grid, dim3(64), 0, st, A);
        else hipLaunchKernelGGL((k1_scan<64, SCAN_SUMMARY>), grid, dim3(64), 0, st, A);
    }
}

void mc_launch_group_scan(const int32_t *cnt, int64_t n, int32_t *local, int64_t *group_sum, hipStream_t st) {
    hipLaunchKernelGGL(k1_group_scan, dim3((unsigned)((n + GROUP - 1) / GROUP)), dim3(GROUP), 0, st, cnt, n, local, group_sum);
}

void mc_launch_list(const K1Args &A, Payload *sorted, int gather, hipStream_t st, hipEvent_t stop) {
    const dim3 grid((unsigned)((A.T.n_tiles * 8 + 255) / 256));
    if (stop) hipExtLaunchKernelGGL(k1_list, grid, dim3(256), 0, st, nullptr, stop, 0, A, sorted, gather);
    else hipLaunchKernelGGL(k1_list, grid, dim3(256), 0, st, A, sorted, gather);
}
