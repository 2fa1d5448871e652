// mc_rows.h -- the row-by-row walk of ONE window from global memory (extract_contexts.py:179-239 for a single flush): what the
// fast kernels leave to "rare" (windows longer than k1_emit looks back, what k1_fused's run table cannot answer, its special
// closers, slots of more than 128 events).  Shared by mc_emit.hip (k1_rare, k1_rare_dev, k1_bigfix), mc_fused.hip (the
// prediction below) and mc_classify.hip (the side stream's kernel finishes the rare windows of its own records).
#ifndef MC_ROWS_H
#define MC_ROWS_H
#include "mc_dev.h"

namespace {

// The rows of the window of site m whose last row is r (name block descriptor d): events per slot, eight bits each (big: a
// slot holds more than 128 -- NumPy's recursion proper), the window's first row, the slot of the block's stray event (R5; -1:
// none in this window).  ONE walk over positions and flag bytes; what emit_record builds the record from and what
// window_too_many() predicts from: the two agree by construction.
struct WindowRows { unsigned long long cnt8; int64_t ws; int stray_slot; bool big; };
__device__ __forceinline__ WindowRows window_rows(const DevTable &T, const NbDesc &d, int64_t r, int m, int k) {
    WindowRows W;
    W.cnt8 = 0; W.big = false; W.ws = r; W.stray_slot = -1;
    // back to the first row at position >= m-k+1 (never before d.first() / the block start).  Eight rows at a time, their sixteen
    // loads side by side: a row at a time is a round trip per row, and the thread's workgroup (k1_fused: the piece; the side kernel:
    // the stretch) waits for this walk
    const int64_t lb = max(d.row_begin, d.first());
    bool done = false;
    for (int64_t top = r; top >= lb && !done; top -= 8) {
        int p8[8];
        uint32_t f8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t rr = max(top - e, lb);
            p8[e] = T.pos[rr];
            f8[e] = T.flags[rr];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t rr = top - e;
            if (done || rr < lb) continue;
            if (f8[e] & MC_F_MODEL_N) continue;
            const int p = p8[e];
            if (p < m - k + 1) { done = true; continue; }
            if (p > m) continue;        // (cannot happen in a regular block; a block taken for regular on its first rows may not be)
            const int sh = 8 * (m - p);
            if (((W.cnt8 >> sh) & 0xFFull) >= 128ull) W.big = true;
            else W.cnt8 += 1ull << sh;
            W.ws = rr;
        }
    }
    // the stray event of a palindromic first site row: first in the slot of its pseudo-position
    if (d.stray_q != NO_STRAY) {
        const int sq = m - d.stray_q;
        if (sq >= 0 && sq < k) {
            W.stray_slot = sq;
            if (((W.cnt8 >> (8 * sq)) & 0xFFull) >= 128ull) W.big = true;
            else W.cnt8 += 1ull << (8 * sq);
        }
    }
    return W;
}
__device__ __forceinline__ int window_skips(const WindowRows &W, int k) {
    int nskip = 0;
    for (int s = 0; s < k; ++s) nskip += (((W.cnt8 >> (8 * s)) & 0xFFull) == 0ull);
    return nskip;
}
// Will the record emit_record() writes for this window carry MC_I_TOO_MANY (:239)?  Asked where a window is LEFT to the
// row-by-row walk by a pass whose copy-out is packed: a call has a row in the packed block, and the rows are counted while the
// records are written (k1_emit per packing chunk, k1_fused per piece) -- before the walk has run.
// (inlined: a call would take the kernel's arguments by reference -- copied to scratch at the kernel's entry, every A.x a scratch load)
__device__ __forceinline__ bool window_too_many(const K1Args &A, int nb_abs, int64_t r, int m) {
    const NbDesc d = A.desc[nb_abs];
    return window_skips(window_rows(A.T, d, r, m, A.k), A.k) > A.skip_thresh;
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

    // ---- the window's rows: per-slot event counts packed 8 bits each (a slot with > 128 events goes to k1_bigfix) ----
    const WindowRows WR = window_rows(T, d, r, m, k);
    const unsigned long long cnt8 = WR.cnt8;
    const bool big = WR.big;
    const int64_t ws = WR.ws;
    const int stray_slot = WR.stray_slot;
    const int nskip = window_skips(WR, k);

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


// Records with a slot of more than 128 events (NumPy's pairwise recursion proper): recomputed here, one
// thread per such record, so that k1_emit carries neither the stack nor the registers for it.
__device__ __noinline__ double big_pairwise(RowSrc &S, int64_t &cur, int64_t n) {
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


// A window a pipelined pass left to the row-by-row walk (record q of its ordered payloads), finished by one thread -- including the
// full pairwise recursion if a slot turns out to hold more than 128 events.  rows_counted: the pass's copy-out is packed and the
// record's row was counted where the window was left to this walk (k1_emit per packing chunk, k1_fused per piece), all k slot means
// wide; otherwise the records are counted afterwards (k_pack_count) and 0xFF says "look".
// WITH_BIG = false: without the recursion (its stack is 1.4 KB of scratch per lane, for every wave of the calling kernel) -> true if
// the record needs it (MC_I_BIG is set: the caller has the pass repeated).
template <bool WITH_BIG = true>
__device__ __forceinline__ bool finish_rare_record(const K1Args &A, const Payload *__restrict__ sorted, int64_t q, bool rows_counted) {
    const Payload P = sorted[q];
    const NbDesc d = A.desc[P.nb];
    RowSrc S{A.T.pos, A.T.evmu, A.T.flags, false, 0.0};
    emit_record(A, S, d, P.nb, P.r, P.m, q);
    if constexpr (WITH_BIG) bigfix_record(A, q);
    const uint32_t info = A.O.info[q];
    if (rows_counted && !(info & MC_I_TOO_MANY)) A.O.wmask[q] = (uint8_t)((1u << A.k) - 1u);
    return !WITH_BIG && (info & MC_I_BIG);
}

}  // namespace

#endif  // MC_ROWS_H
