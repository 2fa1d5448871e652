// mc_literal.hip: the literal path -- irregular name blocks, the reference's loop row by row (k_literal, k_merge) -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers: mc_dev.h; the map of the
// kernels: mc_stream.hip.
#include "mc_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// The literal path: name blocks the window rule does not cover (same read name in several blocks, positions going
// backwards, strand changes inside a read, a site at contig position 0, reads spanning contigs).  Runs of such blocks
// are executed row by row exactly as the reference's loop does (extract_contexts.py:147-291), one GPU thread per run;
// they are rare, so this path is written for exactness, not speed.  Its records are merged with the fast path's by
// closing row (= flush order).
// ---------------------------------------------------------------------------------------------------
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


}  // namespace

void mc_launch_literal(const LitArgs &LA, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL(k_literal, dim3(grid), dim3(64), 0, st, LA);
}

void mc_launch_merge(const DevRecords &O, int64_t n_o, const DevRecords &L, int64_t n_l, const DevRecords &M, int k, hipStream_t st) {
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((n_o + n_l + 255) / 256)), dim3(256), 0, st, O, n_o, L, n_l, M, k);
}
