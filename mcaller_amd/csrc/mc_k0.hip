// mc_k0.hip: strand resolve (k0_first_site with the name-block templates, k0_classify, k0_extend) -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers: mc_dev.h; the map of the
// kernels: mc_stream.hip.
#include "mc_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// K0: strand resolve
// ---------------------------------------------------------------------------------------------------
// One wave per name block.  Under the reference's rule for a read it has not seen a site row of yet
// (`read_name != last_read`, :161-174) each unfiltered row is tested on the strand `rev = (col3 != col10)`;
// the first row that holds an 'M' in its k-mer becomes the block's first site row f0.
// (the fields of a descriptor that do not depend on the pass -- rows, contig, mask offset, read -- are prepared once per
// table/reference by nb_template(), so that a pass reads one 64-byte line per block instead of walking five tables)
// The pass-independent fields of block b's descriptor, in two steps: what the block's own tables say (rows, read, segments: one
// load each), then what hangs on its first segment's contig (two more dependent loads) -- the caller sends the block's first
// rows on their way in between
__device__ __forceinline__ void nb_template_rows(const DevTable &T, int b, NbDesc &d, int &seg0) {
    d.row_begin = T.nb_row_begin[b];
    d.row_end = T.nb_row_begin[b + 1];
    d.read = T.nb_read[b];
    seg0 = T.nb_seg_begin[b];
    d.vf = (uint32_t)(T.nb_seg_begin[b + 1] - seg0);       // segments (contigs) of the block
    d.first_delta = -1;
    d.stray_q = NO_STRAY;
    d.stray_d = 0;
    d.extra_mpos = 0;
    d.mode = MODE_NONE;
    d.rev = 0;
    d.filtered = 0;
    d.xflags = 0;
}
__device__ __forceinline__ void nb_template_contig(const DevTable &T, const DevRef &R, int seg0, NbDesc &d) {
    d.contig = T.seg_contig[seg0];
    d.mask_off = R.word_off[d.contig];
    d.contig_len = (int32_t)R.contig_len[d.contig];
    const int64_t sd = R.seq_off[d.contig] - 32 * d.mask_off;
    d.seq_delta = (sd > (int64_t)INT32_MIN && sd <= (int64_t)INT32_MAX) ? (int32_t)sd : NO_SEQ_DELTA;
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
                              int skip_thresh, unsigned long long pass_no, int hyp, int make_tmpl) {
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
    // make_tmpl: the table (or the reference) is new -- the block's template is made here and kept for the passes to come: a
    // launch of its own costs the queue 5 us, here it is two loads in front of the mask words, beside the block's first rows
    NbDesc d;
    int tm_seg0 = 0;
    if (make_tmpl) nb_template_rows(T, b, d, tm_seg0);
    else d = T.nb_tmpl[b];
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
    // (the usual block -- one segment, one wave: the first round's rows set out now, in front of the template's two loads)
#ifndef MC_K0_FS
#define MC_K0_FS 8
#endif
    constexpr int FS = MC_K0_FS;         // stripes of 64 rows per round
    uint32_t fl[FS], fl_next[FS];
    int ps[FS], ps_next[FS];
    const bool rows_out = W == 1 && n_seg == 1 && d.row_begin < d.row_end;
    if (rows_out) first_site_rows<FS>(T, d.row_begin, d.row_end, lane, fl, ps);
    if (make_tmpl) {
        nb_template_contig(T, R, tm_seg0, d);
        if (lane == 0 && wave == 0) T.nb_tmpl[b] = d;
    }
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
            if (W == 1) {
                if (sb < se && !rows_out) first_site_rows<FS>(T, sb, se, lane, fl, ps);
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


}  // namespace

// (four waves per name block unless the reads are short: see k0_first_site)
void mc_launch_first_site(const DevTable &T, const DevRef &R, const double *qual, double qual_thresh, int k, NbDesc *desc, int64_t *nb_f0,
                          Counters *cnt, int classify, int skip_thresh, unsigned long long pass_no, int hyp, int make_tmpl, hipStream_t st) {
    if (MC_K0_WAVES > 1 && T.n_rows >= (int64_t)T.n_nb * 2048)
        hipLaunchKernelGGL(k0_first_site<MC_K0_WAVES>, dim3((unsigned)T.n_nb), dim3(64 * MC_K0_WAVES), 0, st, T, R, qual, qual_thresh, k,
                           desc, nb_f0, cnt, classify, skip_thresh, pass_no, hyp, make_tmpl);
    else
        hipLaunchKernelGGL(k0_first_site<1>, dim3((unsigned)(((int64_t)T.n_nb * 64 + 255) / 256)), dim3(256), 0, st, T, R, qual, qual_thresh,
                           k, desc, nb_f0, cnt, classify, skip_thresh, pass_no, hyp, make_tmpl);
}

void mc_launch_classify(const DevTable &T, const DevRef &R, NbDesc *desc, const int64_t *nb_f0, int entry_read, int k, int skip_thresh,
                        Counters *cnt, unsigned long long pass_no, hipStream_t st) {
    hipLaunchKernelGGL(k0_classify, dim3((unsigned)((T.n_nb + 255) / 256)), dim3(256), 0, st, T, R, desc, nb_f0, entry_read, k, skip_thresh,
                       cnt, pass_no);
}

void mc_launch_extend(const DevTable &T, NbDesc *desc, const int64_t *nb_f0, int entry_read, Counters *cnt, unsigned long long pass_no,
                      hipStream_t st) {
    hipLaunchKernelGGL(k0_extend, dim3((unsigned)((T.n_nb + 255) / 256)), dim3(256), 0, st, T, desc, nb_f0, entry_read, cnt, pass_no);
}
