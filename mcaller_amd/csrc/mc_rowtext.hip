// mc_rowtext.hip -- the rows of a pass as TEXT, made on the device (gfx950 / MI355X).
//
// What it replaces on the host: mc_format_diffs (mc_format.cpp), the row writer of extract_contexts.py:207-216 -- for a one-base
// motif 1.3 GB of rows per 10^8 events, which the host's cores (rationed, and busy moving the input text into pinned memory)
// made in longer than the text took over the link.  Here the records of a pipelined pass are printed where they are: the packed
// block the side stream's kernel left in HBM (pack_layout / pack_tail: what the host formatter reads from its pinned copy), the
// table's segments, the read names in the shard's TEXT (the device parser noted where each segment's name stands), the contig
// names and the marked reference.  One lane per record:
//
//   k_rt_count     records without MC_I_TOO_MANY per block of 256 records | wide slot means per block of 256 call rows
//   k_rt_scan      the two lists of block sums -> offsets (one workgroup)
//   k_rt_wide      wide slot means before every call row; the 64-bit slot means gathered into one list, in the rows' order
//   k_rt_digits    a lane per NUMBER of that list (and per read quality): its shortest round-trip digits, 16 bytes each -- every lane
//                  of a wave in the digit generation (a lane per row would run it for the rows' 64-bit slots while the rows whose slot
//                  travels as an integer wait, then the other way round, six slots in turn: three times the instructions)
//   k_rt_rows<false>   every record's row, counted from the digits' counts: its length, block sums of the lengths
//   k_rt_scan_len  offsets of the blocks' rows; the total against the room there is
//   k_rt_rows<true>    the rows, written at their offsets
//   k_rt_copy      the text into pinned host memory (by the compute units: the DMA engines are busy with the next shards' text),
//                  with the status block the host reads
//
// A record the device does not print -- a context that leaves the contig, an unknown sub-model key (the reference's exit and
// crash paths), an unscored record, a number outside mc_rowtext.h's range -- sets `host_needed`: the shard's rows then come from the host
// formatter as before.  Numbers: mc_rowtext.h (shortest round-trip digits in exact integer arithmetic, Python's layout).
#include "mc_dev.h"
#include "mc_rowtext.h"

namespace {

constexpr int RT_B = 256;       // records (call rows) per workgroup

__device__ __forceinline__ uint32_t rt_wave_incl(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive prefix of v over the workgroup's 256 threads, *total = the sum (s_w: four words of LDS)
__device__ __forceinline__ uint32_t rt_block_excl(uint32_t v, uint32_t *s_w, uint32_t *total) {
    const uint32_t inc = rt_wave_incl(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; ++i) base += s_w[i];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}

struct PackView {               // the packed block of a pass (pack_layout / pack_tail), on the device
    const int32_t *close32;
    const int64_t *close64;
    const int32_t *site_pos, *site_seg;
    const uint32_t *info;
    const int32_t *lo32;
    const double *prob;
    const unsigned char *wmask;
    const uint32_t *hi32;
};

__device__ __forceinline__ PackView pack_view(const RowTextIn &I) {
    const PackLayout L = pack_layout(I.n, I.close32);
    const PackTail T = pack_tail(L.feats, (size_t)I.m, I.k, (size_t)I.n_wide);
    PackView V;
    V.close32 = I.close32 ? reinterpret_cast<const int32_t *>(I.pack) : nullptr;
    V.close64 = I.close32 ? nullptr : reinterpret_cast<const int64_t *>(I.pack);
    V.site_pos = reinterpret_cast<const int32_t *>(I.pack + L.pos);
    V.site_seg = reinterpret_cast<const int32_t *>(I.pack + L.seg);
    V.info = reinterpret_cast<const uint32_t *>(I.pack + L.info);
    V.lo32 = reinterpret_cast<const int32_t *>(I.pack + T.lo32);
    V.prob = reinterpret_cast<const double *>(I.pack + T.prob);
    V.wmask = I.pack + T.wmask;
    V.hi32 = reinterpret_cast<const uint32_t *>(I.pack + T.hi32);
    return V;
}

__global__ __launch_bounds__(RT_B) void k_rt_count(RowTextIn I, RowTextScratch S, unsigned n_rec_blocks) {
    __shared__ uint32_t s_w[4];
    const PackView V = pack_view(I);
    uint32_t v = 0, total;
    if (blockIdx.x < n_rec_blocks) {
        const int64_t j = (int64_t)blockIdx.x * RT_B + threadIdx.x;
        if (j < I.n) v = (V.info[j] & MC_I_TOO_MANY) ? 0u : 1u;
        (void)rt_block_excl(v, s_w, &total);
        if (threadIdx.x == 0) S.kept_blk[blockIdx.x] = total;
        if (blockIdx.x == 0 && threadIdx.x == 0) { S.st->n_bytes = 0; S.st->n_rows = 0; S.st->host_needed = 0; S.st->too_small = 0; }
    } else {
        const int64_t b = (int64_t)blockIdx.x - n_rec_blocks, r = b * RT_B + threadIdx.x;
        if (r < I.m) v = (uint32_t)__popc((unsigned)V.wmask[r]);
        (void)rt_block_excl(v, s_w, &total);
        if (threadIdx.x == 0) S.wide_blk[b] = total;
    }
}

// block sums -> offsets in place, a[n] = the total (one workgroup; the lists are short: a block of 256 per entry)
__device__ void rt_scan_list(uint32_t *a, int64_t n, uint32_t *s_w) {
    uint32_t carry = 0;
    for (int64_t i0 = 0; i0 < n; i0 += RT_B) {
        const int64_t i = i0 + threadIdx.x;
        const uint32_t v = i < n ? a[i] : 0u;
        uint32_t total;
        const uint32_t ex = rt_block_excl(v, s_w, &total);
        if (i < n) a[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) a[n] = carry;
}

__global__ __launch_bounds__(RT_B) void k_rt_scan(RowTextScratch S, int64_t n_rec_blocks, int64_t n_row_blocks) {
    __shared__ uint32_t s_w[4];
    rt_scan_list(S.kept_blk, n_rec_blocks, s_w);
    rt_scan_list(S.wide_blk, n_row_blocks, s_w);
}

__global__ __launch_bounds__(RT_B) void k_rt_wide(RowTextIn I, RowTextScratch S) {
    __shared__ uint32_t s_w[4];
    const PackView V = pack_view(I);
    const int64_t r = (int64_t)blockIdx.x * RT_B + threadIdx.x;
    uint32_t total;
    const uint32_t v = r < I.m ? (uint32_t)__popc((unsigned)V.wmask[r]) : 0u;
    const uint32_t ex = rt_block_excl(v, s_w, &total);
    if (r < I.m) {
        uint32_t w = S.wide_blk[blockIdx.x] + ex;
        S.wide_pref[r] = w;
        const unsigned mask = V.wmask[r];
        const int32_t *lo = V.lo32 + (size_t)r * I.k;
        for (int s2 = 0; s2 < I.k; ++s2)
            if ((mask >> s2) & 1u) {
                const uint64_t bits = ((uint64_t)V.hi32[w] << 32) | (uint32_t)lo[s2];
                double f;
                __builtin_memcpy(&f, &bits, 8);
                S.wval[w++] = f;
            }
    }
}

// the printed digits of every 64-bit slot mean and of every read's quality: a lane per number
__global__ __launch_bounds__(RT_B) void k_rt_digits(RowTextIn I, RowTextScratch S) {
    const int64_t i = (int64_t)blockIdx.x * RT_B + threadIdx.x;
    if (i >= I.n_wide + I.n_qual) return;
    const double v = i < I.n_wide ? S.wval[i] : I.qual[i - I.n_wide];
    uint64_t lo;
    uint32_t meta;
    rt_num_pack(rt_num_of(v), &lo, &meta);         // (a number this code does not print -- nan, out of range -- says so in its meta word: whether
    S.num_lo[i] = lo;                              //  a row needs it is the row's business, an empty slot's garbage is never looked at)
    S.num_meta[i] = meta;
}

__device__ __forceinline__ int rt_comp_of(int c) {          // base_comps, extract_contexts.py:11; -1: KeyError there
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'N': return 'N';
        case 'M': return 'M';
        default: return -1;
    }
}

// Where a row's characters go in the writing pass: eight at a time once the address is a multiple of eight (a lane's row is its own
// stretch of the text: whole words inside it are its alone; the bytes in front of the first such word and behind the last go one by
// one) -- a store per character was one L2 request per character, 64 lines per instruction.
struct RtStoreWords {
    char *p;                    // next character's place
    uint64_t acc = 0;
    int fill = -1;              // characters gathered in acc; -1: the address has not reached a multiple of eight yet
    __device__ __forceinline__ explicit RtStoreWords(char *at) : p(at) { if ((reinterpret_cast<uintptr_t>(at) & 7u) == 0) fill = 0; }
    __device__ __forceinline__ void put(char c) {
        if (fill < 0) {
            *p++ = c;
            if ((reinterpret_cast<uintptr_t>(p) & 7u) == 0) fill = 0;
            return;
        }
        acc |= (uint64_t)(unsigned char)c << (8 * fill);
        ++p;
        if (++fill == 8) { *reinterpret_cast<uint64_t *>(p - 8) = acc; acc = 0; fill = 0; }
    }
    __device__ __forceinline__ void put_num(const RtNum &n) { rt_put_num(*this, n); }
    __device__ __forceinline__ void flush() {
        for (int i = 0; i < fill; ++i) p[i - fill] = (char)(acc >> (8 * i));
        fill = fill < 0 ? fill : 0;
    }
};

// ... and in the counting pass: a number's length comes from its digits' count
struct RtCountRows {
    int n = 0;
    __device__ __forceinline__ void put(char) { ++n; }
    __device__ __forceinline__ void put_num(const RtNum &num) { n += rt_num_length(num); }
};

// The row of record j (call row `row`) into the sink, as mc_format.cpp format_range writes it; false: the host decides
template <class Sink>
__device__ unsigned rt_row(const RowTextIn &I, const PackView &V, const RowTextScratch &S, int64_t j, uint32_t row, uint32_t info, Sink &o) {
    const uint32_t *wide_pref = S.wide_pref;
    const double p1 = V.prob[row];
    if (!(p1 == p1)) return 1u << 1;                                               // (scored by the host)
    const int32_t seg = V.site_seg[j];
    if (seg < 0 || seg >= I.n_seg) return 1u << 2;
    const int32_t rid = I.seg_read[seg];
    if (rid < 0 || rid >= I.n_qual) return 1u << 3;
    const int64_t crow = V.close32 ? (int64_t)V.close32[j] : V.close64[j];
    // segment that holds the closing row (searchsorted(seg_row_begin, crow, 'right') - 1): the site's own, the next, or a search
    int32_t cseg = seg;
    if (!(crow >= I.seg_begin[seg] && crow < I.seg_begin[seg + 1])) {
        if (seg + 1 < I.n_seg && crow >= I.seg_begin[seg + 1] && crow < I.seg_begin[seg + 2]) cseg = seg + 1;
        else {
            int32_t lo = 0, hi = I.n_seg + 1;                                    // first index with seg_begin[index] > crow
            while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (I.seg_begin[mid] > crow) hi = mid; else lo = mid + 1; }
            cseg = lo - 1;
        }
    }
    int32_t cc;
    if (cseg >= I.n_seg) { cc = I.tail_contig; if (cc < 0) return 1u << 4; }        // R8: the closing row's contig
    else { if (cseg < 0) return 1u << 5; cc = I.seg_contig[cseg]; }
    if (cc < 0 || cc >= I.R.n_contigs) return 1u << 6;
    {
        const char *nm = I.cn_chars + I.cn_off[cc];
        const int len = (int)I.cn_len[cc];
        for (int i = 0; i < len; ++i) o.put(nm[i]);
        o.put('\t');
    }
    {
        const KpSeg sg = I.segs[seg];
        const char *nm = I.text + sg.name_off;
        for (int i = 0; i < sg.name_len; ++i) o.put(nm[i]);
        o.put('\t');
    }
    const int32_t mpos = V.site_pos[j];
    if (mpos < 0) { o.put('-'); rt_put_uint(o, (uint32_t)(-(int64_t)mpos)); }
    else rt_put_uint(o, (uint32_t)mpos);
    o.put('\t');
    // the marked context, last_ref[mpos-k+1 : mpos+k]  (:194; mc_format.cpp build_context)
    const int k = I.k;
    {
        if (info & MC_I_EDGE) return 1u << 7;
        const int32_t cid = I.seg_contig[seg];
        if (cid < 0 || cid >= I.R.n_contigs) return 1u << 8;
        const int64_t L = I.R.contig_len[cid], lo = (int64_t)mpos - k + 1, hi = (int64_t)mpos + k;
        if (lo < 0 || hi > L) return 1u << 9;
        const bool rev = info & MC_I_REV;
        const uint8_t *seq = I.R.seq + I.R.seq_off[cid];
        const uint32_t *bits = (rev ? I.R.mr : I.R.mf) + I.R.word_off[cid];
        const int n = 2 * k - 1;
        char ctx[2 * MC_MAX_K];
        for (int i = 0; i < n; ++i) {
            const int64_t p = lo + i;
            const int c = ((bits[p >> 5] >> (p & 31)) & 1u) ? 'M' : seq[p];
            if (!rev) ctx[i] = (char)c;
            else {
                const int cmp = rt_comp_of(c);
                if (cmp < 0) return 1u << 10;
                ctx[n - 1 - i] = (char)cmp;
            }
        }
        if (ctx[k - 1] != 'M') return 1u << 11;                                      // :224-228
        const unsigned char nxt = (unsigned char)ctx[k];
        if (I.sub_of_char[nxt] == 255) return 1u << 12;                             // :218-223
        if (((info >> MC_I_NEXT_SHIFT) & 0xFFu) != nxt) return 1u << 13;
        for (int i = 0; i < n; ++i) o.put(ctx[i]);
        o.put('\t');
    }
    {
        const uint32_t empty = info & MC_I_EMPTY_MASK;
        const unsigned mask = V.wmask[row];
        uint32_t wide = wide_pref[row];
        const int32_t *lo = V.lo32 + (size_t)row * k;
        for (int s = 0; s < k; ++s) {
            const bool is_wide = (mask >> s) & 1u;
            const uint32_t w = wide;
            wide += is_wide ? 1u : 0u;
            if ((empty >> s) & 1u) o.put('0');                                    // literal int 0  (:186)
            else if (!is_wide) rt_put_fixed4(o, lo[s]);
            else {
                const RtNum num = rt_num_unpack(S.num_lo[w], S.num_meta[w]);       // (k_rt_digits)
                if (!num.ok) return 1u << 14;
                o.put_num(num);
            }
            o.put(',');
        }
    }
    {
        const RtNum num = rt_num_unpack(S.num_lo[(size_t)I.n_wide + rid], S.num_meta[(size_t)I.n_wide + rid]);     // str(read quality)
        if (!num.ok) return 1u << 15;
        o.put_num(num);
    }
    o.put('\t');
    o.put((info & MC_I_REV) ? '-' : '+');
    o.put('\t');
    if (p1 >= 0.5) { for (int i = 0; i < I.lab_meth_len; ++i) o.put(I.lab_meth[i]); }        // :200-206
    else { for (int i = 0; i < I.lab_unmeth_len; ++i) o.put(I.lab_unmeth[i]); }
    o.put('\t');
    if (!rt_put_prob2(o, p1, rint(p1 * 100.0))) return 1u << 16;                    // np.round(p, 2)  (:207)
    o.put('\n');
    return 0u;
}

// WRITE = false: every record's row counted -> rec_len, rec_row, len_blk[block]; true: written at its offset
template <bool WRITE>
__global__ __launch_bounds__(RT_B) void k_rt_rows(RowTextIn I, RowTextScratch S, char *out) {
    __shared__ uint32_t s_w[4];
    const PackView V = pack_view(I);
    const int64_t j = (int64_t)blockIdx.x * RT_B + threadIdx.x;
    if (!WRITE) {
        uint32_t info = MC_I_TOO_MANY, total;
        if (j < I.n) info = V.info[j];
        const uint32_t kept = (info & MC_I_TOO_MANY) ? 0u : 1u;
        const uint32_t row = S.kept_blk[blockIdx.x] + rt_block_excl(kept, s_w, &total);
        uint32_t len = 0;
        if (kept) {
            RtCountRows c;
            const unsigned why = rt_row(I, V, S, j, row, info, c);
            if (!why) len = (uint32_t)c.n;
            else atomicOr(&S.st->host_needed, why);
        }
        if (j < I.n) { S.rec_len[j] = len; S.rec_row[j] = row; }
        (void)rt_block_excl(len, s_w, &total);
        if (threadIdx.x == 0) S.len_blk[blockIdx.x] = total;
    } else {
        if (S.st->host_needed || S.st->too_small) return;                       // (uniform: nothing is written for such a pass)
        uint32_t len = 0, total;
        if (j < I.n) len = S.rec_len[j];
        const unsigned long long at = S.len_blk[blockIdx.x] + rt_block_excl(len, s_w, &total);
        if (len) {
            RtStoreWords w(out + at);
            (void)rt_row(I, V, S, j, S.rec_row[j], V.info[j], w);
            w.flush();
        }
    }
}

// offsets of the blocks' rows (one workgroup); the total and the number of rows -> the status
__global__ __launch_bounds__(RT_B) void k_rt_scan_len(RowTextScratch S, int64_t n_rec_blocks, unsigned long long out_cap) {
    __shared__ uint32_t s_w[4];
    unsigned long long carry = 0;
    for (int64_t i0 = 0; i0 < n_rec_blocks; i0 += RT_B) {
        const int64_t i = i0 + threadIdx.x;
        const uint32_t v = i < n_rec_blocks ? (uint32_t)S.len_blk[i] : 0u;       // (a block's rows: below 2^32 bytes)
        uint32_t total;
        const uint32_t ex = rt_block_excl(v, s_w, &total);
        if (i < n_rec_blocks) S.len_blk[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        S.st->n_bytes = carry;
        S.st->n_rows = S.kept_blk[n_rec_blocks];
        if (carry > out_cap) S.st->too_small = 1;
    }
}

__global__ __launch_bounds__(256) void k_rt_copy(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src, const RowTextStatus *st,
                                                 RowTextStatus *st_host) {
    const RowTextStatus s = *st;
    if (blockIdx.x == 0 && threadIdx.x == 0) *st_host = s;
    if (s.host_needed || s.too_small) return;
    const size_t bytes = (size_t)s.n_bytes, n16 = bytes / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
}

}  // namespace

// scratch a pass of n records, m call rows needs (elements; see RowTextScratch)
void mc_row_text_scratch_sizes(int64_t n, int64_t m, int64_t *rec_blocks, int64_t *row_blocks) {
    *rec_blocks = (n + RT_B - 1) / RT_B;
    *row_blocks = (m + RT_B - 1) / RT_B;
}

// The rows of the pass whose packed block is I.pack -> out (device, out_cap bytes) -> out_host (pinned, as the device sees it),
// status -> st_host (likewise).  n > 0.
void mc_launch_row_text(const RowTextIn &I, const RowTextScratch &S, char *out, size_t out_cap, char *out_host, RowTextStatus *st_host,
                        hipStream_t st) {
    int64_t nbr, nbw;
    mc_row_text_scratch_sizes(I.n, I.m, &nbr, &nbw);
    hipLaunchKernelGGL(k_rt_count, dim3((unsigned)(nbr + nbw)), dim3(RT_B), 0, st, I, S, (unsigned)nbr);
    hipLaunchKernelGGL(k_rt_scan, dim3(1), dim3(RT_B), 0, st, S, nbr, nbw);
    if (nbw > 0) hipLaunchKernelGGL(k_rt_wide, dim3((unsigned)nbw), dim3(RT_B), 0, st, I, S);
    const int64_t n_num = I.n_wide + I.n_qual;
    if (n_num > 0) hipLaunchKernelGGL(k_rt_digits, dim3((unsigned)((n_num + RT_B - 1) / RT_B)), dim3(RT_B), 0, st, I, S);
    hipLaunchKernelGGL(k_rt_rows<false>, dim3((unsigned)nbr), dim3(RT_B), 0, st, I, S, out);
    hipLaunchKernelGGL(k_rt_scan_len, dim3(1), dim3(RT_B), 0, st, S, nbr, (unsigned long long)out_cap);
    hipLaunchKernelGGL(k_rt_rows<true>, dim3((unsigned)nbr), dim3(RT_B), 0, st, I, S, out);
    const unsigned blocks = (unsigned)std::min<size_t>((out_cap / 16 + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(k_rt_copy, dim3(blocks), dim3(256), 0, st, (unsigned char *)out_host, (const unsigned char *)out, (const RowTextStatus *)S.st, st_host);
}
