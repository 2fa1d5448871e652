// mc_classify.hip: predict_proba (k2_mlp, k3_forest, k3_simple) and the packing of a pass's records for the copy-out (k_pack) -- part of libmcaller_hip.so's device side (gfx950 / MI355X); shared structures and helpers: mc_dev.h; the map of the
// kernels: mc_stream.hip.
#include "mc_dev.h"
#include "mc_rows.h"

namespace {

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
// (PIN = false -- the fast forward, whose hidden layer needs none of them: ordinary constants, alive only where they are used)
template <bool PIN>
struct VgprConstT {
    double v;
    __device__ __forceinline__ explicit VgprConstT(double x) : v(x) { if (PIN) asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ operator double() const { return v; }
};
template <bool PIN>
struct ExpConstsT {
    using VgprConst = VgprConstT<PIN>;
    VgprConst two_log2e{2.8853900817779268}, magic{6755399441055744.0}, half_ln2_hi{-0.3465735901845619},
        half_ln2_lo{-9.541074646352939e-11};
    VgprConst t_max4{87.5};        // e^{2t} <= e^175: the product of four such (e^{2t} + 1) stays finite; tanh(87.5) = 1 in double
    VgprConst g9{5.1405589494805136e-05}, g8{0.0002828297056809958}, g7{0.0014109321451518497}, g6{0.00634918945176432},
        g5{0.02539682542470863}, g4{0.08888888907016779}, g3{0.26666666666656197}, g2{0.6666666666659861},
        g1{1.3333333333333335}, g0{2.0000000000000004};
};
using ExpConsts = ExpConstsT<true>;

template <class EC>
__device__ __forceinline__ double exp2t_plus1(double t, const EC &C) {      // (0 <= t <= 350)
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

template <class EC>
__device__ __forceinline__ double tanh_1exp(double x, const EC &C) {
    const double q = recip_ge1(exp2t_plus1(fmin(fabs(x), C.t_max4), C));
    return copysign(fma(-2.0, q, 1.0), x);
}

// four at a time, step by step side by side (four independent chains in flight: the Horner scheme alone is a dependent
// sequence of twelve), and one reciprocal, of the product of the four denominators (each <= e^175 + 1)
template <class EC>
__device__ __forceinline__ void tanh_4(double &x0, double &x1, double &x2, double &x3, const EC &C) {
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
template <class EC>
__device__ __forceinline__ double logistic(double z, const EC &C) {
    const double q = recip_ge1(exp2t_plus1(fmin(0.5 * fabs(z), 350.0), C));   // 1 / (e^{|z|} + 1)
    return z >= 0.0 ? 1.0 - q : q;
}

// ---- the fast forward: fp32 where the answer does not depend on it ----
// The reference prints np.round(p, 2) and the label p >= 0.5 (:200-207): 101 thresholds.  A probability computed in fp32 is
// within a bound of the fp64 one that the weights and the record's inputs give (DevMlp.margin, mc_ctx_set_mlp); a record whose
// fast probability lies farther than that from every threshold prints the same characters either way, and the others -- one in
// a few hundred -- are evaluated again in fp64 (phase D of k2_mlp).  The hidden layer in fp32 issues at twice the rate and the
// tanh is ten instructions instead of thirty: what SURVEY.md section 7 names as the alternative to fp64 throughout.
// tanh(a) = 1 - 2 / (1 + 2^s), s = 2 log2(e) a (the factor is in the weights, DevMlp.wp32): one v_exp_f32, one v_rcp_f32, one fma.
// 2^s = inf gives 1, 2^s = 0 gives -1, s = 0 gives 0 exactly (the padding unit of an odd layer).  Absolute error against
// tanh(s ln2 / 2) <= K2_TANH32_MAX_ERR over all floats s (k_tanh32_err).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tanh32s(float s) {
    return __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(s)), -2.0f, 1.0f);
}
__device__ __forceinline__ f2 tanh32s_2(f2 s) {
    const f2 d = (f2){__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)} + 1.0f;
    const f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return __builtin_elementwise_fma(r, (f2){-2.0f, -2.0f}, (f2){1.0f, 1.0f});
}

// distance of p from the nearest threshold the reference's row depends on: the ties of np.round(p, 2) (p x 100 = j + 0.5) and 0.5
__device__ __forceinline__ double threshold_distance(double p) {
    const double t = p * 100.0;
    const double d = fabs((t - floor(t)) - 0.5) * 0.01;
    return fmin(d, fabs(p - 0.5));
}

// One lane per record, one hidden unit after the other inside the lane, the weights as SCALAR operands: the records a wave
// takes belong to one sub-model, so W1[:, j], b1[j], W2[j] are the same for its 64 lanes -- they come through the scalar
// cache into SGPRs (nine s_load'ed doubles per hidden unit; the fast forward: nine float PAIRS per pair of units) and the vector
// pipe issues nothing but the arithmetic, no LDS reads, no address arithmetic, no butterfly.
//
// A workgroup takes a contiguous stretch of the records (the pass's records divided evenly over the workgroups, K2B at a
// time), a record per thread, and
//   A. finds the records that are scored at all (skipped records and records whose context leaves the contig are not), fetches
//      their read quality (a chain of three dependent loads per record) and lists them sub-model by sub-model in LDS: a wave
//      counts its records of every sub-model with a ballot and claims room in the sub-models' lists with ONE LDS atomic (a lane
//      per sub-model) -- the order of a list is whatever the waves' atomics made it, and nothing depends on it: a record's sums are its own;
//   B. wave w computes a quarter of the hidden units for some of the groups of 64 list entries: the quarter of the SIMD it runs
//      on, so that the four SIMDs of the CU carry the same load whatever the number of groups.  Partial sums go to LDS, under
//      the record's place in the stretch;
//   C. the thread that fetched a record adds its quarters in a fixed order (the result does not depend on which wave ran when)
//      and the logistic function gives the probability.
// Two barriers per stretch (after A, after B) and nothing else that all waves wait for: what A of the next stretch writes (the
// lists, the qualities, the slots, the other of two sets of list lengths) was last read in B of this one, what B writes in C.
// (Until round 5: per-wave counts, a barrier, prefix sums by 64 threads over 16 counts each, a barrier, the lists padded to whole
// groups, a barrier; C over the list entries; the fp64 evaluations of the fast forward behind a fifth barrier in every stretch --
// 10 us of a stretch's 24 with the hidden layer at 10.)
// The earlier version (eight lanes per record, pairs of records per lane group, weights in LDS: 126 LDS reads and ~1200
// VALU instructions per step of 16 records, 3.1 uneven waves per SIMD) took 53 us for the headline pass.
// The records of a fused dense pass (k1_fused) lie in fixed room per piece of the table, the slots a piece did not fill are holes --
// two slots in three.  A stretch of 1024 SLOTS would spend its barriers and its lists on 300 records: the stretches are made
// of whole pieces instead, as many as hold at most 1024 records together (eleven, typically), from the pieces' counts -- every wave
// adds up the same sixteen counts, nothing is shared.
struct K2Pieces {
    const int32_t *cnt;     // [n] records of every piece (k1_fused), nullptr: the records are dense
    int room;               // slots per piece
    int64_t n;              // pieces
};
#ifndef MC_K2_THREADS
#define MC_K2_THREADS 1024
#endif
constexpr int K2_THREADS = MC_K2_THREADS;
constexpr int K2B = K2_THREADS;                 // records per workgroup iteration: one per thread
constexpr int K2_WAVES = K2_THREADS / 64;
constexpr int K2_FIX = K2B + 256;               // the fast forward: room for the records to evaluate again (a stretch adds at most K2B)
static_assert(K2_WAVES >= 4 && K2_WAVES % 4 == 0, "four unit quarters");
static_assert(K2_MAXM <= 16, "a wave's list claims are one atomic of K2_MAXM lanes; the loops over the sub-models are written out");

#ifdef MC_K2_TRACE      // (variant build for tools/k2_trace.py: 100 MHz time stamps of every wave's phases)
__device__ unsigned long long g_k2_trace[1024 * 16 * 16];
#define K2_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024) g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + (i)] = wall_clock64(); g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + 8 + (i)] = clock64(); } while (0)
// ... and of wave 0 in every stretch of the first 256 workgroups: [workgroup][stretch < 24][8]
__device__ unsigned long long g_k2_timeline[256 * 24 * 8];
#define K2_TL(i) do { if (tid == 0 && blockIdx.x < 256 && stretch_no < 24) g_k2_timeline[((size_t)blockIdx.x * 24 + stretch_no) * 8 + (i)] = wall_clock64(); } while (0)
// (wall clock of the side kernel's own phases, an array of their own)
__device__ unsigned long long g_side_trace[1024 * 16 * 16];
#define K2_WALL(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_side_trace[((size_t)blockIdx.x * K2_WAVES + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define K2_STAMP(i) do { } while (0)
#define K2_TL(i) do { } while (0)
#define K2_WALL(i) do { } while (0)
#endif


#define MC_SCALAR_MEM __attribute__((address_space(4)))    // constant address space: uniform loads from it are s_load

// k2_mlp<.., PACK = true>: the side stream of a pipelined pass as ONE kernel -- the workgroup that scores a stretch of records first
// finishes the windows of ITS records that the emit left to the row-by-row walk (what k1_rare_dev did in a launch of its own) and
// packs the stretch for the copy-out as it goes (what k_pack_count + k_pack did in two more launches, reading every record -- and,
// behind k1_fused, every HOLE -- again): the narrow columns of all records, the slot means (32-bit integers where they are
// fl(d / 1e4)) and the probability of the calls, compacted -- the layout of pack_layout() / pack_tail() exactly as k_pack writes it.
// Nothing is shared between workgroups: where a record's rows go follows from counts that are complete when the kernel starts --
// the emit counts calls and wide slot means as it writes the records (k1_emit per packing chunk, Counters-side chunk_cnt; k1_fused
// per piece, piece_kw), a window left to the walk is counted there by the walk's own rule (window_too_many, mc_rows.h).
// score == 0: no classifier (the pass was asked for features only): rare windows and packing alone.
struct SidePack {
    K1Args A;                     // the pass: table, reference, descriptors, records, counters, rare_list, chunk_cnt / piece_kw
    const Payload *sorted;        // its payloads in record order (the walk's input)
    unsigned char *out;           // the block the copy-out moves
    size_t out_bytes;             // ... its size: a pass that needs more is marked (Counters.overflow, pack_need) and repeated
    Counters *host_status;        // the pass's counters as the host reads them (pinned)
    int close32, score;
};
struct NoPack {};
#ifndef MC_SIDE_WAVES
#define MC_SIDE_WAVES 4
#endif
// (ROOMY: 128 registers a lane and one workgroup per CU -- the side kernel of a sparse motif, where a CU gets one workgroup anyway and
// a stretch: the packing's and the walk's registers fit beside the hidden layer's without a spill)
// (TH: threads per workgroup = records per stretch.  512 for the side kernel of a dense reference: two workgroups per CU at 128
// registers a lane -- one's loads and packing behind the other's hidden layer)
template <int NI_T, bool FAST, bool PACK = false, bool ROOMY = false, int TH = K2_THREADS>
__global__ __launch_bounds__(TH) __attribute__((amdgpu_waves_per_eu((FAST && !ROOMY) ? 8 : (ROOMY && FAST ? MC_SIDE_WAVES : 4), (FAST && !ROOMY) ? 8 : (ROOMY && FAST ? MC_SIDE_WAVES : 4)))) void k2_mlp(DevMlp M, const double *__restrict__ feats, int k,
                                                     const int32_t *__restrict__ site_seg, const int32_t *__restrict__ seg_read,
                                                     const double *__restrict__ qual, const uint32_t *__restrict__ info,
                                                     const uint8_t *__restrict__ submodel_in, int64_t n,
                                                     double *__restrict__ prob, const unsigned long long *__restrict__ n_dev,
                                                     const unsigned int *__restrict__ overflow, K2Pieces P,
                                                     std::conditional_t<PACK, SidePack, NoPack> SP) {
    if constexpr (PACK) {
        // (the host reads the pass's counters from pinned memory, whatever became of the pass)
        if (*overflow) {
            constexpr unsigned head_words = offsetof(Counters, end_of_head) / 4;
            if (blockIdx.x == 0 && threadIdx.x < head_words)
                reinterpret_cast<volatile unsigned int *>(SP.host_status)[threadIdx.x] = reinterpret_cast<const unsigned int *>(SP.A.cnt)[threadIdx.x];
            return;
        }
    } else if (overflow && *overflow) return;   // (pipelined pass with record buffers too small: it is repeated)
    if (n_dev) n = min(n, (int64_t)*n_dev);     // the count is on the device only (pipelined passes): n is the capacity
    K2_WALL(0);
    constexpr int KB = TH, KW = TH / 64, KFIX = TH + 256;      // records per stretch, waves, room for the records to evaluate again
    static_assert(KW >= 4 && KW % 4 == 0, "four unit quarters");
    constexpr int NX = NI_T ? NI_T : MC_MAX_K + 1;
    const int H = M.n_hidden, NI = NI_T ? NI_T : M.n_in, S = NI + 2, NM = min(M.n_models, K2_MAXM);
    using part_t = std::conditional_t<FAST, float, double>;
    __shared__ uint16_t s_list[K2_MAXM][KB];   // per sub-model: its records (their places in the stretch), in the order of the waves' claims
    __shared__ int s_tot[2][K2_MAXM];           // ... their number: one set per stretch, the other is cleared meanwhile
    __shared__ double s_q[KB];                 // read quality of the stretch's records
    __shared__ part_t s_part[4][KB];           // partial output sums of the four unit quarters, by the record's place
    __shared__ float s_marg[FAST ? KB : 1];    // the fast forward: how far from the fp64 probability the record's may lie
    __shared__ unsigned long long s_fixlist[FAST ? KFIX : 1];    // ... the records to evaluate again in fp64: record | sub-model << 56
    __shared__ int s_nfix;
    __shared__ int32_t s_slot[KB];             // a stretch made of pieces: the slot of every record, from the stretch's first piece's first slot
    __shared__ uint32_t s_fixrow[(FAST && PACK) ? KFIX : 1];   // PACK: ... and their rows in the packed block
    __shared__ int2 s_pk[PACK ? 2 : 1][PACK ? KW : 1];    // PACK: a wave's calls and their wide slot means in the stretch (two sets, like s_tot)
    __shared__ unsigned long long s_base[PACK ? 2 : 1][4];      // PACK: calls, wide slot means, records in front of the stretch (two sets)
    __shared__ unsigned long long s_red[PACK ? 6 : 1][PACK ? KW : 1];
    __shared__ unsigned long long s_lay[PACK ? 8 : 1];     // PACK: where the packed block's columns begin (byte offsets): looked up where a stretch is packed,
                                                           // not carried through the hidden layer in registers the weights need
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the compiler has to know that this is uniform: scalar loads)
    const unsigned long long below = (1ull << lane) - 1ull;
    const ExpConstsT<!FAST> C;
    K2_STAMP(0);
    // Which quarter of the hidden units a wave takes: the one of the SIMD it runs on, so that the four SIMDs of the CU carry
    // a quarter of the arithmetic each whatever the number of groups (the waves of one SIMD share its groups).  (If the
    // workgroup's waves did not land on all four SIMDs: by wave number.)
    __shared__ uint32_t s_simd_of_wave[KW / 4];       // a byte per wave
    const int simd = (int)__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4) & 3;      // HW_ID[5:4]
    // (the sub-model of a record's context character: from LDS, not a load behind the record's info word)
    __shared__ uint8_t s_soc[256];
    if (tid < 256) s_soc[tid] = M.sub_of_char ? M.sub_of_char[tid] : (uint8_t)255;
    if (lane == 0) reinterpret_cast<uint8_t *>(s_simd_of_wave)[wave] = (uint8_t)simd;
    if (tid < 2 * K2_MAXM) (&s_tot[0][0])[tid] = 0;
    if (tid == 0) s_nfix = 0;
    __syncthreads();
    int quarter, g_first, g_step;
    {
        uint32_t per_simd = 0;                  // a byte per SIMD: its waves
        int slot = 0;                           // waves of this wave's SIMD with a smaller number
        for (int w4 = 0; w4 < KW / 4; ++w4) {
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
        g_step = by_simd ? (int)((per_simd >> (8 * simd)) & 0xFFu) : KW / 4;
    }
    const bool pieces = P.cnt != nullptr;
    const int64_t n_units = pieces ? P.n : n;                    // what the workgroups share out: pieces, or records
    const int64_t per = (n_units + gridDim.x - 1) / gridDim.x;
    int64_t lo = min(n_units, blockIdx.x * per), hi = min(n_units, lo + per);
    // ---- PACK: where this workgroup's records go in the packed block -- calls, wide slot means and records in front of its range, and
    // in all (the emit's counts: nothing here waits for another workgroup); the pass's counters for the host; then the windows of its
    // own records that were left to the row-by-row walk ----
    K2_WALL(1);
    if constexpr (PACK) {
        unsigned long long v[6] = {0, 0, 0, 0, 0, 0};           // calls, wide, records: in all; in front of the range
        if (pieces) {
            // (one word per piece: records | calls << 9 | wide slot means << 18; thirty-two pieces per thread and turn, eight 16-byte
            // loads side by side: a turn is a round trip, and a table of 10^8 rows has 10^5 pieces -- a piece at a time this took a
            // workgroup 45-85 us.  The array has room behind the last piece)
            for (int64_t j0 = 32 * (int64_t)tid; j0 < P.n; j0 += 32 * TH) {
                int4 k4[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) k4[e] = *reinterpret_cast<const int4 *>(SP.A.piece_kw + min(j0 + 4 * e, (P.n - 1) & ~(int64_t)3));
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const int64_t j = j0 + e;
                    if (j >= P.n) continue;
                    const int4 kk = k4[e >> 2];
                    const uint32_t kw = (uint32_t)((e & 3) == 0 ? kk.x : (e & 3) == 1 ? kk.y : (e & 3) == 2 ? kk.z : kk.w);
                    const uint32_t c = kw & 511u, kept = (kw >> 9) & 511u, wide = kw >> 18;
                    v[0] += kept; v[1] += wide; v[2] += c;
                    if (j < lo) { v[3] += kept; v[4] += wide; v[5] += c; }
                }
            }
        } else {
            // (the emit counted per packing chunk: the ranges are whole chunks)
            const int64_t per_c = (n + PACK_WGS - 1) / PACK_WGS;
            const int c0 = (int)((int64_t)blockIdx.x * PACK_WGS / gridDim.x), c1 = (int)((int64_t)(blockIdx.x + 1) * PACK_WGS / gridDim.x);
            lo = min(n, c0 * per_c); hi = min(n, c1 * per_c);
            for (int j = tid; j < PACK_WGS; j += TH) {
                const unsigned long long kk = SP.A.chunk_cnt[PACK_PAD * j], ww = SP.A.chunk_cnt[PACK_PAD * j + 1];
                v[0] += kk; v[1] += ww;
                if (j < c0) { v[3] += kk; v[4] += ww; }
            }
            if (tid == 0) { v[2] = (unsigned long long)n; v[5] = (unsigned long long)lo; }
        }
        K2_WALL(2);
        // (added up across the wave, then one 64-bit LDS atomic per wave and sum.  Not sixteen partial sums per value read back by
        // everybody: the compiler fetches all ninety-six at once and spills them)
        if (tid < 6) s_red[PACK ? tid : 0][0] = 0ull;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            unsigned long long x = v[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if (lane == 0 && x) atomicAdd(&s_red[PACK ? i : 0][0], x);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = s_red[PACK ? i : 0][0];
        __syncthreads();            // (s_red has been read)
        // (the block is sized for the records a table of this size is expected to leave, not for every slot of a fused pass: a pass that
        // needs more says how much and is repeated -- every workgroup comes to the same conclusion from the same sums)
        const size_t need_bytes = pack_tail(pack_layout((int64_t)v[2], SP.close32).feats, (size_t)v[0], k, (size_t)v[1]).end;
        const bool too_big = need_bytes > SP.out_bytes;
        if (too_big) {
            if (blockIdx.x == 0 && tid == 0) {
                atomicExch(&SP.A.cnt->pack_need, (unsigned long long)need_bytes);
                const unsigned was = atomicOr(&SP.A.cnt->overflow, 1u);
                asm volatile("" :: "v"(was));
            }
            lo = hi;                                   // (no stretch; the counters still go to the host, below)
        }
        if (tid == 0) {
            const PackLayout PL = pack_layout((int64_t)v[2], SP.close32);
            const PackTail PTl = pack_tail(PL.feats, (size_t)v[0], k, (size_t)v[1]);
            s_lay[0] = PL.pos; s_lay[1] = PL.seg; s_lay[2] = PL.info; s_lay[3] = PTl.lo32; s_lay[4] = PTl.wmask; s_lay[5] = PTl.hi32;
            s_lay[6] = PTl.prob;
            s_base[0][0] = v[3]; s_base[0][1] = v[4]; s_base[0][2] = v[5];
            s_red[0][1] = v[2]; s_red[1][1] = v[0]; s_red[2][1] = v[1];         // (records, calls, wide slot means in all: for the host, at the end)
        }
        __syncthreads();
        K2_WALL(3);
        const int64_t n_rare = too_big ? 0 : (int64_t)SP.A.cnt->n_rare;
        if (n_rare > 0) {
            const int64_t rlo = pieces ? lo * P.room : lo, rhi = pieces ? hi * P.room : hi;
            for (int64_t i = tid; i < n_rare; i += TH) {
                const int64_t q = SP.A.rare_list[i];
                // (a slot of more than 128 events -- NumPy's recursion proper, 1.4 KB of stack per lane: not in this kernel; the pass
                // is marked like one with irregular reads and repeated by the synchronous path)
#ifndef MC_SIDE_NO_RARE      // (variant build, timing only: what the walk's registers cost the kernel)
                if (q >= rlo && q < rhi && finish_rare_record<false>(SP.A, SP.sorted, q, true)) {
                    // (an atomic at device scope, waited for: performed where the workgroup that is through last will read it)
                    const unsigned long long was = atomicExch(&SP.A.cnt->irregular_pass, SP.A.pass_no);
                    asm volatile("" :: "v"(was));
                }
#endif
            }
            // (this workgroup's own threads read what the walks wrote: same CU, same L1 -- no device-scope fence, which on this chip
            // writes a whole L2 back: 80 us a kernel when every thread executed one)
            __syncthreads();
        }
    }
    K2_WALL(4);
    K2_STAMP(1);
    // the records whose printed digits (or label) could depend on the precision: again in fp64, a wave per record, the hidden
    // units across its lanes (the quality is fetched again: a handful of records per stretch)
    auto again_in_fp64 = [&](int n_fix) {
        for (int f = wave; f < n_fix; f += KW) {
            const unsigned long long ent = s_fixlist[FAST ? f : 0];
            const int64_t r = (int64_t)(ent & ((1ull << 56) - 1ull));
            const int mdl = (int)(ent >> 56);
            double x[NX];
            const double q = qual[seg_read[site_seg[r]]];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = i < k ? feats[r * k + i] : (i == k ? q : 0.0);
            double part = 0.0;
            for (int u = lane; u < H; u += 64) {
                const double *wu = M.wu + ((size_t)mdl * H + u) * S;
                double a0 = x[0] * wu[0];
#pragma unroll
                for (int i = 1; i < NX; ++i)
                    if (i < NI) a0 = fma(x[i], wu[i], a0);
                part = fma(tanh_1exp(a0 + wu[NI], C), wu[NI + 1], part);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            if (lane == 0) {
                const double pr = logistic(part + M.b2[mdl], C);
                prob[r] = pr;
                if constexpr (PACK) reinterpret_cast<double *>(SP.out + s_lay[PACK ? 6 : 0])[s_fixrow[(FAST && PACK) ? f : 0]] = pr;
            }
        }
    };
    int buf = 0, stretch_no = 0;
    (void)stretch_no;
    int64_t base = lo;                          // first record of the stretch (pieces: first piece)
    while (base < hi) {
        K2_STAMP(7);                            // (the stretch begins: slot 15 keeps its clock64, slot 7 is overwritten below)
        K2_TL(0);
        int n_here, np_here = 1, my_slot = tid;
        int64_t slot0 = base;                   // record of the stretch's place `off`: slot0 + (pieces ? s_slot[off] : off)
        if (pieces) {
            // the next pieces that hold at most KB records together (a piece holds at most its room, and that is below KB)
            const bool have = lane < 16 && base + lane < hi;
            const int c = have ? min(max(P.cnt[base + lane], 0), P.room) : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            np_here = max(1, __popcll(__ballot(have && incl <= KB)));
            n_here = min(__builtin_amdgcn_readlane(incl, np_here - 1), KB);
            int before = 0, j = 0;              // the thread's piece: the last one with at most tid records in front of it
#pragma unroll
            for (int t = 0; t < 15; ++t) {
                const int inc_t = __builtin_amdgcn_readlane(incl, t);
                if (t + 1 < np_here && inc_t <= tid) { j = t + 1; before = inc_t; }
            }
            my_slot = j * P.room + (tid - before);
            slot0 = base * (int64_t)P.room;
            s_slot[tid] = my_slot;
        } else n_here = (int)min((int64_t)KB, hi - base);
        // ---- A: the thread's record, and the lists
        // (the read quality is a chain of three dependent loads -- segment, read, quality)
        const int64_t r = slot0 + my_slot;
        int mi = 255;                               // sub-model of record r (255: not scored here)
        double qv = 0.0;
        [[maybe_unused]] int p_nw = -1;                // PACK: wide slot means of the thread's record if it is a call, else -1
        if (tid < n_here) {
            if (submodel_in) mi = submodel_in[r];
            else {
                const uint32_t inf = info[r];
                const int32_t seg = site_seg[r];
                if constexpr (PACK) { if (!(inf & MC_I_TOO_MANY)) p_nw = __popc((uint32_t)SP.A.O.wmask[r] & ((1u << k) - 1u)); }
                bool scored = !(inf & (MC_I_TOO_MANY | MC_I_EDGE));
                if constexpr (PACK) scored = scored && SP.score != 0;
                if (scored) {
                    mi = s_soc[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
                    qv = qual[seg_read[seg]];
                }
            }
        }
        if constexpr (PACK) {
            // the wave's calls and their wide slot means in this stretch: the packing (behind C) finds its rows from the waves' counts
            const int nk = __popcll(__ballot(p_nw >= 0));
            int nw = max(p_nw, 0);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nw += __shfl_xor(nw, o);
            if (lane == 0) s_pk[PACK ? buf : 0][PACK ? wave : 0] = make_int2(nk, nw);
        }
        int rank = 0, cnt_lane = 0;                 // (lane m: the wave's records of sub-model m)
#pragma unroll
        for (int m = 0; m < K2_MAXM; ++m) {         // (a key outside the models is the KeyError path, :218: the host decides)
            const unsigned long long bal = __ballot(mi == m && m < NM);
            if (mi == m) rank = __popcll(bal & below);
            if (lane == m) cnt_lane = __popcll(bal);
        }
        int wbase = 0;
        if (lane < K2_MAXM && cnt_lane) wbase = atomicAdd(&s_tot[buf][lane], cnt_lane);
        int mybase = 0;
#pragma unroll
        for (int m = 0; m < K2_MAXM; ++m) {
            const int b = __builtin_amdgcn_readlane(wbase, m);
            mybase = mi == m ? b : mybase;
        }
        if (mi < NM) s_list[mi][mybase + rank] = (uint16_t)tid;
        s_q[tid] = qv;
        K2_STAMP(2);
        K2_WALL(5);
        K2_TL(1);
        __syncthreads();
        K2_WALL(6);
        K2_TL(2);
        int tot[K2_MAXM], gs[K2_MAXM], n_groups = 0;    // records and first group of every sub-model
#pragma unroll
        for (int m = 0; m < K2_MAXM; ++m) {
            tot[m] = __builtin_amdgcn_readfirstlane(s_tot[buf][m]);
            gs[m] = n_groups;
            n_groups += (tot[m] + 63) >> 6;
        }
        if (tid < K2_MAXM) s_tot[buf ^ 1][tid] = 0;     // (the next stretch's: last read a stretch ago)
        K2_STAMP(3);
        K2_WALL(7);
        // ---- B: a quarter of the hidden units for every fourth (eighth ...) group
        const int u0 = quarter * H / 4, u1 = (quarter + 1) * H / 4;
        for (int g = g_first; g < n_groups; g += g_step) {
            int mdl = 0, tot_m = tot[0], lg = g;    // the group's sub-model, its records, the group's number among its groups
#pragma unroll
            for (int m = 1; m < K2_MAXM; ++m)
                if (g >= gs[m] && tot[m] > 0) { mdl = m; tot_m = tot[m]; lg = g - gs[m]; }
            const int idx = lg * 64 + lane;
            const bool valid = idx < tot_m;         // (the lanes behind the list's end compute the group's first record again)
            const int off = s_list[mdl][valid ? idx : lg * 64];
            const int64_t rr = slot0 + (pieces ? s_slot[off] : off);
            double x[NX];
            if (submodel_in) {                       // plain batched call: X rows of n_in values
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = i < NI ? feats[rr * NI + i] : 0.0;
            } else {                                 // flush records: k slot means + read quality (:189-193)
                const double q = s_q[off];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = i < k ? feats[rr * k + i] : (i == k ? q : 0.0);
            }
            double z = 0.0;
            if constexpr (FAST) {
                // (two hidden units per instruction: v_pk_fma_f32 with the units' weights side by side in an SGPR pair, the record's
                // input in both halves of a register pair.  The weights and the bias carry the factor 2 log2(e) -- DevMlp.wp32 -- so that
                // the sum IS the exponent: tanh(a) = 1 - 2 / (1 + 2^s), one v_exp_f32, one v_rcp_f32, one packed fma per pair)
                f2 xx[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) xx[i] = (f2){(float)x[i], (float)x[i]};
                const int H2 = (H + 1) >> 1;
                const int p0 = quarter * H2 / 4, p1 = (quarter + 1) * H2 / 4;
                const MC_SCALAR_MEM f2 *wp = (const MC_SCALAR_MEM f2 *)M.wp32 + ((size_t)mdl * H2 + p0) * S;
                f2 zz = {0.0f, 0.0f};
                int p = p0;
                for (; p + 2 <= p1; p += 2, wp += 2 * S) {          // two independent chains of pairs
                    f2 a = xx[0] * wp[0], b = xx[0] * wp[S];
#pragma unroll
                    for (int i = 1; i < NX; ++i)
                        if (i < NI) {
                            a = __builtin_elementwise_fma(xx[i], wp[i], a);
                            b = __builtin_elementwise_fma(xx[i], wp[S + i], b);
                        }
                    a += wp[NI]; b += wp[S + NI];
                    zz = __builtin_elementwise_fma(tanh32s_2(a), wp[NI + 1], zz);
                    zz = __builtin_elementwise_fma(tanh32s_2(b), wp[S + NI + 1], zz);
                }
                for (; p < p1; ++p, wp += S) {
                    f2 a = xx[0] * wp[0];
#pragma unroll
                    for (int i = 1; i < NX; ++i)
                        if (i < NI) a = __builtin_elementwise_fma(xx[i], wp[i], a);
                    zz = __builtin_elementwise_fma(tanh32s_2(a + wp[NI]), wp[NI + 1], zz);
                }
                z = (double)(zz.x + zz.y);                           // (stored as the float it is)
                if (quarter == 0 && valid) {                         // how far the record's probability may lie from the fp64 one
                    const MC_SCALAR_MEM float *mg = (const MC_SCALAR_MEM float *)M.margin + (size_t)mdl * (MC_MAX_K + 2);
                    float m = mg[0];
#pragma unroll
                    for (int i = 0; i < NX; ++i)
                        if (i < NI) m = __builtin_fmaf(fabsf(xx[i].x) * 1.0000002f, mg[1 + i], m);
                    s_marg[FAST ? off : 0] = m * 1.000001f;
                }
            } else {
                const MC_SCALAR_MEM double *wu = (const MC_SCALAR_MEM double *)M.wu + ((size_t)mdl * H + u0) * S;
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
            }
            if (valid) s_part[quarter][off] = (part_t)z;
        }
        K2_STAMP(4);
        K2_WALL(8);
        K2_TL(3);
        __syncthreads();
        K2_WALL(9);
        K2_STAMP(5);
        K2_TL(4);
        // ---- C: the output unit of the thread's record
        [[maybe_unused]] double p_pr = __longlong_as_double(0x7ff8000000000000LL);     // (a call that is not scored here: NaN, as the records say)
        [[maybe_unused]] bool p_fix = false;
        if (mi < NM) {
            const double z = (((double)s_part[0][tid] + (double)s_part[1][tid]) + (double)s_part[2][tid]) + (double)s_part[3][tid];
            const double pr = logistic(z + M.b2[mi], C);
            prob[r] = pr;
            const bool fix = FAST && !(threshold_distance(pr) > (double)s_marg[FAST ? tid : 0]);     // (NaN too)
            if constexpr (PACK) { p_pr = pr; p_fix = fix; }
            else if (fix) s_fixlist[FAST ? atomicAdd(&s_nfix, 1) : 0] = (unsigned long long)r | (unsigned long long)mi << 56;
        }
        if constexpr (PACK) {
            // ---- the stretch's records into the packed block: the narrow columns of every record, slot means, mask byte and probability
            // of the calls.  Everything a record needs is fetched HERE (the hidden layer's registers are free again; the record's
            // lines are in the caches from A): nothing of it lives through B ----
            K2_WALL(11);
            // (the next stretch's bases: written whatever this stretch holds -- sixteen pieces without a record, a long read under
            // the quality threshold, are a stretch of nothing, and the stretch behind it reads what is written here)
            if (tid == 0) {
                unsigned kept_s = 0, wide_s = 0;
                for (int w = 0; w < KW; ++w) { const int2 c = s_pk[PACK ? buf : 0][PACK ? w : 0]; kept_s += (unsigned)c.x; wide_s += (unsigned)c.y; }
                s_base[PACK ? buf ^ 1 : 0][0] = s_base[PACK ? buf : 0][0] + kept_s;
                s_base[PACK ? buf ^ 1 : 0][1] = s_base[PACK ? buf : 0][1] + wide_s;
                s_base[PACK ? buf ^ 1 : 0][2] = s_base[PACK ? buf : 0][2] + (unsigned long long)n_here;
            }
            if (tid < n_here) {
                const uint32_t inf = info[r];
                const int32_t seg = site_seg[r], pos = SP.A.O.site_pos[r];
                const int64_t close = SP.A.O.close_row[r];
                const bool keep = !(inf & MC_I_TOO_MANY);
                const uint32_t wm = keep ? ((uint32_t)SP.A.O.wmask[r] & ((1u << k) - 1u)) : 0u;
                double fx[MC_MAX_K];
#pragma unroll
                for (int f = 0; f < MC_MAX_K; ++f) fx[f] = (keep && f < k) ? feats[r * k + f] : 0.0;
                // calls and wide slot means in front of the record: the lanes below, the waves before (their counts: A), the stretches before
                const unsigned long long kb = __ballot(keep);
                const int nw = __popc(wm);
                int w_incl = nw;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int y = __shfl_up(w_incl, o);
                    if (lane >= o) w_incl += y;
                }
                unsigned long long row = s_base[PACK ? buf : 0][0] + (unsigned)__popcll(kb & below), wpos = s_base[PACK ? buf : 0][1] + (unsigned)(w_incl - nw);
#pragma unroll
                for (int w = 0; w < KW; ++w) {
                    const int2 c = s_pk[PACK ? buf : 0][PACK ? w : 0];
                    if (w < wave) { row += (unsigned)c.x; wpos += (unsigned)c.y; }
                }
                const unsigned long long real0 = s_base[PACK ? buf : 0][2];
                const int64_t io = pieces ? (int64_t)real0 + tid : r;               // (holes are never looked at: the host never sees one)
                unsigned char *const out = SP.out;
                if (SP.close32) reinterpret_cast<int32_t *>(out)[io] = (int32_t)close;
                else reinterpret_cast<int64_t *>(out)[io] = close;
                reinterpret_cast<int32_t *>(out + s_lay[0])[io] = pos;
                reinterpret_cast<int32_t *>(out + s_lay[PACK ? 1 : 0])[io] = seg;
                reinterpret_cast<uint32_t *>(out + s_lay[PACK ? 2 : 0])[io] = inf;
                if (keep) {
                    int32_t *o_lo = reinterpret_cast<int32_t *>(out + s_lay[PACK ? 3 : 0]) + row * k;
                    uint32_t *o_hi = reinterpret_cast<uint32_t *>(out + s_lay[PACK ? 5 : 0]);
                    (out + s_lay[PACK ? 4 : 0])[row] = (uint8_t)wm;
                    reinterpret_cast<double *>(out + s_lay[PACK ? 6 : 0])[row] = p_pr;
#pragma unroll
                    for (int f = 0; f < MC_MAX_K; ++f)
                        if (f < k) {
                            // a slot mean that is fl(d / 1e4) travels as d (the emit noted which are not: DevRecords.wmask)
                            if (!((wm >> f) & 1u)) o_lo[f] = (int32_t)rint(fx[f] * 1e4);
                            else {
                                const unsigned long long bits = (unsigned long long)__double_as_longlong(fx[f]);
                                o_lo[f] = (int32_t)(uint32_t)bits;
                                o_hi[wpos++] = (uint32_t)(bits >> 32);
                            }
                        }
                    if (p_fix) {
                        const int at = atomicAdd(&s_nfix, 1);
                        s_fixlist[FAST ? at : 0] = (unsigned long long)r | (unsigned long long)mi << 56;
                        s_fixrow[(FAST && PACK) ? at : 0] = (uint32_t)row;
                    }
                }
            }
        }
        base += pieces ? np_here : KB;
        buf ^= 1;
        K2_STAMP(6);
        K2_WALL(10);
        K2_TL(5);
        ++stretch_no;
#ifdef MC_K2_TRACE
        if (lane == 0 && blockIdx.x < 1024)
            g_k2_trace[((size_t)blockIdx.x * K2_WAVES + wave) * 16 + 7] = (unsigned)simd | (unsigned)quarter << 4 | (unsigned)g_first << 8 | (unsigned)g_step << 16 | (unsigned long long)n_groups << 24;
#endif
    }
    if constexpr (FAST) {
        __syncthreads();
        again_in_fp64(s_nfix);
    }
    K2_WALL(13);
    if constexpr (PACK) {
        // ---- the pass's counters for the host (pinned memory; read behind ev_done), by the workgroup that is through LAST: what the
        // row-by-row walks of the others changed (a pass marked for repetition) is in them.  As k_pack sent them: the head as it is,
        // calls and wide slot means in all, the records without the holes ----
        // (nothing but that mark crosses workgroups, and it is an atomic at device scope: no fence -- a device-scope release writes
        // a whole L2 back on this chip)
        __shared__ int s_last;
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&SP.A.cnt->side_done, 1ull) == (unsigned long long)gridDim.x - 1ull;
        __syncthreads();
        if (s_last) {
            constexpr unsigned head_words = offsetof(Counters, end_of_head) / 4;
            constexpr int kept_word = (int)(offsetof(Counters, n_kept) / 4);
            static_assert(offsetof(Counters, n_wide) == offsetof(Counters, n_kept) + 8 && offsetof(Counters, n_records) == 0, "Counters layout");
            if (tid < (int)head_words && (tid < kept_word || tid >= kept_word + 4) && tid >= 2)
                reinterpret_cast<volatile unsigned int *>(SP.host_status)[tid] =
                    __hip_atomic_load(reinterpret_cast<unsigned int *>(SP.A.cnt) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                *reinterpret_cast<volatile unsigned long long *>(&SP.host_status->n_records) = s_red[0][PACK ? 1 : 0];
                *reinterpret_cast<volatile unsigned long long *>(&SP.host_status->n_kept) = s_red[PACK ? 1 : 0][PACK ? 1 : 0];
                *reinterpret_cast<volatile unsigned long long *>(&SP.host_status->n_wide) = s_red[PACK ? 2 : 0][PACK ? 1 : 0];
            }
        }
    }
    K2_WALL(14);
}

// ---------------------------------------------------------------------------------------------------
// K3: random-forest predict_proba (classifier RF, train_model.py:39-45; call site :199).  One lane per record: inputs cast
// to float32 (scikit-learn's DTYPE), each tree walked with `x[feature] <= threshold` to a leaf, p1 = v1/(v0+v1) per tree
// (predict_proba normalises the leaf values), summed over the trees in order and divided by their number.
// ---------------------------------------------------------------------------------------------------
// Two shapes, picked by the number of records the pass really has (on the device for pipelined passes):
//   few records (a shard of a streamed file in positions mode: a few thousand) -- a WAVE per record, a lane per tree: one lane
//   walking 50 trees one after the other is ~1100 dependent loads, 200 us for 14 000 records however few they are; side by side the
//   trees of a record are ~22 dependent loads.  The per-tree fractions are then added in tree order by every lane (v_readlane
//   with the tree as a scalar): the sum is the one the sequential walk gives, bit for bit;
//   many records (a one-base motif: millions) -- a lane per record as before: the loads in flight are what counts there.
constexpr int K3_THREADS = 256;
__global__ __launch_bounds__(K3_THREADS) void k3_forest(DevForest F, const double *__restrict__ feats, int k,
                                                        const int32_t *__restrict__ site_seg, const int32_t *__restrict__ seg_read,
                                                        const double *__restrict__ qual, const uint32_t *__restrict__ info,
                                                        const uint8_t *__restrict__ submodel_in, int64_t n,
                                                        double *__restrict__ prob, const unsigned long long *__restrict__ n_dev,
                                                        const unsigned int *__restrict__ overflow) {
    __shared__ double s_x[K3_THREADS][MC_MAX_K + 2];
    if (overflow && *overflow) return;          // (pipelined pass with record buffers too small: it is repeated)
    if (n_dev) n = min(n, (int64_t)*n_dev);     // the count is on the device only (pipelined passes): n is the capacity
    const int NI = F.n_in;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (K3_THREADS / 64);
    if (n <= n_waves * 16) {
        // ---- a wave per record, a lane per tree ----
        double *x = s_x[wave * 64];
        for (int64_t r = (int64_t)blockIdx.x * (K3_THREADS / 64) + wave; r < n; r += n_waves) {
            int mi;
            if (submodel_in) {
                mi = submodel_in[r];
                if (lane < NI) x[lane] = (double)(float)feats[r * NI + lane];
            } else {
                const uint32_t inf = info[r];
                if (inf & (MC_I_TOO_MANY | MC_I_EDGE)) continue;            // (the same for all lanes)
                mi = F.sub_of_char[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
                if (lane < k) x[lane] = (double)(float)feats[r * k + lane];
                if (lane == k) x[k] = (double)(float)qual[seg_read[site_seg[r]]];
            }
            if (mi >= F.n_models) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (one wave: its LDS operations execute in order)
            const int t0 = F.model_tree_off[mi], t1 = F.model_tree_off[mi + 1];
            double sum = 0.0;
            for (int tb = t0; tb < t1; tb += 64) {
                const int t = tb + lane;
                double q = 0.0;
                if (t < t1) {
                    int node = F.tree_node_off[t];
                    int l;
                    while ((l = F.left[node]) >= 0) node = (x[F.feature[node]] <= F.threshold[node]) ? l : F.right[node];
                    const double v0 = F.value[2 * (size_t)node], v1 = F.value[2 * (size_t)node + 1];
                    double norm = (-0.0 + v0) + v1;
                    if (norm == 0.0) norm = 1.0;
                    q = v1 / norm;
                }
                const int here = min(64, t1 - tb);
                for (int j = 0; j < here; ++j)                              // in tree order, as the sequential walk adds them
                    sum += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), j), __builtin_amdgcn_readlane(__double2loint(q), j));
            }
            if (lane == 0) prob[r] = sum / (double)(t1 - t0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (x is rewritten by the next record)
        }
        return;
    }
    // ---- a lane per record ----
    double *x = s_x[tid];
    for (int64_t r = (int64_t)blockIdx.x * K3_THREADS + tid; r < n; r += (int64_t)gridDim.x * K3_THREADS) {
        int mi;
        if (submodel_in) {
            mi = submodel_in[r];
            for (int i = 0; i < NI; ++i) x[i] = (double)(float)feats[r * NI + i];
        } else {
            const uint32_t inf = info[r];
            if (inf & (MC_I_TOO_MANY | MC_I_EDGE)) continue;
            mi = F.sub_of_char[(inf >> MC_I_NEXT_SHIFT) & 0xFFu];
            for (int i = 0; i < k; ++i) x[i] = (double)(float)feats[r * k + i];
            x[k] = (double)(float)qual[seg_read[site_seg[r]]];
        }
        if (mi >= F.n_models) continue;
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

// chunk_cnt[PACK_PAD * b] = kept records of chunk b, chunk_cnt[PACK_PAD * b + 1] = their wide slots; holes != 0 (the fused pass
// of a dense reference, k1_fused: fixed room per piece): chunk_cnt[PACK_PAD * b + 2] = the chunk's records that are not holes
__global__ __launch_bounds__(PACK_THREADS) void k_pack_count(DevRecords O, const Counters *__restrict__ cnt, int k,
                                                             unsigned long long *__restrict__ chunk_cnt, int holes) {
    __shared__ unsigned int s_wave[3][PACK_THREADS / 64];
    const int64_t n = cnt->overflow ? 0 : min((int64_t)cnt->n_records, O.capacity);
    const int64_t per = (n + PACK_WGS - 1) / PACK_WGS;
    const int64_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    unsigned int kept = 0, wide = 0, real = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PACK_THREADS) {
        const uint32_t info = O.info[i];
        if (!(info & MC_I_HOLE)) real += 1u;
        if (info & MC_I_TOO_MANY) continue;
        kept += 1u;
        const unsigned wm = O.wmask[i];                      // (k1_emit's note; 0xFF: a record of the rare paths, looked at here)
        if (wm != 0xFFu) wide += (unsigned)__popc(wm);
        else
            for (int f = 0; f < k; ++f) {
                int32_t d;
                wide += slot_is_narrow(O.feats[i * k + f], &d) ? 0u : 1u;
            }
    }
    for (int o = 32; o > 0; o >>= 1) { kept += __shfl_xor(kept, o); wide += __shfl_xor(wide, o); real += __shfl_xor(real, o); }
    if ((threadIdx.x & 63) == 0) { s_wave[0][threadIdx.x >> 6] = kept; s_wave[1][threadIdx.x >> 6] = wide; s_wave[2][threadIdx.x >> 6] = real; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0, w = 0, r = 0;
        for (int j = 0; j < PACK_THREADS / 64; ++j) { t += s_wave[0][j]; w += s_wave[1][j]; r += s_wave[2][j]; }
        chunk_cnt[PACK_PAD * blockIdx.x] = t;
        chunk_cnt[PACK_PAD * blockIdx.x + 1] = w;
        if (holes) chunk_cnt[PACK_PAD * blockIdx.x + 2] = r;
    }
}

__global__ __launch_bounds__(PACK_THREADS) void k_pack(DevRecords O, const Counters *__restrict__ cnt,
                                                       const unsigned long long *__restrict__ chunk_cnt,
                                                       unsigned char *__restrict__ out, size_t out_bytes, int k, int close32,
                                                       Counters *__restrict__ host_status, int holes, int look) {
    // look != 0: the chunk counts come from k_pack_count, which looked at the slot means of the records whose mask byte says 0xFF (the
    // row-by-row paths) -- so does this kernel; look == 0: the emit counted as it wrote the records, every mask byte is what it says
    static_assert(PACK_WGS <= PACK_THREADS && PACK_THREADS % 64 == 0, "one chunk count per thread");
    __shared__ unsigned long long s_sum[6][PACK_THREADS / 64];
    __shared__ unsigned int s_wave[3][PACK_THREADS / 64];
    __shared__ double s_feats[PACK_THREADS * MC_MAX_K];     // the strip's slot means, loaded with consecutive lanes on consecutive words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // kept records (and their wide slots) before this chunk, and in all chunks
    unsigned long long v = tid < PACK_WGS ? chunk_cnt[PACK_PAD * tid] : 0ull, before = tid < (int)blockIdx.x ? v : 0ull;
    unsigned long long w = tid < PACK_WGS ? chunk_cnt[PACK_PAD * tid + 1] : 0ull, wbefore = tid < (int)blockIdx.x ? w : 0ull;
    // (a pass with holes: the records that are not holes, before this chunk and in all -- what the narrow columns are indexed by)
    unsigned long long r = (holes && tid < PACK_WGS) ? chunk_cnt[PACK_PAD * tid + 2] : 0ull, rbefore = tid < (int)blockIdx.x ? r : 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o); before += __shfl_xor(before, o);
        w += __shfl_xor(w, o); wbefore += __shfl_xor(wbefore, o);
        r += __shfl_xor(r, o); rbefore += __shfl_xor(rbefore, o);
    }
    if (lane == 0) { s_sum[0][wave] = v; s_sum[1][wave] = before; s_sum[2][wave] = w; s_sum[3][wave] = wbefore; s_sum[4][wave] = r; s_sum[5][wave] = rbefore; }
    __syncthreads();
    unsigned long long total = 0, base = 0, total_wide = 0, wbase = 0, total_real = 0, rbase = 0;
    for (int j = 0; j < PACK_THREADS / 64; ++j) {
        total += s_sum[0][j]; base += s_sum[1][j]; total_wide += s_sum[2][j]; wbase += s_sum[3][j]; total_real += s_sum[4][j]; rbase += s_sum[5][j];
    }
    constexpr unsigned head_words = offsetof(Counters, end_of_head) / 4;      // everything the host looks at
    constexpr int kept_word = (int)(offsetof(Counters, n_kept) / 4);          // (n_kept and n_wide: two words each, set below)
    static_assert(offsetof(Counters, n_wide) == offsetof(Counters, n_kept) + 8, "n_kept, n_wide side by side");
    if (blockIdx.x == 0) {
        static_assert(offsetof(Counters, n_records) == 0, "the record count: the block's first two words");
        if (tid < (int)head_words && (tid < kept_word || tid >= kept_word + 4) && !(holes && tid < 2 && !cnt->overflow))     // (holes: the count that goes out is set below)
            reinterpret_cast<volatile unsigned int *>(host_status)[tid] = reinterpret_cast<const unsigned int *>(cnt)[tid];
        if (tid == 0) {
            *reinterpret_cast<volatile unsigned long long *>(&host_status->n_kept) = total;
            *reinterpret_cast<volatile unsigned long long *>(&host_status->n_wide) = total_wide;
        }
    }
    if (cnt->overflow) return;                                           // (the host runs such a pass again, synchronously)
    {
        // (the block is sized for the records expected, not for every slot: a pass that needs more says so and is repeated)
        const int64_t n_all = min((int64_t)cnt->n_records, O.capacity);
        const size_t need = pack_tail(pack_layout(holes ? (int64_t)total_real : n_all, close32).feats, (size_t)total, k, (size_t)total_wide).end;
        if (need > out_bytes) {
            if (blockIdx.x == 0 && tid == 0) {
                *reinterpret_cast<volatile unsigned long long *>(&host_status->pack_need) = (unsigned long long)need;
                *reinterpret_cast<volatile unsigned int *>(&host_status->overflow) = 1u;
            }
            return;
        }
    }
    const int64_t n = min((int64_t)cnt->n_records, O.capacity);
    const int64_t per = (n + PACK_WGS - 1) / PACK_WGS;
    const int64_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    // (holes: the host is told the number of records that are not holes -- written behind the copy of the counters above by the
    // same thread, so it stands)
    const int64_t n_out = holes ? (int64_t)total_real : n;
    if (holes && blockIdx.x == 0 && tid == 0) *reinterpret_cast<volatile unsigned long long *>(&host_status->n_records) = (unsigned long long)n_out;
    const PackLayout L = pack_layout(n_out, close32);
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
        const uint32_t info = i < hi ? O.info[i] : (MC_I_TOO_MANY | MC_I_HOLE);
        const bool valid = i < hi && !(info & MC_I_HOLE);
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
            const bool literal = noted != 0xFFu || !look;
            for (int f = 0; f < k; ++f) {
                const double x = s_feats[tid * k + f];
                int32_t d = 0;
                const bool narrow = literal ? !((noted >> f) & 1u) : slot_is_narrow(x, &d);
                if (narrow) lo32[f] = literal ? (int32_t)rint(x * 1e4) : d;
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
        const unsigned long long vmask = __ballot(valid);
        if (lane == 63) s_wave[1][wave] = w_incl;
        if (lane == 0) { s_wave[0][wave] = (unsigned int)__popcll(kmask); s_wave[2][wave] = (unsigned int)__popcll(vmask); }
        __syncthreads();
        unsigned int in_strip = (unsigned int)__popcll(kmask & ((1ull << lane) - 1ull)), strip = 0, w_off = w_incl - n_w, w_strip = 0;
        unsigned int r_in_strip = (unsigned int)__popcll(vmask & ((1ull << lane) - 1ull)), r_strip = 0;
        for (int j = 0; j < PACK_THREADS / 64; ++j) {
            if (j < wave) { in_strip += s_wave[0][j]; w_off += s_wave[1][j]; r_in_strip += s_wave[2][j]; }
            strip += s_wave[0][j];
            w_strip += s_wave[1][j];
            r_strip += s_wave[2][j];
        }
        if (valid) {
            const int64_t io = holes ? (int64_t)(rbase + r_in_strip) : i;        // (holes are compacted away: the host never sees one)
            if (close32) o_close32[io] = (int32_t)O.close_row[i];
            else o_close[io] = O.close_row[i];
            o_pos[io] = O.site_pos[i];
            o_seg[io] = O.site_seg[i];
            o_info[io] = info;
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
        rbase += r_strip;
    }
}


}  // namespace

// k2_mlp workgroups: the records are divided evenly over them (the kernel does that with the count on the device); one
// per CU, fewer for a handful of records
#ifndef MC_K2_WG_PER_CU
#define MC_K2_WG_PER_CU 1
#endif
#ifndef MC_SIDE_WGS_SPARSE
#define MC_SIDE_WGS_SPARSE 128
#endif
#ifndef MC_K2_WG_MANY          // ... where there are millions of records (the fast forward)
#define MC_K2_WG_MANY 2
#endif

// the classifier of a context -- MLP (k2_mlp: the 7-input instance for k = 6, the reference's models, or the general one), forest
// (k3_forest) or one of the closed forms (k3_simple) -- over n records (n_dev: the count is on the device, n is the capacity)
void mc_launch_classifier(const DevMlp &M, const DevForest &F, const DevSimple &S, int n_cu, hipStream_t st, const double *feats, int k,
                          const int32_t *site_seg, const int32_t *seg_read, const double *qual, const uint32_t *info,
                          const uint8_t *submodel_in, int64_t n, double *prob, const unsigned long long *n_dev, const unsigned int *overflow,
                          const int32_t *piece_cnt, int piece_room, int64_t n_pieces) {
    if (n <= 0) return;
    const K2Pieces P{(piece_room > 0 && piece_room < K2B) ? piece_cnt : nullptr, piece_room, n_pieces};
    if (F.left)                // (a wave per record or a lane per record: the kernel looks at the count, which may be on the device only)
        hipLaunchKernelGGL(k3_forest, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)n_cu * 8))), dim3(K3_THREADS), 0, st,
                           F, feats, k, site_seg, seg_read, qual, info, submodel_in, n, prob, n_dev, overflow);
    else if (S.params)
        hipLaunchKernelGGL(k3_simple, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, S, feats, k, site_seg, seg_read, qual, info,
                           submodel_in, n, prob, n_dev, overflow);
    else {
        const bool fast = M.fast && M.wp32 && !submodel_in;
        // (two workgroups per CU where there are millions of records -- a one-base motif; the passes of a sparse motif score a few
        // hundred thousand beside the next pass's scan, and a second workgroup per CU there takes the scan's wave slots: k2_mlp
        // 33 -> 93 us, the scan 164 -> 184)
        const bool many = P.cnt != nullptr || n >= ((int64_t)4 << 20);
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)n_cu * ((fast && many) ? MC_K2_WG_MANY : MC_K2_WG_PER_CU)));
        // (flush records: the fast forward unless the context was told otherwise; a plain batched call -- the estimator protocol,
        // mc_mlp_forward -- is fp64 throughout: its caller gets raw probabilities)
        if (M.n_in == 7 && fast)
            hipLaunchKernelGGL((k2_mlp<7, true>), dim3(grid), dim3(K2_THREADS), 0, st, M, feats, k, site_seg, seg_read, qual, info, submodel_in, n,
                               prob, n_dev, overflow, P, NoPack{});
        else if (M.n_in == 7)
            hipLaunchKernelGGL((k2_mlp<7, false>), dim3(grid), dim3(K2_THREADS), 0, st, M, feats, k, site_seg, seg_read, qual, info, submodel_in, n,
                               prob, n_dev, overflow, P, NoPack{});
        else if (fast)
            hipLaunchKernelGGL((k2_mlp<0, true>), dim3(grid), dim3(K2_THREADS), 0, st, M, feats, k, site_seg, seg_read, qual, info, submodel_in, n,
                               prob, n_dev, overflow, P, NoPack{});
        else
            hipLaunchKernelGGL((k2_mlp<0, false>), dim3(grid), dim3(K2_THREADS), 0, st, M, feats, k, site_seg, seg_read, qual, info, submodel_in, n,
                               prob, n_dev, overflow, P, NoPack{});
    }
}

void mc_launch_pack_count(const DevRecords &O, const Counters *cnt, int k, unsigned long long *chunk_cnt, int holes, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_count, dim3(PACK_WGS), dim3(PACK_THREADS), 0, st, O, cnt, k, chunk_cnt, holes);
}

void mc_launch_pack(const DevRecords &O, const Counters *cnt, const unsigned long long *chunk_cnt, unsigned char *out, size_t out_bytes, int k,
                    int close32, Counters *host_status, int holes, int look, hipStream_t st, hipEvent_t stop) {
    if (stop) hipExtLaunchKernelGGL(k_pack, dim3(PACK_WGS), dim3(PACK_THREADS), 0, st, nullptr, stop, 0, O, cnt, chunk_cnt, out, out_bytes, k, close32, host_status, holes, look);
    else hipLaunchKernelGGL(k_pack, dim3(PACK_WGS), dim3(PACK_THREADS), 0, st, O, cnt, chunk_cnt, out, out_bytes, k, close32, host_status, holes, look);
}

// The side stream of a pipelined pass as ONE kernel (k2_mlp<.., PACK>): rare windows, MLP (score != 0), packing.  -> false: not for this
// pass (another classifier, a piece's room beyond a stretch): the caller launches k1_rare_dev, the classifier and the packing kernels.
bool mc_launch_side(const DevMlp &M, bool other_classifier, int n_cu, hipStream_t st, const K1Args &A, const Payload *sorted, const int32_t *seg_read,
                    const double *qual, int64_t cap, int score, unsigned char *out, size_t out_bytes, int close32, Counters *host_status,
                    int piece_room, int64_t n_pieces, hipEvent_t stop) {
    static const bool off = getenv("MCALLER_SIDE_FUSED") && atoi(getenv("MCALLER_SIDE_FUSED")) == 0;       // (tests, profiles: the three kernels)
    if (off || cap <= 0) return false;
    if (score && (other_classifier || !M.W1)) return false;
    const bool by_piece = piece_room > 0;
    constexpr int TH_SIDE = MC_SIDE_MAX_ROOM;
    if (by_piece ? (piece_room >= TH_SIDE || !A.piece_kw || !A.piece_cnt) : !A.chunk_cnt) return false;
    SidePack SP;
    SP.A = A; SP.sorted = sorted; SP.out = out; SP.out_bytes = out_bytes; SP.host_status = host_status; SP.close32 = close32; SP.score = score;
    const K2Pieces P{by_piece ? A.piece_cnt : nullptr, piece_room, n_pieces};
    const bool fast = score && M.fast && M.wp32;
    // Workgroups of 512 threads (a stretch: 512 records) at 128 registers a lane, two to a CU -- with 1024 threads and 64 registers
    // the packing spills inside the stretch loop (dense: 2.2 ms per 10^8 rows against 1.2), and one workgroup of 1024 per CU is no
    // faster than two of 512.  A dense reference: two per CU, the pieces shared out evenly.  A sparse motif: a power of two of them --
    // the ranges are whole packing chunks -- and FEW: the kernel runs beside the next pass's scan, and what counts is how little of
    // the chip it takes from it, not when it is through.  (10^8 rows, 2 x 10^5 records, pipelined step: 256 workgroups of 1024
    // threads 0.2745 ms, 128 / 64 / 32 of them 0.2638 / 0.2637 / 0.2634; of 512 threads: 256 / 128 / 64 / 32 0.2602 / 0.2570 /
    // 0.2633 / 0.2778; the three kernels this one replaces: 0.2732.  MCALLER_SIDE_GRID: another number)
    unsigned grid;
    const int grid_env = getenv("MCALLER_SIDE_GRID") ? atoi(getenv("MCALLER_SIDE_GRID")) : 0;      // (read per launch: tests change it)
    if (by_piece) grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_pieces, grid_env > 0 ? (int64_t)grid_env : (int64_t)n_cu * (MC_SIDE_WAVES / 2)));
    else {
        const int want = grid_env > 0 ? grid_env : MC_SIDE_WGS_SPARSE;
        grid = 1;
        while (grid * 2 <= (unsigned)std::min<int64_t>(std::min(want, PACK_WGS), (int64_t)n_cu * 2)) grid *= 2;
    }
    const unsigned long long *n_dev = (const unsigned long long *)&A.cnt->n_records;
    const unsigned int *overflow = (const unsigned int *)&A.cnt->overflow;
#define MC_SIDE_LAUNCH(NI, FA) hipExtLaunchKernelGGL((k2_mlp<NI, FA, true, true, TH_SIDE>), dim3(grid), dim3(TH_SIDE), 0, st, nullptr, stop, 0, M, (const double *)A.O.feats, A.k, \
        (const int32_t *)A.O.site_seg, seg_read, qual, (const uint32_t *)A.O.info, (const uint8_t *)nullptr, cap, A.O.prob, n_dev, overflow, P, SP)
    if (score && M.n_in == 7 && fast) MC_SIDE_LAUNCH(7, true);
    else if (score && M.n_in == 7) MC_SIDE_LAUNCH(7, false);
    else if (fast) MC_SIDE_LAUNCH(0, true);
    else MC_SIDE_LAUNCH(0, false);
#undef MC_SIDE_LAUNCH
    return true;
}

// tanh32s(s) against the fp64 tanh(s ln2 / 2) over EVERY float s: the largest absolute difference (what K2_TANH32_MAX_ERR has to cover)
namespace {
__global__ void k_tanh32_err(unsigned long long *out) {
    unsigned long long worst = 0ull;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        if (!(fabsf(x) <= 3.4e38f)) continue;                // (NaN, inf)
        const double err = fabs((double)tanh32s(x) - tanh((double)x * 0.34657359027997264));
        const unsigned long long b = (unsigned long long)__double_as_longlong(err);     // (non-negative doubles order like their bits)
        worst = b > worst ? b : worst;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(worst, o); worst = v > worst ? v : worst; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, worst);
}
}  // namespace
extern "C" int mc_debug_tanh32_max_err(double *out) {
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 8));
    HIP_TRY(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_tanh32_err, dim3(4096), dim3(256), 0, 0, d);
    unsigned long long h = 0;
    HIP_TRY(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    memcpy(out, &h, 8);
    return 0;
}

#ifdef MC_K2_TRACE
extern "C" int mc_debug_k2_timeline(unsigned long long *out, int64_t n_words) {
    if (n_words > 256 * 24 * 8) n_words = 256 * 24 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_timeline), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
extern "C" int mc_debug_side_trace(unsigned long long *out, int64_t n_words) {
    if (n_words > 1024 * 16 * 16) n_words = 1024 * 16 * 16;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_side_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
extern "C" int mc_debug_k2_trace(unsigned long long *out, int64_t n_words) {
    if (n_words > 1024 * 16 * 16) n_words = 1024 * 16 * 16;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -10;
}
#endif
