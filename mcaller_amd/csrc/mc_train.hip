// libmcaller_hip.so -- the classifier fit behind `--train` on the GPU (gfx950 / MI355X).  C ABI: include/mcaller_hip.h.
//
// The reference fits scikit-learn's MLPClassifier(hidden_layer_sizes=(100), alpha=0.001, activation='tanh') on the
// labelled feature rows the hot path produced (train_model.py:47,:81-100) and scores it with 5-fold GroupKFold
// (train_model.py:62-65,:92): six independent fits per sub-model.  Here every fit is four workgroups that run the whole Adam
// optimisation on chip and meet once per batch (a counter in global memory, release / acquire fences around it, the waves' partial
// sums in global memory in two sets taken in turn; a workgroup that waits for seconds fails the call instead of hanging); what ONE
// workgroup does:
//
//   * four waves; lane l of every wave owns hidden units l and l+64 (H <= 128): their input weights, bias, output weight,
//     both Adam moments and the gradient accumulators live in that lane's registers -- nothing is re-read per step;
//   * a minibatch (<= 200 rows of <= 9 doubles) is staged in LDS, the next one is prefetched into registers while the
//     current one is processed; wave w takes rows w, w+4, ...: forward (tanh), one wave reduction for the output unit,
//     logistic + log-loss, backward into the register accumulators;
//   * per batch the four waves exchange their partial gradients through LDS and add them in a fixed order, so every wave
//     holds the same totals and applies the same Adam update to its own copy of the parameters: results are
//     deterministic and independent of scheduling;
//   * epoch order: a 4-round Feistel permutation of the row index, cycle-walked into range (no shuffle buffer); start
//     weights: Glorot-uniform from a counter-based generator -- both defined in oracle/mlp_fit_oracle.py, which restates
//     the same optimiser on the CPU and is pinned against scikit-learn's own runs.
//
// A workgroup is bound by what ONE wave per SIMD can issue in fp64: per row and wave ~300 instructions -- the two units' dot products
// and tanh (written out: 27 instructions against the library's ~100), the wave sum, exp and log of the output, the backward sums.
// Config 5's six fits (9 244 rows, 200 epochs, 9 400 Adam steps each), one workgroup per fit: 0.71 s; the written-out tanh, two
// rows at a time per wave (their chains side by side), exp and log once for the two rows (lane r takes row r's output): 0.45; fused
// multiply-adds: 0.41; FOUR workgroups per fit, on one XCD, meeting once per batch: **0.25 s**.  (Eight waves per workgroup halve a
// lane's registers and spill: 0.75 s; three / four rows at a time: 0.49 / 0.53.)  fp64 throughout, like scikit-learn.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mcaller_hip.h"

void mc_set_error(const char *fmt, ...);
int mc_internal_device(const mc_ctx *c);
hipStream_t mc_internal_stream(const mc_ctx *c);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            mc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -10;                                                                     \
        }                                                                                   \
    } while (0)

// (x * w + z as one fma in this file: the dot products and the gradient sums are half the instructions of a row, and nothing here
// is held to a CPU sum bit for bit -- the fit oracle and scikit-learn's own runs are matched to stated tolerances; 0.45 -> 0.41 s)
#pragma clang fp contract(fast)

namespace {

#ifndef MC_FIT_WAVES
#define MC_FIT_WAVES 4
#endif
constexpr int FW = MC_FIT_WAVES;       // waves per fit: one per SIMD -- a lane holds 100 doubles of state (parameters, both moments, gradients): eight
                                       // waves halve the registers a lane may have and spill (measured: 747 ms for config 5's six fits against 509)
constexpr int FT = FW * 64;
#ifndef MC_FIT_ROWS
#define MC_FIT_ROWS 2
#endif
constexpr int FROWS = MC_FIT_ROWS;     // rows a wave takes at a time (their fp64 chains side by side)
#ifndef MC_FIT_GROUPS
#define MC_FIT_GROUPS 4                // workgroups per fit (config 5's six fits: 1 / 2 / 4 / 8 / 16 workgroups: 0.42 / 0.29 / 0.25 / 0.31-0.34 / 0.95 s --
                                       // the meeting costs ~10 us per batch and grows with the workgroups that have to arrive)
#endif
constexpr int DMAX = MC_MAX_K + 1;     // inputs: k slot means + read quality
constexpr int HMAX = 128;              // two hidden units per lane
constexpr int NC = 2 * (DMAX + 2);     // gradient components per lane: W1[DMAX][2], b1[2], W2[2]
constexpr int NP = NC + 2;             // + the wave's loss sum and output-bias gradient
constexpr int NPH = NP / 2;            // ... exchanged through LDS in two halves (FW x NPH x 64 doubles: 48 KB)
static_assert(NP % 2 == 0, "two halves");

struct FitJob {
    long long tr_off, n_tr, va_off, n_va;
    unsigned long long seed;
};

struct FitArgs {
    const double *X;         // [n_samples * d]
    const uint8_t *y;        // [n_samples] 0/1
    const FitJob *jobs;
    const int32_t *tr_idx, *va_idx;
    int d, H, batch, max_iter, n_iter_no_change, shuffle;
    double alpha, lr, beta1, beta2, eps, tol;
    const double *init;      // optional start weights per job [d*H + 2H + 1], or null
    double *W1, *b1, *W2, *b2;   // per job: [d*H], [H], [H], [1]
    double *loss_curve;      // [n_jobs * max_iter]
    int32_t *n_iter;         // [n_jobs]
    long long *val_correct;  // [n_jobs]
    int G;                   // workgroups per fit
    int n_jobs, by_xcd;      // fits; fit j = workgroups j, j + 8, ... (n_jobs <= 8) instead of j G .. j G + G - 1
    double *xpart;           // [n_jobs][2][G * FW][NP][64]: the waves' partial sums of a batch (two sets, taken in turn)
    unsigned *bar;           // [n_jobs]: workgroups that have written their partial sums, over all batches so far
    int *failed;             // a workgroup waited for the others for seconds: the call fails
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned long long index) {
    return (double)(splitmix64(seed + index * 0xD1342543DE82EF95ull) >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

// bijection of [0, n): 4-round Feistel network on the next even power of two, cycle-walked into range
__device__ __forceinline__ uint32_t feistel_perm(uint32_t i, uint32_t n, uint32_t key, int half, uint32_t mask) {
    uint32_t x = i;
    for (;;) {
        uint32_t L = x >> half, R = x & mask;
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t F = mix32(R * 0x9E3779B1u + key + r * 0x85EBCA6Bu) & mask;
            const uint32_t nl = R;
            R = L ^ F;
            L = nl;
        }
        x = (L << half) | R;
        if (x < n) return x;
    }
}

// tanh(x) = sign(x) (1 - 2 / (e^{2|x|} + 1)) written out (the scheme of mc_classify.hip's fp64 forward, where it is derived: magic-number
// rounding, a degree-11 polynomial fitted at Chebyshev nodes, 2^n from one integer instruction, the hardware's reciprocal estimate and
// two Newton steps): 25 instructions against the library's ~100 with their branches; absolute error < 1e-15.
__device__ __forceinline__ double fit_tanh(double x) {
    const double t = fmin(fabs(x), 87.5);
    const double tt = fma(t, 2.8853900817779268, 6755399441055744.0);
    const double n = tt - 6755399441055744.0;
    double r = fma(n, -0.3465735901845619, t);
    r = fma(n, -9.541074646352939e-11, r);
    double p = 5.1405589494805136e-05;
    p = fma(p, r, 0.0002828297056809958);
    p = fma(p, r, 0.0014109321451518497);
    p = fma(p, r, 0.00634918945176432);
    p = fma(p, r, 0.02539682542470863);
    p = fma(p, r, 0.08888888907016779);
    p = fma(p, r, 0.26666666666656197);
    p = fma(p, r, 0.6666666666659861);
    p = fma(p, r, 1.3333333333333335);
    p = fma(p, r, 2.0000000000000004);
    p = fma(p, r, 2.0);
    p = fma(p, r, 1.0);
    const double d = fma(p, __hiloint2double((__double2loint(tt) + 1023) << 20, 0), 1.0);      // e^{2t} + 1 >= 2
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    q = fma(fma(-d, q, 1.0), q, q);
    return copysign(fma(-2.0, q, 1.0), x);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(FT) void k4_mlp_fit(FitArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double(*s_part)[NPH][64] = reinterpret_cast<double(*)[NPH][64]>(s_raw);                 // [FW][NPH][64]
    double *s_x = reinterpret_cast<double *>(s_raw + sizeof(double) * FW * NPH * 64);       // [batch * d]
    uint8_t *s_y = reinterpret_cast<uint8_t *>(s_x + (size_t)A.batch * A.d);                // [batch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // G workgroups per fit: the rows of a batch are dealt to their G x FW waves, every workgroup holds the parameters and takes the
    // same Adam step from the same totals (added in the same order everywhere: the results do not depend on which wave ran when)
    // (the workgroups of a fit sit on ONE XCD when there are at most eight fits: consecutive workgroup numbers go to the XCDs in turn,
    // so fit j takes the numbers j, j + 8, j + 16, ... -- what they exchange stays in that XCD's L2)
    const int G = A.G;
    const int job = A.by_xcd ? (int)(blockIdx.x % 8) : (int)(blockIdx.x / G), part = A.by_xcd ? (int)(blockIdx.x / 8) : (int)(blockIdx.x % G);
    if (job >= A.n_jobs) return;
    const int gw = part * FW + wave, GW = G * FW;
    unsigned step = 0;                          // batches so far (what the fit's workgroups count at their meeting point)
    const FitJob J = A.jobs[job];
    const int d = A.d, H = A.H;
    const uint32_t n = (uint32_t)J.n_tr;
    const int B = (int)min((long long)A.batch, J.n_tr);
    const int h[2] = {lane, lane + 64};
    const bool valid[2] = {h[0] < H, h[1] < H};

    // ---- parameters, moments, gradient accumulators: registers ----
    double w1[DMAX][2], bb1[2], w2[2], b2;
    double m_w1[DMAX][2], m_b1[2], m_w2[2], m_b2 = 0.0;
    double v_w1[DMAX][2], v_b1[2], v_w2[2], v_b2 = 0.0;
    {
        const double bound_h = sqrt(6.0 / (double)(d + H)), bound_o = sqrt(6.0 / (double)(H + 1));
        const double *init = A.init ? A.init + (size_t)job * ((size_t)d * H + 2 * (size_t)H + 1) : nullptr;
        auto start = [&](unsigned long long idx, double bound) -> double {
            return init ? init[idx] : -bound + 2.0 * bound * uniform01(J.seed, idx);
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < DMAX; ++i) {
                w1[i][j] = (valid[j] && i < d) ? start((unsigned long long)i * H + h[j], bound_h) : 0.0;
                m_w1[i][j] = v_w1[i][j] = 0.0;
            }
            bb1[j] = valid[j] ? start((unsigned long long)d * H + h[j], bound_h) : 0.0;
            w2[j] = valid[j] ? start((unsigned long long)d * H + H + h[j], bound_o) : 0.0;
            m_b1[j] = v_b1[j] = m_w2[j] = v_w2[j] = 0.0;
        }
        b2 = start((unsigned long long)d * H + 2 * (unsigned long long)H, bound_o);
    }

    int bits = 2;
    while (bits < 32 && (1ull << bits) < (unsigned long long)n) ++bits;
    bits += bits & 1;
    const int half = bits / 2;
    const uint32_t mask = (1u << half) - 1u;

    // row `slot` of the epoch's order -> registers of thread `slot % FT` (prefetch), later -> LDS
    double xr[DMAX];
    uint8_t yr = 0;
    auto fetch = [&](uint32_t key, long long b0, int nb) {
        if (tid < nb) {
            const uint32_t pos = (uint32_t)(b0 + tid);
            const uint32_t src = A.shuffle ? feistel_perm(pos, n, key, half, mask) : pos;
            const long long sid = A.tr_idx[J.tr_off + src];
#pragma unroll
            for (int i = 0; i < DMAX; ++i) xr[i] = i < d ? A.X[sid * d + i] : 0.0;
            yr = A.y[sid];
        }
    };
    auto stage = [&](int nb) {
        if (tid < nb) {
#pragma unroll
            for (int i = 0; i < DMAX; ++i)
                if (i < d) s_x[tid * d + i] = xr[i];
            s_y[tid] = yr;
        }
    };

    double b1t = 1.0, b2t = 1.0;               // beta^t
    double best = INFINITY;
    int no_improve = 0, n_epochs = 0;
    const double feps = 2.220446049250313e-16;  // np.finfo(float64).eps: probabilities are clipped to [eps, 1-eps]

    uint32_t key = (uint32_t)(splitmix64((J.seed ^ 0xA5A5A5A55A5A5A5Aull) + 0ull) & 0xFFFFFFFFull);
    if (n > 0) fetch(key, 0, B);
    for (int epoch = 0; epoch < A.max_iter && n > 0; ++epoch) {
        double acc = 0.0;
        for (long long b0 = 0; b0 < (long long)n; b0 += B) {
            const int nb = (int)min((long long)B, (long long)n - b0);
            stage(nb);
            __syncthreads();
            {   // prefetch the batch after this one (the next epoch's first batch at the end of an epoch)
                long long nb0 = b0 + B;
                uint32_t nkey = key;
                if (nb0 >= (long long)n) {
                    nb0 = 0;
                    nkey = (uint32_t)(splitmix64((J.seed ^ 0xA5A5A5A55A5A5A5Aull) + (unsigned long long)(epoch + 1)) & 0xFFFFFFFFull);
                }
                fetch(nkey, nb0, (int)min((long long)B, (long long)n - nb0));
            }
            double g_w1[DMAX][2], g_b1[2] = {0.0, 0.0}, g_w2[2] = {0.0, 0.0}, g_b2 = 0.0, loss = 0.0;
#pragma unroll
            for (int i = 0; i < DMAX; ++i) g_w1[i][0] = g_w1[i][1] = 0.0;
            // (FROWS rows at a time: a wave is alone on its SIMD, and one row is a handful of dependent fp64 chains -- tanh, the wave
            // sum, exp, log -- that leave it idle; the second row's chains run beside the first's.  The sums take the rows in the
            // order they always had: w, w + FW, w + 2 FW, ...)
            for (int s = gw; s < nb; s += FROWS * GW) {
                int sr[FROWS];
#pragma unroll
                for (int r = 0; r < FROWS; ++r) sr[r] = s + r * GW < nb ? s + r * GW : s;
                double x[FROWS][DMAX], yy[FROWS], a[FROWS][2], part[FROWS];
#pragma unroll
                for (int r = 0; r < FROWS; ++r) {
#pragma unroll
                    for (int i = 0; i < DMAX; ++i) x[r][i] = i < d ? s_x[sr[r] * d + i] : 0.0;
                    yy[r] = (double)s_y[sr[r]];
                }
#pragma unroll
                for (int r = 0; r < FROWS; ++r) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double z = bb1[j];
#pragma unroll
                        for (int i = 0; i < DMAX; ++i) z += x[r][i] * w1[i][j];
                        a[r][j] = valid[j] ? fit_tanh(z) : 0.0;
                    }
                    part[r] = a[r][0] * w2[0] + a[r][1] * w2[1];
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
                    for (int r = 0; r < FROWS; ++r) part[r] += __shfl_xor(part[r], o);
                }
                // (after the wave sum every lane holds every row's output: lane r takes row r's, so that exp, the division and log --
                // a third of a row's instructions -- are issued once for the FROWS rows, not once per row; then back to all lanes)
                double p[FROWS], term[FROWS];
                {
                    double my_part = part[0], my_y = yy[0];
#pragma unroll
                    for (int r = 1; r < FROWS; ++r) { my_part = lane == r ? part[r] : my_part; my_y = lane == r ? yy[r] : my_y; }
                    const double my_p = 1.0 / (1.0 + exp(-(b2 + my_part)));
                    const double pc = fmin(fmax(my_p, feps), 1.0 - feps);
                    const double my_term = my_y > 0.0 ? log(pc) : log(1.0 - pc);
#pragma unroll
                    for (int r = 0; r < FROWS; ++r) {
                        p[r] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_p), r), __builtin_amdgcn_readlane(__double2loint(my_p), r));
                        term[r] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_term), r), __builtin_amdgcn_readlane(__double2loint(my_term), r));
                    }
                }
#pragma unroll
                for (int r = 0; r < FROWS; ++r) {
                    if (s + r * GW >= nb) break;
                    loss -= term[r];
                    const double delta = p[r] - yy[r];
                    g_b2 += delta;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        g_w2[j] += a[r][j] * delta;
                        const double dh = delta * w2[j] * (1.0 - a[r][j] * a[r][j]);
                        g_b1[j] += dh;
#pragma unroll
                        for (int i = 0; i < DMAX; ++i) g_w1[i][j] += x[r][i] * dh;
                    }
                }
            }
            // ---- the waves' partial sums -> LDS -> every wave adds them in the same order; in two halves (the room) ----
            double comp[NP];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int i = 0; i < DMAX; ++i) comp[i * 2 + j] = g_w1[i][j];
                comp[2 * DMAX + j] = g_b1[j];
                comp[2 * DMAX + 2 + j] = g_w2[j];
            }
            comp[NC] = loss;
            comp[NC + 1] = g_b2;
            if (G == 1) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int c = 0; c < NPH; ++c) s_part[wave][c][lane] = comp[hf * NPH + c];
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < NPH; ++c) {
                        double t = s_part[0][c][lane];
#pragma unroll
                        for (int w = 1; w < FW; ++w) t += s_part[w][c][lane];
                        comp[hf * NPH + c] = t;
                    }
                    __syncthreads();                  // LDS is rewritten by the other half / the next batch
                }
            } else {
                // ---- the workgroups of the fit: every WAVE writes its sums, the workgroups meet, wave w of every workgroup adds up a
                // quarter of the components over all G x FW waves in their order, the quarters go round through LDS ----
                static_assert(NP % FW == 0 && NP <= FW * NPH, "a share of the components per wave; the totals fit the LDS block");
                constexpr int NPW = NP / FW;
                double *mine = A.xpart + ((((size_t)job * 2 + (step & 1u)) * GW + gw) * NP) * 64;
#pragma unroll
                for (int c = 0; c < NP; ++c) mine[c * 64 + lane] = comp[c];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __syncthreads();
                if (tid == 0) {
                    atomicAdd(&A.bar[job], 1u);
                    const unsigned want = (step + 1u) * (unsigned)G;
                    for (long spin = 0; ; ++spin) {
                        if ((int)(__atomic_load_n(&A.bar[job], __ATOMIC_RELAXED) - want) >= 0) break;
                        if (spin > (1l << 24) || __atomic_load_n(A.failed, __ATOMIC_RELAXED)) { atomicExch(A.failed, 1); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (__atomic_load_n(A.failed, __ATOMIC_RELAXED)) return;
                const double *all = A.xpart + (((size_t)job * 2 + (step & 1u)) * GW) * NP * 64;
                double t[NPW];
#pragma unroll
                for (int c = 0; c < NPW; ++c) t[c] = all[(wave * NPW + c) * 64 + lane];
                for (int w2 = 1; w2 < GW; ++w2) {
#pragma unroll
                    for (int c = 0; c < NPW; ++c) t[c] += all[((size_t)w2 * NP + wave * NPW + c) * 64 + lane];
                }
                double(*s_tot)[64] = reinterpret_cast<double(*)[64]>(s_raw);            // [NP][64], over the waves' block
#pragma unroll
                for (int c = 0; c < NPW; ++c) s_tot[wave * NPW + c][lane] = t[c];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NP; ++c) comp[c] = s_tot[c][lane];
                __syncthreads();                      // (the next batch's totals go to the same place)
                ++step;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int i = 0; i < DMAX; ++i) g_w1[i][j] = comp[i * 2 + j];
                g_b1[j] = comp[2 * DMAX + j];
                g_w2[j] = comp[2 * DMAX + 2 + j];
            }
            loss = comp[NC];
            g_b2 = comp[NC + 1];

            double sq = w2[0] * w2[0] + w2[1] * w2[1];
#pragma unroll
            for (int i = 0; i < DMAX; ++i) sq += w1[i][0] * w1[i][0] + w1[i][1] * w1[i][1];
            const double values = wave_sum(sq);
            const double inv_nb = 1.0 / (double)nb;
            acc += (loss * inv_nb + 0.5 * A.alpha * values * inv_nb) * (double)nb;

            // ---- Adam ----
            b1t *= A.beta1;
            b2t *= A.beta2;
            const double lr_t = A.lr * sqrt(1.0 - b2t) / (1.0 - b1t);
            auto adam = [&](double &p_, double &m_, double &v_, double g) {
                m_ = A.beta1 * m_ + (1.0 - A.beta1) * g;
                v_ = A.beta2 * v_ + (1.0 - A.beta2) * g * g;
                p_ += -lr_t * m_ / (sqrt(v_) + A.eps);
            };
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (valid[j]) {
#pragma unroll
                    for (int i = 0; i < DMAX; ++i)
                        if (i < d) adam(w1[i][j], m_w1[i][j], v_w1[i][j], (g_w1[i][j] + A.alpha * w1[i][j]) * inv_nb);
                    adam(bb1[j], m_b1[j], v_b1[j], g_b1[j] * inv_nb);
                    adam(w2[j], m_w2[j], v_w2[j], (g_w2[j] + A.alpha * w2[j]) * inv_nb);
                }
            }
            adam(b2, m_b2, v_b2, g_b2 * inv_nb);
        }
        const double epoch_loss = acc / (double)n;
        if (tid == 0 && part == 0) A.loss_curve[(size_t)job * A.max_iter + epoch] = epoch_loss;
        n_epochs = epoch + 1;
        key = (uint32_t)(splitmix64((J.seed ^ 0xA5A5A5A55A5A5A5Aull) + (unsigned long long)(epoch + 1)) & 0xFFFFFFFFull);
        if (epoch_loss > best - A.tol) ++no_improve;
        else no_improve = 0;
        if (epoch_loss < best) best = epoch_loss;
        if (no_improve > A.n_iter_no_change) break;
    }

    // ---- results ----
    if (wave == 0 && part == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (valid[j]) {
#pragma unroll
                for (int i = 0; i < DMAX; ++i)
                    if (i < d) A.W1[(size_t)job * d * H + (size_t)i * H + h[j]] = w1[i][j];
                A.b1[(size_t)job * H + h[j]] = bb1[j];
                A.W2[(size_t)job * H + h[j]] = w2[j];
            }
        }
        if (lane == 0) {
            A.b2[job] = b2;
            A.n_iter[job] = n_epochs;
        }
    }
    // held-out rows: accuracy as scikit-learn scores it (predict: p > 0.5)
    long long correct = 0;
    for (long long s = gw; s < J.n_va; s += GW) {
        const long long sid = A.va_idx[J.va_off + s];
        double a[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double z = bb1[j];
#pragma unroll
            for (int i = 0; i < DMAX; ++i) z += (i < d ? A.X[sid * d + i] : 0.0) * w1[i][j];
            a[j] = valid[j] ? fit_tanh(z) : 0.0;
        }
        const double out = b2 + wave_sum(a[0] * w2[0] + a[1] * w2[1]);
        const double p = 1.0 / (1.0 + exp(-out));
        correct += ((p > 0.5) == (A.y[sid] != 0)) ? 1 : 0;
    }
    if (lane == 0 && J.n_va > 0) atomicAdd(reinterpret_cast<unsigned long long *>(&A.val_correct[job]), (unsigned long long)correct);
}

template <typename T>
int to_device(std::vector<void *> &pool, T **dst, const T *src, size_t n, hipStream_t st) {
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n * sizeof(T), 256)) != hipSuccess) {
        mc_set_error("hipMalloc of %zu bytes failed", n * sizeof(T));
        return -10;
    }
    pool.push_back(q);
    *dst = (T *)q;
    if (src && n) {
        if (hipMemcpyAsync(q, src, n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) {
            mc_set_error("H2D copy of %zu bytes failed", n * sizeof(T));
            return -10;
        }
    } else if (n) {
        (void)hipMemsetAsync(q, 0, n * sizeof(T), st);
    }
    return 0;
}

}  // namespace

extern "C" int mc_mlp_fit(mc_ctx *c, const mc_fit_params *P, const double *X, const uint8_t *y, int64_t n_samples, int32_t n_jobs,
                          const int64_t *train_off, const int32_t *train_idx, const int64_t *val_off, const int32_t *val_idx,
                          const uint64_t *seeds, const double *init, double *W1, double *b1, double *W2, double *b2,
                          double *loss_curve, int32_t *n_iter, int64_t *val_correct) {
    HIP_TRY(hipSetDevice(mc_internal_device(c)));
    hipStream_t st = mc_internal_stream(c);
    if (!P || P->n_in < 1 || P->n_in > DMAX || P->n_hidden < 1 || P->n_hidden > HMAX || P->batch_size < 1 || P->max_iter < 1 ||
        n_jobs < 1 || n_samples < 1) {
        mc_set_error("mc_mlp_fit: unsupported shape (inputs 1..%d, hidden 1..%d)", DMAX, HMAX);
        return -12;
    }
    const size_t lds = sizeof(double) * FW * NPH * 64 + (size_t)P->batch_size * P->n_in * 8 + (size_t)P->batch_size;
    if (P->batch_size > FT || lds > 64 * 1024) {
        mc_set_error("mc_mlp_fit: batch size %d does not fit (max %d rows)", P->batch_size, FT);
        return -12;
    }
    for (int j = 0; j < n_jobs; ++j) {
        if (train_off[j + 1] < train_off[j] || val_off[j + 1] < val_off[j] || train_off[j + 1] - train_off[j] > (int64_t)1 << 31) {
            mc_set_error("mc_mlp_fit: bad offsets for job %d", j);
            return -12;
        }
        for (int64_t i = train_off[j]; i < train_off[j + 1]; ++i)
            if (train_idx[i] < 0 || train_idx[i] >= n_samples) { mc_set_error("mc_mlp_fit: row index out of range"); return -12; }
        for (int64_t i = val_off[j]; i < val_off[j + 1]; ++i)
            if (val_idx[i] < 0 || val_idx[i] >= n_samples) { mc_set_error("mc_mlp_fit: row index out of range"); return -12; }
    }
    const int d = P->n_in, H = P->n_hidden;
    std::vector<FitJob> jobs((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        jobs[(size_t)j].tr_off = train_off[j];
        jobs[(size_t)j].n_tr = train_off[j + 1] - train_off[j];
        jobs[(size_t)j].va_off = val_off[j];
        jobs[(size_t)j].n_va = val_off[j + 1] - val_off[j];
        jobs[(size_t)j].seed = seeds ? seeds[j] : (uint64_t)(P->seed + (uint64_t)j);
    }
    std::vector<void *> pool;
    auto cleanup = [&]() { for (void *p : pool) (void)hipFree(p); };
    FitArgs A;
    double *dX, *dinit = nullptr, *dW1, *db1, *dW2, *db2, *dcurve;
    uint8_t *dy;
    FitJob *djobs;
    int32_t *dtr, *dva, *dnit;
    long long *dcorrect;
    double *dxpart;
    unsigned *dbar;
    int *dfailed;
    // workgroups per fit: four (the rows of a batch of 200 over 16 waves), one for small batches or on request
    static const int groups_env = getenv("MCALLER_FIT_WGS") ? atoi(getenv("MCALLER_FIT_WGS")) : 0;
    int G = std::max(1, std::min(16, groups_env > 0 ? groups_env : (P->batch_size >= 64 ? MC_FIT_GROUPS : 1)));
    // The workgroups of a fit meet once per batch by spinning on a counter: every one of them has to be resident for the others to
    // get past the meeting.  More workgroups than the device holds at once (many jobs in one call) -> one workgroup per fit, which
    // waits for nobody.
    if (G > 1) {
        int per_cu = 0, n_cu = 0, dev_id = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev_id) == hipSuccess && hipGetDeviceProperties(&prop, dev_id) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k4_mlp_fit, FT, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        const int64_t wgs = (n_jobs <= 8 ? 8 : (int64_t)n_jobs) * G;            // (by_xcd launches 8 x G)
        if (per_cu <= 0 || n_cu <= 0 || wgs > (int64_t)per_cu * n_cu) G = 1;
    }
    const size_t per_job = (size_t)d * H + 2 * (size_t)H + 1;
    int rc = 0;
    rc |= to_device(pool, &dX, X, (size_t)n_samples * d, st);
    rc |= to_device(pool, &dy, y, (size_t)n_samples, st);
    rc |= to_device(pool, &djobs, jobs.data(), (size_t)n_jobs, st);
    rc |= to_device(pool, &dtr, train_idx, (size_t)std::max<int64_t>(train_off[n_jobs], 1), st);
    rc |= to_device(pool, &dva, val_idx, (size_t)std::max<int64_t>(val_off[n_jobs], 1), st);
    if (init) rc |= to_device(pool, &dinit, init, per_job * n_jobs, st);
    rc |= to_device<double>(pool, &dW1, nullptr, (size_t)n_jobs * d * H, st);
    rc |= to_device<double>(pool, &db1, nullptr, (size_t)n_jobs * H, st);
    rc |= to_device<double>(pool, &dW2, nullptr, (size_t)n_jobs * H, st);
    rc |= to_device<double>(pool, &db2, nullptr, (size_t)n_jobs, st);
    rc |= to_device<double>(pool, &dcurve, nullptr, (size_t)n_jobs * P->max_iter, st);
    rc |= to_device<int32_t>(pool, &dnit, nullptr, (size_t)n_jobs, st);
    rc |= to_device<long long>(pool, &dcorrect, nullptr, (size_t)n_jobs, st);
    rc |= to_device<double>(pool, &dxpart, nullptr, (size_t)n_jobs * 2 * G * FW * NP * 64, st);
    rc |= to_device<unsigned>(pool, &dbar, nullptr, (size_t)n_jobs, st);
    rc |= to_device<int>(pool, &dfailed, nullptr, 1, st);
    if (rc) { cleanup(); return -10; }
    A.X = dX; A.y = dy; A.jobs = djobs; A.tr_idx = dtr; A.va_idx = dva;
    A.d = d; A.H = H; A.batch = P->batch_size; A.max_iter = P->max_iter; A.n_iter_no_change = P->n_iter_no_change;
    A.shuffle = P->shuffle;
    A.alpha = P->alpha; A.lr = P->lr_init; A.beta1 = P->beta1; A.beta2 = P->beta2; A.eps = P->epsilon; A.tol = P->tol;
    A.G = G; A.n_jobs = n_jobs; A.by_xcd = (G > 1 && n_jobs <= 8) ? 1 : 0; A.xpart = dxpart; A.bar = dbar; A.failed = dfailed;
    A.init = dinit; A.W1 = dW1; A.b1 = db1; A.W2 = dW2; A.b2 = db2; A.loss_curve = dcurve; A.n_iter = dnit; A.val_correct = dcorrect;
    hipLaunchKernelGGL(k4_mlp_fit, dim3((unsigned)(A.by_xcd ? 8 * G : n_jobs * G)), dim3(FT), lds, st, A);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(W1, dW1, (size_t)n_jobs * d * H * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b1, db1, (size_t)n_jobs * H * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(W2, dW2, (size_t)n_jobs * H * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b2, db2, (size_t)n_jobs * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(loss_curve, dcurve, (size_t)n_jobs * P->max_iter * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(n_iter, dnit, (size_t)n_jobs * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(val_correct, dcorrect, (size_t)n_jobs * 8, hipMemcpyDeviceToHost, st);
    int fit_failed = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&fit_failed, dfailed, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) {
        mc_set_error("mc_mlp_fit failed: %s", hipGetErrorString(e));
        return -10;
    }
    if (fit_failed) {
        mc_set_error("mc_mlp_fit: the %d workgroups of a fit did not meet (MCALLER_FIT_WGS=1 runs a fit in one workgroup)", G);
        return -10;
    }
    return 0;
}
