// How many vector / scalar instructions does a CU issue per clock?  The budget of the instruction-bound kernels (k1_emit_runs,
// k1_scan<130>): N independent chains per lane, 8 waves per SIMD, every CU busy; wave-instructions per clock per CU at the clock
// the run reports (wall_clock64 is 100 MHz: the rate is given per microsecond and per clock of an assumed 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 2048, CH = 8;
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(unsigned *out, unsigned seed) {
    unsigned a[CH];
    double d[CH];
    unsigned long long q[CH];
    float f[CH];
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 g[CH];
    f4 acc4[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { a[c] = seed + threadIdx.x * (c + 1); d[c] = (double)a[c]; q[c] = a[c]; f[c] = (float)(a[c] & 1023u) * 1e-3f; g[c] = (f2){f[c], f[c] + 1.0f}; acc4[c] = (f4){f[c], 0.f, 0.f, 0.f}; }
    unsigned s = seed;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (KIND == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (KIND == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
            if (KIND == 4) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");
            if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(seed) : );
            if (KIND == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (KIND == 8) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[c]), "v"(seed) : "vcc");
            if (KIND == 9) asm volatile("v_readlane_b32 %0, %1, 3" : "+s"(s) : "v"(a[c]));
            if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CH]));
            if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(g[c]) : "v"(g[(c + 1) % CH]));
            if (KIND == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
            if (KIND == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
            if (KIND == 14) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(g[c]) : "v"(g[(c + 1) % CH]));
            if (KIND == 15) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[c]) : "v"(seed));
            if (KIND == 16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[c]) : "v"(f[c]), "v"(f[(c + 1) % CH]));
            if (KIND == 17) {           // the mix of the fast forward's hidden layer: 8 packed + 2 exp + 2 rcp per pair of units
                if (c < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
                else if (c < 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(g[c]) : "v"(g[(c + 1) % CH]));
            }
            if (KIND == 18) {           // MFMA from the odd waves, packed fma from the even ones: do the two pipes overlap?
                if ((threadIdx.x >> 6) & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[c]) : "v"(f[c]), "v"(f[(c + 1) % CH]));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(g[c]) : "v"(g[(c + 1) % CH]));
            }
        }
    }
    unsigned acc = s;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c] ^ (unsigned)d[c] ^ (unsigned)q[c] ^ __float_as_uint(f[c]) ^ __float_as_uint(g[c].x + g[c].y) ^ __float_as_uint(acc4[c].x + acc4[c].y + acc4[c].z + acc4[c].w);
    if (acc == 0x12345678u) out[0] = acc;
}
template <int KIND>
void run(const char *name, unsigned *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, 7u + rep);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double wave_instr = (double)grid * 4 * ITER * CH;
    printf("%-16s %.4f ms  %.1f wave-instructions / us / CU = %.2f per clock per CU at 2.4 GHz\n", name, best, wave_instr / 256 / (best * 1e3),
           wave_instr / 256 / (best * 1e-3 * 2.4e9));
}
int main() {
    unsigned *out; CK(hipMalloc((void **)&out, 4));
    run<0>("v_add_u32", out); run<1>("v_xor_b32", out); run<5>("v_cndmask_b32", out); run<8>("v_cmp_lt_u32", out); run<6>("v_mul_lo_u32", out);
    run<3>("v_lshl_add_u64", out); run<2>("v_add_f64", out); run<7>("v_fma_f64", out); run<9>("v_readlane_b32", out); run<4>("s_add_u32", out);
    run<10>("v_fma_f32", out); run<11>("v_pk_fma_f32", out); run<14>("v_pk_mul_f32", out); run<12>("v_exp_f32", out); run<13>("v_rcp_f32", out);
    run<15>("v_bfi_b32", out); run<16>("v_mfma_f32_16x16x4_f32", out); run<17>("mix: 2 exp 2 rcp 8 pk_fma (12 counted as 8)", out); run<18>("half mfma, half pk_fma", out);
    return 0;
}
