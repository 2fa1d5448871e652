// The scan's load pattern in isolation: persistent one-wave workgroups; each reads a tile (NQ KB of positions + NQ/4 KB of
// flags) into registers with 16-byte loads, writes it to LDS, issues the next tile's loads, "works" on the LDS tile for
// WORK dependent LDS round trips, takes the next tile (grid-stride).  MODE 0: as described; 1: no barriers; 2: registers
// consumed by XORs instead of LDS stores.  What bandwidth can this structure reach, and what does the staging cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int NQ, int MODE, int WORK>
__global__ __launch_bounds__(64) void k_burst(const uint4 *__restrict__ pos, const uint4 *__restrict__ fl, long n_tiles, unsigned *out) {
    __shared__ uint4 s_pos[NQ * 64];
    __shared__ uint4 s_fl[(NQ / 4) * 64 + 1];
    const int tid = threadIdx.x;
    uint4 r[NQ], f[NQ / 4];
    unsigned acc = 0;
    long tile = blockIdx.x;
    {
        const long t0 = tile < n_tiles ? tile : n_tiles - 1;          // (unconditional loads: the arrays stay in registers)
#pragma unroll
        for (int j = 0; j < NQ; ++j) r[j] = pos[(t0 * NQ + j) * 64 + tid];
#pragma unroll
        for (int j = 0; j < NQ / 4; ++j) f[j] = fl[(t0 * (NQ / 4) + j) * 64 + tid];
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) s_pos[j * 64 + tid] = r[j];
#pragma unroll
            for (int j = 0; j < NQ / 4; ++j) s_fl[j * 64 + tid] = f[j];
        } else {
#pragma unroll
            for (int j = 0; j < NQ; ++j) acc ^= r[j].x ^ r[j].w;
#pragma unroll
            for (int j = 0; j < NQ / 4; ++j) acc ^= f[j].x;
        }
        if (MODE == 0) __syncthreads();
        {
            const long nx = tile + gridDim.x, nxt = nx < n_tiles ? nx : tile;   // (the last round re-reads its own tile)
#pragma unroll
            for (int j = 0; j < NQ; ++j) r[j] = pos[(nxt * NQ + j) * 64 + tid];
#pragma unroll
            for (int j = 0; j < NQ / 4; ++j) f[j] = fl[(nxt * (NQ / 4) + j) * 64 + tid];
        }
        if (MODE != 2) {
            unsigned idx = (tid * 7) % (NQ * 64);
            for (int w = 0; w < WORK; ++w) {                       // dependent LDS reads: the "walk"
                const unsigned v = s_pos[idx].x;
                acc ^= v;
                idx = (idx + (v & 63) + 1) % (NQ * 64);
            }
        }
        if (MODE == 0) __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int NQ, int MODE, int WORK>
void run(const void *a, const void *b, size_t rows, unsigned *out, int wgs_per_cu) {
    const long n_tiles = (long)(rows / (NQ * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_burst<NQ, MODE, WORK>), dim3(256 * wgs_per_cu), dim3(64), 0, 0, (const uint4 *)a, (const uint4 *)b, n_tiles, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_burst<NQ, MODE, WORK>, 64, 0);
    printf("mode %d work %3d tile %5d rows, %2d WG/CU launched (occupancy %2d): %.4f ms  %.2f TB/s\n", MODE, WORK, NQ * 256, wgs_per_cu, occ,
           best, n_tiles * NQ * 256 * 5.0 / best / 1e9);
}
int main() {
    const size_t rows = 100000000;
    void *a, *b; unsigned *out;
    CK(hipMalloc(&a, rows * 4 + (1 << 20))); CK(hipMalloc(&b, rows + (1 << 20))); CK(hipMalloc((void **)&out, 4));
    CK(hipMemset(a, 1, rows * 4)); CK(hipMemset(b, 2, rows));
    run<12, 2, 0>(a, b, rows, out, 8);
    run<12, 0, 0>(a, b, rows, out, 8);
    run<12, 1, 0>(a, b, rows, out, 8);
    run<12, 0, 16>(a, b, rows, out, 8);
    run<12, 0, 64>(a, b, rows, out, 8);
    run<12, 0, 128>(a, b, rows, out, 8);
    run<8, 0, 0>(a, b, rows, out, 8);
    run<8, 0, 0>(a, b, rows, out, 12);
    run<8, 0, 64>(a, b, rows, out, 12);
    run<4, 0, 0>(a, b, rows, out, 16);
    run<4, 0, 32>(a, b, rows, out, 16);
    run<16, 0, 0>(a, b, rows, out, 4);
    run<16, 0, 0>(a, b, rows, out, 7);
    return 0;
}
