// How fast can 0.4 GB (int32 column) + 0.1 GB (byte column) be streamed through the CUs at all?  Plain read kernels
// (grid-stride, 16 B per lane, XOR-reduced so nothing is optimised away) at several grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_read2(const uint4 *__restrict__ a, size_t na, const uint4 *__restrict__ b, size_t nb, unsigned *out) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = t; i < na; i += stride) { const uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    for (size_t i = t; i < nb; i += stride) { const uint4 v = b[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t rows = 100000000;
    void *a, *b; unsigned *out;
    CK(hipMalloc(&a, rows * 4)); CK(hipMalloc(&b, rows)); CK(hipMalloc((void **)&out, 4));
    CK(hipMemset(a, 1, rows * 4)); CK(hipMemset(b, 2, rows));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_read2, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, rows * 4 / 16, (const uint4 *)b, rows / 16, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("grid %5d x 256: %.4f ms  %.2f TB/s\n", grid, best, rows * 5.0 / best / 1e9);
    }
    return 0;
}
