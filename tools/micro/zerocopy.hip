// Microbenchmark: a kernel that writes N MB into pinned host memory (zero-copy stores) vs hipMemcpyAsync D2H.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_export(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t bytes = 16u << 20;
    void *d, *h, *hd;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer(&hd, h, 0));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int grid : {64, 256, 1024, 4096}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(k_export, dim3(grid), dim3(256), 0, st, (const uint4 *)d, (uint4 *)hd, bytes / 16);
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("kernel export grid %5d: %.3f ms  %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, st));
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("hipMemcpyAsync D2H:        %.3f ms  %.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    printf("first bytes on host: %d\n", ((unsigned char *)h)[12345]);
    return 0;
}
