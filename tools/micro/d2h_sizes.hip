// hipMemcpyAsync D2H into pinned memory at several sizes (GB/s): is the rate size-independent?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mb : {16, 64, 256, 750, 1500}) {
        const size_t bytes = mb << 20;
        void *d, *h;
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(d, 1, bytes));
        CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a, st));
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("D2H %5zu MB rep %d: %.3f ms  %.1f GB/s\n", mb, rep, ms, bytes / ms / 1e6);
        }
        CK(hipFree(d)); CK(hipHostFree(h));
    }
    return 0;
}
