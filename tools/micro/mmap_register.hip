// Microbenchmark: two ways to get a file's bytes (page cache warm) to the GPU, window by window, at T host threads --
//   A. pread into pinned memory (hipHostMalloc'ed once), then hipMemcpyAsync H2D: what the streamed reader does (mc_read_file_range);
//      a CPU copy per byte;
//   B. mmap the window (MAP_SHARED | MAP_POPULATE), hipHostRegister it, hipMemcpyAsync H2D straight out of the page cache,
//      hipHostUnregister, munmap: no CPU copy per byte, but the registration pins and maps every 4 KB page.
// Both are run one window after the other without overlap (what a byte costs, not what a pipeline hides) and, B, with the
// registration of the next window done by a second thread while the current one is copied (what a pipeline would see).
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/micro/mmap_register.hip -o /tmp/mmap_register && /tmp/mmap_register FILE [window MB] [threads] [max GB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    // usage: mmap_register [FILE [window MB] [threads] [max GB]]; without a file: 1 GB of zeros written to /tmp first
    char own[64] = "";
    if (argc < 2) {
        snprintf(own, sizeof own, "/tmp/mc_mmap_register_%d.bin", (int)getpid());
        FILE *f = fopen(own, "wb");
        if (!f) { printf("cannot write %s\n", own); return 1; }
        std::vector<char> block(16 << 20, 1);
        for (int i = 0; i < 64; ++i) fwrite(block.data(), 1, block.size(), f);
        fclose(f);
    }
    const char *path = argc < 2 ? own : argv[1];
    const size_t window = (size_t)(argc > 2 ? atoi(argv[2]) : 128) << 20;
    const int T = argc > 3 ? atoi(argv[3]) : 2;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }
    struct stat sb;
    fstat(fd, &sb);
    size_t total = (size_t)sb.st_size;
    if (argc > 4) total = std::min(total, (size_t)(atof(argv[4]) * 1e9));
    total -= total % 4096;
    const size_t n_win = total / window;
    if (n_win < 2) { printf("file too small\n"); return 1; }
    void *dev;
    CK(hipMalloc(&dev, window));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    {   // warm the page cache
        std::vector<char> buf(8 << 20);
        for (size_t off = 0; off < n_win * window; off += buf.size()) (void)!pread(fd, buf.data(), buf.size(), (off_t)off);
    }
    // ---- A: pread into pinned memory + H2D ----
    void *pin;
    CK(hipHostMalloc(&pin, window, hipHostMallocDefault));
    double a_read = 0, a_copy = 0;
    for (size_t w = 0; w < n_win; ++w) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const size_t lo = window * (size_t)t / (size_t)T, hi = window * (size_t)(t + 1) / (size_t)T;
                for (size_t off = lo; off < hi;) {
                    const ssize_t got = pread(fd, (char *)pin + off, std::min<size_t>(hi - off, 8 << 20), (off_t)(w * window + off));
                    if (got <= 0) break;
                    off += (size_t)got;
                }
            });
        for (auto &x : th) x.join();
        double t1 = now();
        CK(hipMemcpyAsync(dev, pin, window, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        a_read += t1 - t0; a_copy += t2 - t1;
    }
    const double gb = (double)(n_win * window) / 1e9;
    printf("{\"window_MB\": %zu, \"threads\": %d, \"GB\": %.2f,\n", window >> 20, T, gb);
    printf(" \"pread_into_pinned\": {\"read_s\": %.4f, \"read_GBps\": %.2f, \"h2d_s\": %.4f, \"h2d_GBps\": %.2f, \"serial_GBps\": %.2f},\n", a_read, gb / a_read,
           a_copy, gb / a_copy, gb / (a_read + a_copy));
    // ---- B: mmap + hipHostRegister + H2D ----
    double b_map = 0, b_reg = 0, b_copy = 0, b_unreg = 0;
    for (size_t w = 0; w < n_win; ++w) {
        double t0 = now();
        void *m = mmap(nullptr, window, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, (off_t)(w * window));
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        double t1 = now();
        hipError_t e = hipHostRegister(m, window, hipHostRegisterDefault);
        if (e != hipSuccess) { printf(" \"mmap_register\": {\"error\": \"hipHostRegister: %s\"}}\n", hipGetErrorString(e)); return 0; }
        double t2 = now();
        CK(hipMemcpyAsync(dev, m, window, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(m));
        munmap(m, window);
        double t4 = now();
        b_map += t1 - t0; b_reg += t2 - t1; b_copy += t3 - t2; b_unreg += t4 - t3;
    }
    printf(" \"mmap_register\": {\"mmap_populate_s\": %.4f, \"register_s\": %.4f, \"register_us_per_page\": %.3f, \"h2d_s\": %.4f, \"h2d_GBps\": %.2f, "
           "\"unregister_munmap_s\": %.4f, \"serial_GBps\": %.2f},\n", b_map, b_reg, b_reg * 1e6 / ((double)(n_win * window) / 4096.0), b_copy, gb / b_copy,
           b_unreg, gb / (b_map + b_reg + b_copy + b_unreg));
    // ---- B': the next window mapped and registered by another thread while this one travels ----
    {
        std::vector<void *> maps(n_win, nullptr);
        double t0 = now();
        maps[0] = mmap(nullptr, window, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        CK(hipHostRegister(maps[0], window, hipHostRegisterDefault));
        for (size_t w = 0; w < n_win; ++w) {
            std::thread ahead([&, w] {
                if (w + 1 < n_win) {
                    (void)hipSetDevice(0);
                    maps[w + 1] = mmap(nullptr, window, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, (off_t)((w + 1) * window));
                    (void)hipHostRegister(maps[w + 1], window, hipHostRegisterDefault);
                }
            });
            CK(hipMemcpyAsync(dev, maps[w], window, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            ahead.join();
            CK(hipHostUnregister(maps[w]));
            munmap(maps[w], window);
        }
        const double dt = now() - t0;
        printf(" \"mmap_register_one_window_ahead\": {\"seconds\": %.4f, \"GBps\": %.2f},\n", dt, gb / dt);
    }
    // ---- A': pread of the next window by T threads while this one travels (two pinned buffers) ----
    {
        void *pin2;
        CK(hipHostMalloc(&pin2, window, hipHostMallocDefault));
        void *bufs[2] = {pin, pin2};
        auto read_win = [&](size_t w, void *dst) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    const size_t lo = window * (size_t)t / (size_t)T, hi = window * (size_t)(t + 1) / (size_t)T;
                    for (size_t off = lo; off < hi;) {
                        const ssize_t got = pread(fd, (char *)dst + off, std::min<size_t>(hi - off, 8 << 20), (off_t)(w * window + off));
                        if (got <= 0) break;
                        off += (size_t)got;
                    }
                });
            for (auto &x : th) x.join();
        };
        double t0 = now();
        read_win(0, bufs[0]);
        for (size_t w = 0; w < n_win; ++w) {
            std::thread ahead([&, w] { if (w + 1 < n_win) read_win(w + 1, bufs[(w + 1) & 1]); });
            CK(hipMemcpyAsync(dev, bufs[w & 1], window, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            ahead.join();
        }
        const double dt = now() - t0;
        printf(" \"pread_one_window_ahead\": {\"seconds\": %.4f, \"GBps\": %.2f}}\n", dt, gb / dt);
    }
    if (own[0]) unlink(own);
    return 0;
}
