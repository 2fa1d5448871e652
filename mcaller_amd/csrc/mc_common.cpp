// Error text, version, host core count and the pinned host buffer pool of libmcaller_hip.so (C ABI: include/mcaller_hip.h).
#include "../../include/mcaller_hip.h"

#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <mutex>
#include <vector>

static thread_local char g_err[1024] = "";

void mc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mc_last_error(void) { return g_err; }
extern "C" const char *mc_version(void) { return "mcaller_hip 0.2 (gfx950)"; }

// Cores this process may USE: the CPUs of its affinity mask (a worker that bound itself to the NUMA node of its GPU,
// mc_bind_to_device_numa_node, or a job inside a cpuset), capped by the CPU-time quota of its control group (cpu.max: a
// container that is granted 16 CPUs' worth of time per 100 ms on a 256-CPU host) -- and by $MCALLER_HOST_CORES (the workers of a
// multi-GPU run share one quota).  Parser, reader, formatter and FASTQ threads are that many: with 64 threads runnable under a
// quota of 16 the group burns its period's allowance in 25 ms and EVERY thread of the process -- the one that feeds the GPU
// included -- is stopped for the other 75 (file to file at 10^8 rows: 0.5 s on a busy box instead of 0.3).
static int affinity_cpus(void) {
    const int max_cpus = 8192;
    cpu_set_t *set = CPU_ALLOC(max_cpus);
    int n = 0;
    if (set) {
        const size_t bytes = CPU_ALLOC_SIZE(max_cpus);
        CPU_ZERO_S(bytes, set);
        if (sched_getaffinity(0, bytes, set) == 0) n = CPU_COUNT_S(bytes, set);
        CPU_FREE(set);
    }
    return n < 1 ? 1 : n;
}

static int quota_of(const char *path) {          // CPUs' worth of time in one cpu.max ("<quota> <period>" | "max <period>"), 0: none
    long long quota = 0, period = 0;
    if (FILE *f = fopen(path, "r")) {
        char q[64] = "";
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)std::max<long long>(1, (quota + period - 1) / period);
}

static int cgroup_quota_cpus(void) {           // -> CPUs' worth of time the control groups grant, 0: unlimited / unknown
    // cgroup v2: the process's own group and every ancestor up to the root the container sees (a quota set on a systemd slice
    // or on a pod applies to everything below it): the smallest one counts
    int best = 0;
    auto take = [&](int q) { if (q > 0 && (best == 0 || q < best)) best = q; };
    char own[512] = "";
    if (FILE *f = fopen("/proc/self/cgroup", "r")) {
        char line[512];
        while (fgets(line, sizeof(line), f))
            if (strncmp(line, "0::", 3) == 0) {
                char *nl = strchr(line, '\n');
                if (nl) *nl = 0;
                snprintf(own, sizeof(own), "%s", line + 3);
            }
        fclose(f);
    }
    bool v2 = false;
    if (FILE *g = fopen("/sys/fs/cgroup/cpu.max", "r")) { fclose(g); v2 = true; }
    for (;;) {                                   // "/a/b/c" -> "/a/b" -> "/a" -> ""
        char path[1100];
        snprintf(path, sizeof(path), "/sys/fs/cgroup%s/cpu.max", own);
        if (FILE *g = fopen(path, "r")) { fclose(g); v2 = true; take(quota_of(path)); }
        char *slash = strrchr(own, '/');
        if (!slash) break;
        *slash = 0;
    }
    if (!v2) {                                   // cgroup v1
        long long quota = 0, period = 0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = 0; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) take((int)std::max<long long>(1, (quota + period - 1) / period));
    }
    return best;
}

extern "C" int mc_host_cores(void) {
    int n = affinity_cpus();
    static const int quota = cgroup_quota_cpus();
    if (quota > 0) n = std::min(n, quota);
    if (const char *e = getenv("MCALLER_HOST_CORES")) { if (atoi(e) > 0) n = std::min(n, atoi(e)); }
    return n < 1 ? 1 : n;
}

// ---------------------------------------------------------------------------------------------------
// Pinned host memory, recycled.  A table that is streamed to the GPU (mc_ctx_upload_table_async) has to sit in pinned
// memory for the DMA engines to read it at PCIe speed and without a staging copy; pinning is expensive (page-table work
// per 4 KB), so blocks are kept and handed out again: after the first two or three shards of a file no allocation happens.
// Without a GPU (CPU-only test runs) the blocks are plain aligned memory.
// ---------------------------------------------------------------------------------------------------
namespace {
struct Block {
    void *p;
    size_t bytes;
    bool pinned, busy;
};
std::mutex g_pool_mu;
std::vector<Block> g_pool;
size_t g_keep_bytes = (size_t)8 << 30;
int g_parser_uses_pool = 0;

void release_block(Block &b) {
    if (b.pinned) (void)hipHostFree(b.p);
    else free(b.p);
}
}  // namespace

int mc_parser_uses_pool() { return g_parser_uses_pool; }

extern "C" void *mc_host_alloc(int64_t bytes_in) {
    const size_t bytes = (size_t)std::max<int64_t>(bytes_in, 1);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (size_t i = 0; i < g_pool.size(); ++i) {
            const Block &b = g_pool[i];
            if (!b.busy && b.bytes >= bytes && (best < 0 || b.bytes < g_pool[(size_t)best].bytes)) best = (int)i;
        }
        if (best >= 0 && g_pool[(size_t)best].bytes <= 2 * bytes + ((size_t)4 << 20)) {
            g_pool[(size_t)best].busy = true;
            return g_pool[(size_t)best].p;
        }
    }
    // a new block, with head room for the next table of about this size; 2 MB granules
    size_t want = bytes + bytes / 8;
    want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    Block nb{nullptr, want, true, true};
    if (hipHostMalloc(&nb.p, want, hipHostMallocDefault) != hipSuccess || !nb.p) {
        (void)hipGetLastError();
        nb.pinned = false;
        nb.p = aligned_alloc(4096, want);
        if (!nb.p) return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(nb);
    return nb.p;
}

extern "C" void mc_host_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t idle = 0;
    int at = -1;
    for (size_t i = 0; i < g_pool.size(); ++i) {
        if (g_pool[i].p == p) at = (int)i;
        else if (!g_pool[i].busy) idle += g_pool[i].bytes;
    }
    if (at < 0) return;                                  // not ours
    Block &b = g_pool[(size_t)at];
    b.busy = false;
    if (idle + b.bytes > g_keep_bytes) {                 // more idle memory than we were told to keep: give it back
        release_block(b);
        g_pool.erase(g_pool.begin() + at);
    }
}

extern "C" int mc_host_is_pinned(const void *p) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (const Block &b : g_pool)
        if ((const char *)p >= (const char *)b.p && (const char *)p < (const char *)b.p + b.bytes) return b.pinned ? 1 : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// A byte range of a file where the DMA engines can read it WITHOUT a copy by the CPU: the range mapped (MAP_SHARED: the page
// cache's own pages) and registered with the runtime.  What the streamed reader did until round 5 -- pread into pinned memory, a
// CPU copy per byte -- reaches 30 GB/s on the two host threads a worker of an eight-GPU run has (38 on sixteen); the mapped range
// travels at the link's 57 GB/s, and mapping + registering + releasing 128 MB costs 0.7 ms (tools/micro/mmap_register.hip,
// NOTES.md section 14: "ruled out on paper" in round 4 at a microsecond per page, measured at 0.011).
// ---------------------------------------------------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
namespace {
struct MapRec { void *base; size_t len; };
}
extern "C" int mc_map_file_range(const char *path, int64_t lo, int64_t hi, void **handle, const char **ptr) {
    if (!path || !handle || !ptr || lo < 0 || hi <= lo) {
        mc_set_error("mc_map_file_range: bad arguments");
        return -12;
    }
    *handle = nullptr;
    *ptr = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        mc_set_error("cannot open %s: %s", path, strerror(errno));
        return -1;
    }
    const int64_t page = sysconf(_SC_PAGESIZE), a0 = lo & ~(page - 1);
    const size_t len = (size_t)(hi - a0);
    void *base = mmap(nullptr, len, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, (off_t)a0);
    close(fd);
    if (base == MAP_FAILED) {
        mc_set_error("mmap of %s failed: %s", path, strerror(errno));
        return -1;
    }
    if (hipHostRegister(base, len, hipHostRegisterPortable) != hipSuccess) {        // (no GPU, or the runtime refuses file pages: the caller reads)
        (void)hipGetLastError();
        munmap(base, len);
        mc_set_error("hipHostRegister of %zu mapped bytes of %s failed", len, path);
        return -10;
    }
    *handle = new MapRec{base, len};
    *ptr = (const char *)base + (lo - a0);
    return 0;
}

extern "C" void mc_unmap_file_range(void *handle) {
    MapRec *m = (MapRec *)handle;
    if (!m) return;
    (void)hipHostUnregister(m->base);
    munmap(m->base, m->len);
    delete m;
}

extern "C" int mc_host_pool_config(int32_t parser_uses_pool, int64_t keep_bytes) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_parser_uses_pool = parser_uses_pool ? 1 : 0;
    if (keep_bytes >= 0) g_keep_bytes = (size_t)keep_bytes;
    size_t idle = 0;
    for (size_t i = 0; i < g_pool.size();) {
        if (!g_pool[i].busy && idle + g_pool[i].bytes > g_keep_bytes) {
            release_block(g_pool[i]);
            g_pool.erase(g_pool.begin() + (long)i);
            continue;
        }
        if (!g_pool[i].busy) idle += g_pool[i].bytes;
        ++i;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Marking a contig's motif sites (extract_contexts.py:33-41,:60-73 for a motif): seq.upper(), then str.replace(motif, repl)
// for the motif and its reverse complement -- left to right, non-overlapping, len(repl) == len(motif).  CPython spends 14 ms
// on E. coli with the interpreter lock held, at the start of every streamed file, while the threads that feed the GPU wait
// for the lock; here the two strands are done side by side without it.  ASCII only (the caller checks).
// ---------------------------------------------------------------------------------------------------
void mc_parallel_for(int n, const std::function<void(int)> &f);

static void replace_equal(const char *seq, int64_t n, const char *motif, const char *repl, int32_t m, char *out) {
    memcpy(out, seq, (size_t)n);
    if (m <= 0 || m > n) return;
    if (m == 1) {
        const char c0 = motif[0], r0 = repl[0];
        for (int64_t i = 0; i < n; ++i) out[i] = seq[i] == c0 ? r0 : seq[i];
        return;
    }
    const char *p = seq, *end = seq + n;
    while (p + m <= end) {
        const char *hit = (const char *)memmem(p, (size_t)(end - p), motif, (size_t)m);
        if (!hit) break;
        memcpy(out + (hit - seq), repl, (size_t)m);
        p = hit + m;
    }
}

extern "C" int mc_mark_motifs(const char *seq, int64_t n, const char *motif_fwd, const char *repl_fwd, int32_t m_fwd,
                              const char *motif_rev, const char *repl_rev, int32_t m_rev, char *upper_out, char *fwd_out,
                              char *rev_out) {
    if (!seq || n < 0 || !upper_out || !fwd_out || !rev_out || m_fwd < 0 || m_rev < 0 || (m_fwd > 0 && (!motif_fwd || !repl_fwd)) ||
        (m_rev > 0 && (!motif_rev || !repl_rev))) {
        mc_set_error("mc_mark_motifs: bad arguments");
        return -12;
    }
    for (int64_t i = 0; i < n; ++i) {
        const char ch = seq[i];
        upper_out[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch;
    }
    mc_parallel_for(2, [&](int strand) {
        if (strand == 0) replace_equal(upper_out, n, motif_fwd, repl_fwd, m_fwd, fwd_out);
        else replace_equal(upper_out, n, motif_rev, repl_rev, m_rev, rev_out);
    });
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Worker threads, kept and PINNED.  The parser, the FASTQ reader and the row formatter run one task per piece of their input;
// a file streamed in shards calls them once per shard, and starting a few dozen std::threads per call (one at a time, by the
// caller) was half a shard's parse time.  mc_parallel_for hands the tasks to threads that stay around: one per CPU the
// process may run on, each bound to its CPU.  The binding is the point: threads that are merely woken (condition variable,
// or freshly started) land where the waker runs -- the scheduler stacks them on a few cores and spreads them out over the
// next tens of milliseconds; measured on the 2 x 64-core host, 64 equal pieces of a 1.17 GB file took 13 ms (the fastest) to
// 95 ms (the slowest) each, and the slowest one is the parse time.  Several callers at once (two parser threads, the
// formatter beside them) share the workers: a worker takes the next task of whichever job has one.  The caller works too.
// ---------------------------------------------------------------------------------------------------
#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>
#include <pthread.h>

namespace {
struct Job {
    const std::function<void(int)> *fn;
    int n;
    std::atomic<int> next{0};
    std::atomic<int> finished{0};
    int visitors = 0;                       // workers that picked this job and have not let go of it yet (guarded by Workers::mu)
};

struct Workers {
    std::mutex mu;                          // guards jobs, n_threads; the condition variables
    std::condition_variable wake, done;
    std::vector<Job *> jobs;                // jobs that may still have tasks to hand out
    std::vector<int> cpus;                  // the CPUs the workers are bound to: of the process's affinity mask, physical cores first
    int n_first = 0;                        // ... how many of the mask's CPUs are the first hardware thread of their core
    int n_threads = 0;

    Workers() {
        const int max_cpus = 8192;
        cpu_set_t *set = CPU_ALLOC(max_cpus);
        if (set) {
            const size_t bytes = CPU_ALLOC_SIZE(max_cpus);
            CPU_ZERO_S(bytes, set);
            if (sched_getaffinity(0, bytes, set) == 0)
                for (int c = 0; c < max_cpus; ++c)
                    if (CPU_ISSET_S(c, bytes, set)) cpus.push_back(c);
            CPU_FREE(set);
        }
        // The first hardware thread of every core in front, the other threads of the cores behind them (from the kernel's
        // topology files: a host without SMT, or a mask that holds first threads only, has nothing in the second part).
        // Fewer workers than CPUs (a CPU-time quota, a share of a multi-GPU run: mc_host_cores): spread over the physical cores,
        // so that they sit on both sockets and share no core -- and ROTATED by $MCALLER_HOST_CORE_OFFSET: the workers of a sharded
        // run whose GPUs hang off the same NUMA node are bound to the same CPUs and would all pick the same few of them
        // (multi_gpu._worker gives worker r the offset r x its share)
        std::vector<int> first, rest;
        for (int c : cpus) {
            char path[128];
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            int lowest = c;
            if (FILE *f = fopen(path, "r")) {
                char list[256] = "";
                if (fgets(list, (int)sizeof(list), f)) {
                    for (char *q = list; *q;) {                 // "3,131" or "6-7": the smallest sibling that is in our mask
                        char *end = nullptr;
                        const long a = strtol(q, &end, 10);
                        if (end == q) break;
                        long b = a;
                        q = end;
                        if (*q == '-') { b = strtol(q + 1, &end, 10); q = end; }
                        for (long x = a; x <= b; ++x)
                            if (x < lowest && std::find(cpus.begin(), cpus.end(), (int)x) != cpus.end()) lowest = (int)x;
                        while (*q == ',' || *q == '\n' || *q == ' ') ++q;
                    }
                }
                fclose(f);
            }
            (lowest == c ? first : rest).push_back(c);
        }
        n_first = (int)first.size();
        cpus = first;
        cpus.insert(cpus.end(), rest.begin(), rest.end());
        const int use = mc_host_cores();
        if (use < (int)cpus.size()) {
            const size_t pool = (size_t)n_first >= (size_t)use ? (size_t)n_first : cpus.size();
            size_t off = 0;
            if (const char *e = getenv("MCALLER_HOST_CORE_OFFSET")) { if (atoll(e) > 0) off = (size_t)atoll(e); }
            std::vector<int> pick;
            for (int i = 0; i < use; ++i) pick.push_back(cpus[(off + (size_t)i * pool / (size_t)use) % pool]);
            cpus.swap(pick);
        }
    }

    // run tasks of job j until it has none left to hand out; returns the number this thread ran
    static void run_tasks(Job *j) {
        int ran = 0;
        for (int i; (i = j->next.fetch_add(1)) < j->n;) { (*j->fn)(i); ++ran; }
        if (ran) j->finished.fetch_add(ran);
    }

    void loop(int cpu) {
        if (cpu >= 0) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpu, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
        }
        for (;;) {
            Job *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                wake.wait(lk, [&] {
                    for (Job *x : jobs)
                        if (x->next.load(std::memory_order_relaxed) < x->n) { j = x; return true; }
                    return false;
                });
                j->visitors += 1;           // (the job lives on its caller's stack: the caller waits for its visitors to leave)
            }
            run_tasks(j);
            {
                std::lock_guard<std::mutex> lk(mu);
                j->visitors -= 1;
                if (j->visitors == 0 && j->finished.load() >= j->n) done.notify_all();
            }
        }
    }
};
Workers *g_workers = nullptr;               // (never destroyed: its threads may outlive static destructors)
std::once_flag g_workers_once;
}  // namespace

void mc_parallel_for(int n, const std::function<void(int)> &f) {
    if (n <= 0) return;
    if (n == 1) { f(0); return; }
    std::call_once(g_workers_once, [] { g_workers = new Workers(); });
    Workers &W = *g_workers;
    Job job;
    job.fn = &f;
    job.n = n;
    int n_threads_now;
    {
        std::lock_guard<std::mutex> lk(W.mu);
        const int want = std::min(n - 1, std::max((int)W.cpus.size() - 1, 0));
        while (W.n_threads < want) {
            const int cpu = W.cpus[(size_t)W.n_threads % W.cpus.size()];
            std::thread([&W, cpu] { W.loop(cpu); }).detach();
            ++W.n_threads;
        }
        W.jobs.push_back(&job);
        n_threads_now = W.n_threads;
    }
    // (as many workers as the job has tasks for: waking them all -- a few hundred threads on a big host, each to find the tasks
    // gone -- costs the process CPU time it may not have)
    if (n - 1 >= n_threads_now) W.wake.notify_all();
    else for (int i = 0; i < n - 1; ++i) W.wake.notify_one();
    Workers::run_tasks(&job);                                  // the caller takes tasks as well
    {
        std::unique_lock<std::mutex> lk(W.mu);
        // every task has been handed out: off the list (no worker can pick the job any more), then wait for the tasks that
        // are still running and for the workers that hold a pointer to the job
        for (size_t i = 0; i < W.jobs.size(); ++i)
            if (W.jobs[i] == &job) { W.jobs.erase(W.jobs.begin() + (long)i); break; }
        W.done.wait(lk, [&] { return job.visitors == 0 && job.finished.load() >= n; });
    }
}
