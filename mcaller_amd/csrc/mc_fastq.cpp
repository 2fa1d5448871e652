// FASTQ -> (read key, mean phred) pairs for libmcaller_hip.so (C ABI: include/mcaller_hip.h, mc_fastq_*).
//
// What the reference computes per record (read_qual.py:6-19): key = id.split(':')[0].split('_')[0] where id is the first
// whitespace-delimited token of the title line without its '@'; value = np.mean of the per-base phred scores
// (ASCII - 33), i.e. an exact integer sum divided by the count in float64.  Four-line records only (what nanopore
// basecallers write); a record whose third line does not start with '+' is an error, like a sequence/quality length
// mismatch (Biopython's FastqGeneralIterator raises ValueError for both in this shape of file).
//
// The file is mapped (or inflated through libz for `.gz`, loaded with dlopen so that the library has no link-time
// dependency on it) and cut into one piece per thread.  A piece starts at the first line L with L[0]=='@' and
// (L+2)[0]=='+': a quality line may start with '@', but then L+2 is a sequence line, which never starts with '+'.
// Any irregularity (error, failed resynchronisation) reruns the whole buffer on one thread so that the first error in
// file order is the one reported.
#include "../../include/mcaller_hip.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

void mc_set_error(const char *fmt, ...);
void mc_parallel_for(int n, const std::function<void(int)> &f);   // mc_common.cpp: tasks 0..n-1 on the kept worker threads

struct mc_fastq {
    std::string pool;               // keys, '\n' after each
    std::vector<int64_t> off;       // n+1 offsets into pool (offset of key i; off[n] = pool.size())
    std::vector<double> mean;
};

namespace {

struct Piece {
    std::string pool;
    std::vector<int64_t> len;       // key lengths
    std::vector<double> mean;
    std::string error;
    const char *stop = nullptr;     // where this piece stopped reading (start of the next record)
};

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

struct Line {
    const char *b, *e;              // [b, e) without the line terminator ("\n" or "\r\n")
    const char *next;               // start of the following line
    bool any;                       // false at end of buffer (Python's readline() == '')
};

inline Line read_line(const char *p, const char *end) {
    Line L;
    if (p >= end) { L.b = L.e = L.next = end; L.any = false; return L; }
    const char *nl = static_cast<const char *>(memchr(p, '\n', size_t(end - p)));
    L.b = p;
    L.any = true;
    if (nl) { L.e = nl; L.next = nl + 1; } else { L.e = end; L.next = end; }
    if (L.e > L.b && L.e[-1] == '\r' && nl) --L.e;          // "\r\n" is one line break under universal newlines
    return L;
}

inline bool blank(const Line &L) {
    for (const char *q = L.b; q < L.e; ++q) if (!is_space((unsigned char)*q)) return false;
    return true;
}

// Parses records whose title line starts in [p, limit); reads past limit to finish the last one.
void parse_records(const char *p, const char *limit, const char *end, Piece &out) {
    while (p < limit) {
        Line title = read_line(p, end);
        if (!title.any) break;
        if (blank(title)) { p = title.next; continue; }
        if (*title.b != '@') { out.error = "Records in Fastq files should start with '@' character"; return; }
        Line seq = read_line(title.next, end);
        Line plus = read_line(seq.next, end);
        if (plus.b >= plus.e || *plus.b != '+') { out.error = "multi-line FASTQ records are not supported"; return; }
        Line qual = read_line(plus.next, end);
        const char *qe = qual.e;
        while (qe > qual.b && qe[-1] == '\r') --qe;
        const char *sb = seq.b, *se = seq.e;
        while (sb < se && is_space((unsigned char)*sb)) ++sb;
        while (se > sb && is_space((unsigned char)se[-1])) --se;
        if ((qe - qual.b) != (se - sb)) {
            out.error = "Lengths of sequence and quality values differs for " + std::string(title.b, title.e);
            return;
        }
        const char *ib = title.b + 1;
        while (ib < title.e && is_space((unsigned char)*ib)) ++ib;
        const char *ie = ib;
        while (ie < title.e && !is_space((unsigned char)*ie)) ++ie;
        if (ie == ib) { out.error = "FASTQ record without a read id"; return; }
        const char *ke = ib;                                 // id.split(':')[0].split('_')[0]
        while (ke < ie && *ke != ':' && *ke != '_') ++ke;
        const unsigned char *q = reinterpret_cast<const unsigned char *>(qual.b);
        const int64_t n = qe - qual.b;
        uint64_t s = 0;
        for (int64_t i = 0; i < n; ++i) s += q[i];
        const int64_t sum = int64_t(s) - 33 * n;
        out.pool.append(ib, size_t(ke - ib));
        out.pool.push_back('\n');
        out.len.push_back(ke - ib);
        out.mean.push_back(n ? double(sum) / double(n) : std::nan(""));
        p = qual.next;
    }
    out.stop = p;
}

// First record start at or after p (p is a line start): a non-blank line beginning with '@' whose line+2 begins with '+'.
const char *resync(const char *p, const char *end) {
    for (int tries = 0; tries < 16 && p < end; ++tries) {
        Line a = read_line(p, end);
        if (!a.any) return end;
        if (a.b < a.e && *a.b == '@') {
            Line b = read_line(a.next, end);
            Line c = read_line(b.next, end);
            if (c.b < c.e && *c.b == '+') return p;
        }
        p = a.next;
    }
    return p >= end ? end : nullptr;
}

typedef void *(*gzopen_t)(const char *, const char *);
typedef int (*gzread_t)(void *, void *, unsigned);
typedef int (*gzclose_t)(void *);
typedef int (*gzbuffer_t)(void *, unsigned);

int inflate_file(const char *path, std::vector<char> &buf) {
    void *z = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!z) { mc_set_error("mc_fastq_read_quality: %s is gzip-compressed and libz.so.1 cannot be loaded", path); return -3; }
    gzopen_t zopen = (gzopen_t)dlsym(z, "gzopen");
    gzread_t zread = (gzread_t)dlsym(z, "gzread");
    gzclose_t zclose = (gzclose_t)dlsym(z, "gzclose");
    gzbuffer_t zbuffer = (gzbuffer_t)dlsym(z, "gzbuffer");
    if (!zopen || !zread || !zclose) { mc_set_error("mc_fastq_read_quality: libz.so.1 lacks gzopen/gzread/gzclose"); return -3; }
    void *g = zopen(path, "rb");
    if (!g) { mc_set_error("mc_fastq_read_quality: cannot open %s", path); return -2; }
    if (zbuffer) zbuffer(g, 1u << 20);
    size_t used = 0;
    buf.resize(size_t(1) << 24);
    for (;;) {
        if (buf.size() - used < (size_t(1) << 22)) buf.resize(buf.size() * 2);
        int got = zread(g, buf.data() + used, unsigned(std::min<size_t>(buf.size() - used, size_t(1) << 30)));
        if (got < 0) { zclose(g); mc_set_error("mc_fastq_read_quality: %s is not a readable gzip file", path); return -2; }
        if (got == 0) break;
        used += size_t(got);
    }
    zclose(g);
    buf.resize(used);
    return 0;
}

}  // namespace

extern "C" int mc_fastq_read_quality(const char *path, int32_t n_threads, mc_fastq **out) {
    if (!path || !out) { mc_set_error("mc_fastq_read_quality: null argument"); return -1; }
    *out = nullptr;
    std::vector<char> inflated;
    const char *base = nullptr;
    size_t size = 0;
    void *map = nullptr;
    if (strstr(path, ".gz")) {                              // the reference's test: fastqfi.find(".gz") != -1 (read_qual.py:7)
        int rc = inflate_file(path, inflated);
        if (rc) return rc;
        base = inflated.data();
        size = inflated.size();
    } else {
        int fd = open(path, O_RDONLY);
        if (fd < 0) { mc_set_error("mc_fastq_read_quality: cannot open %s", path); return -2; }
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); mc_set_error("mc_fastq_read_quality: cannot stat %s", path); return -2; }
        size = size_t(st.st_size);
        if (size) {
            map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { close(fd); mc_set_error("mc_fastq_read_quality: cannot map %s", path); return -2; }
            madvise(map, size, MADV_SEQUENTIAL);
            base = static_cast<const char *>(map);
        }
        close(fd);
    }
    const char *end = base + size;
    int nt = n_threads > 0 ? n_threads : mc_host_cores();       // the cores this process may use, not the machine's
    if (nt < 1) nt = 1;
    if (size / (size_t(1) << 22) + 1 < size_t(nt)) nt = int(size / (size_t(1) << 22) + 1);   // pieces of at least 4 MB

    std::vector<Piece> pieces;
    pieces.resize(size_t(nt));
    bool regular = true;
    if (nt > 1) {
        std::vector<const char *> starts(size_t(nt) + 1, end);
        starts[0] = base;
        for (int t = 1; t < nt && regular; ++t) {
            const char *p = base + size / size_t(nt) * size_t(t);
            const char *nl = static_cast<const char *>(memchr(p, '\n', size_t(end - p)));
            const char *s = nl ? resync(nl + 1, end) : end;
            if (!s) regular = false; else starts[size_t(t)] = s < starts[size_t(t) - 1] ? starts[size_t(t) - 1] : s;
        }
        if (regular) {
            mc_parallel_for(nt, [&](int t) { parse_records(starts[size_t(t)], starts[size_t(t) + 1], end, pieces[size_t(t)]); });
            for (int t = 0; t < nt && regular; ++t) {
                if (!pieces[size_t(t)].error.empty()) regular = false;
                else if (starts[size_t(t)] < starts[size_t(t) + 1] && pieces[size_t(t)].stop != starts[size_t(t) + 1]) regular = false;
            }
        }
    }
    if (nt == 1 || !regular) {
        pieces.assign(1, Piece());
        parse_records(base, end, end, pieces[0]);
    }
    int rc = 0;
    mc_fastq *f = nullptr;
    if (!pieces[0].error.empty() && pieces.size() == 1) {
        mc_set_error("%s", pieces[0].error.c_str());
        rc = -4;
    } else {
        f = new mc_fastq();
        size_t n = 0, bytes = 0;
        for (auto &p : pieces) { n += p.mean.size(); bytes += p.pool.size(); }
        f->pool.reserve(bytes);
        f->off.reserve(n + 1);
        f->mean.reserve(n);
        for (auto &p : pieces) {
            int64_t o = int64_t(f->pool.size());
            for (int64_t l : p.len) { f->off.push_back(o); o += l + 1; }
            f->pool += p.pool;
            f->mean.insert(f->mean.end(), p.mean.begin(), p.mean.end());
        }
        f->off.push_back(int64_t(f->pool.size()));
    }
    if (map) munmap(map, size);
    *out = f;
    return rc;
}

extern "C" int64_t mc_fastq_view(const mc_fastq *f, const char **key_pool, const int64_t **key_off, const double **mean) {
    if (!f) return -1;
    if (key_pool) *key_pool = f->pool.data();
    if (key_off) *key_off = f->off.data();
    if (mean) *mean = f->mean.data();
    return int64_t(f->mean.size());
}

extern "C" void mc_fastq_free(mc_fastq *f) { delete f; }
