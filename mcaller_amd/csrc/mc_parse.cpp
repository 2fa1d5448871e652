// Native eventalign parser (host side of libmcaller_hip.so).
//
// Replaces the text ingest of the reference's hot loop -- extract_contexts.py:140-152 (seek, readlines
// batches, line.split()[:12]) plus the per-row conversions it feeds (int(read_pos) :175, int(read_ind)
// :162/:169, float(event_current)-float(model_current) :286, the k-mer comparisons :167/:169) -- by one
// pass that turns ~128 B of text per row into the 17 B/row columnar table of include/mcaller_hip.h.
//
// Currents are stored as integers in units of 1e-4 pA.  np.round(float(e)-float(m), 4) (:286) always
// equals fl(R/1e4) for the integer R = rint((e-m)*1e4); for the plain decimals nanopolish prints
// (<= 4 fractional digits) R is exactly E4-M4, so the table keeps E4 and M4 and the subtraction stays on
// the GPU.  For any other spelling strtod is used and the row stores (R, 0).
#include "../../include/mcaller_hip.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

void mc_set_error(const char *fmt, ...);
int mc_parser_uses_pool();      // mc_common.cpp: the columns of parsed tables live in the pinned host pool (mc_host_pool_config)
void mc_parallel_for(int n, const std::function<void(int)> &f);   // mc_common.cpp: tasks 0..n-1 on the kept worker threads

namespace {

inline bool is_ws(unsigned char c) {
    // ASCII subset of what str.split() treats as whitespace
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);
}

struct Tok {
    const char *p;
    size_t n;
};

// first 12 whitespace-separated tokens of [s, e); returns the number found (<= 12)
inline int split12(const char *s, const char *e, Tok *t) {
    int n = 0;
    while (s < e && n < 12) {
        while (s < e && is_ws((unsigned char)*s)) ++s;
        if (s >= e) break;
        const char *b = s;
        while (s < e && !is_ws((unsigned char)*s)) ++s;
        t[n].p = b;
        t[n].n = (size_t)(s - b);
        ++n;
    }
    return n;
}

inline bool parse_int(const Tok &t, int64_t *out) {
    const char *p = t.p, *e = t.p + t.n;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) {
        neg = (*p == '-');
        ++p;
    }
    if (p >= e) return false;
    int64_t v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
        if (v > (int64_t)1 << 40) return false;
    }
    *out = neg ? -v : v;
    return true;
}

// plain decimal with <= 4 fractional digits -> value * 1e4 as an integer
inline bool parse_e4_fast(const Tok &t, int64_t *out) {
    const char *p = t.p, *e = t.p + t.n;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) {
        neg = (*p == '-');
        ++p;
    }
    int64_t v = 0;
    int nd = 0;
    for (; p < e && *p >= '0' && *p <= '9'; ++p) {
        v = v * 10 + (*p - '0');
        if (++nd > 9) return false;
    }
    int nf = 0;
    if (p < e && *p == '.') {
        ++p;
        for (; p < e && *p >= '0' && *p <= '9'; ++p) {
            if (++nf > 4) return false;
            v = v * 10 + (*p - '0');
        }
    }
    if (p != e || (nd == 0 && nf == 0)) return false;
    for (int i = nf; i < 4; ++i) v *= 10;
    *out = neg ? -v : v;
    return true;
}

inline bool parse_double_slow(const Tok &t, double *out) {
    if (t.n == 0 || t.n > 64) return false;
    char buf[72];
    memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    for (size_t i = 0; i < t.n; ++i)
        if (buf[i] == 'x' || buf[i] == 'X') return false;  // float() has no hex form
    char *end = nullptr;
    errno = 0;
    double v = strtod(buf, &end);
    if (end != buf + t.n) return false;
    *out = v;
    return true;
}

struct SvHash {
    size_t operator()(std::string_view s) const { return std::hash<std::string_view>()(s); }
};

}  // namespace

// a column of the final table: raw memory, never zero-filled (every element is written by the piece that owns it) --
// plain, or a block of the pinned host pool when the table is going to be streamed to the GPU (mc_host_pool_config)
template <typename T>
struct RawCol {
    T *p = nullptr;
    size_t n = 0;
    bool pooled = false;
    RawCol() = default;
    RawCol(const RawCol &) = delete;
    RawCol &operator=(const RawCol &) = delete;
    ~RawCol() { release(); }
    void release() {
        if (pooled) mc_host_free(p); else free(p);
        p = nullptr;
    }
    bool alloc(size_t count) {
        release();
        n = count;
        pooled = mc_parser_uses_pool() != 0;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 64;     // (the GPU side reads whole 16-byte groups)
        p = pooled ? (T *)mc_host_alloc((int64_t)bytes) : (T *)malloc(bytes);
        return p != nullptr;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
};

struct mc_parsed {
    RawCol<int32_t> pos, evmu, idx;        // evmu: (event, model) pairs, two entries per row
    RawCol<uint8_t> flags;
    int n_pieces = 0;                      // pieces the byte range was cut into (one parser thread each)
    std::vector<int64_t> seg_begin;
    std::vector<int32_t> seg_read, seg_contig;
    std::vector<std::string> read_names;
    std::vector<std::string> unknown;
};

// The byte range [lo, hi) the reference's batch loop consumes (:141-148).
static void consumed_range(const char *base, int64_t fsize, int64_t startline, int64_t endline,
                           int64_t *lo, int64_t *hi) {
    int64_t linepos = startline - 500 > 0 ? startline - 500 : 0;
    if (linepos > fsize) linepos = fsize;
    *lo = linepos;
    while (linepos <= endline - 500) {
        if (linepos >= fsize) break;  // the reference would spin here; we stop
        // one readlines(8000000) batch: lines are taken until their total size exceeds the hint
        int64_t probe = linepos + 8000000;
        if (probe >= fsize) {
            linepos = fsize;
        } else {
            const void *nl = memchr(base + probe, '\n', (size_t)(fsize - probe));
            linepos = nl ? (int64_t)((const char *)nl - base) + 1 : fsize;
        }
    }
    *hi = linepos;
}

namespace {

// What one thread makes of its byte range.  Read names stay text here (one entry per change of name); ids are assigned
// when the chunks are stitched together, in file order, so equal names get equal ids across the whole file.
struct Chunk {
    std::vector<int32_t> pos, evmu, idx;   // evmu: (event, model) pairs
    std::vector<uint8_t> flags;
    std::vector<int64_t> seg_begin;        // local row index
    std::vector<int32_t> seg_name;         // index into names
    std::vector<int32_t> seg_contig;
    std::vector<std::string> names;        // one per name change inside the chunk
    std::vector<std::string> unknown;
    int rc = 0;
    std::string err;
    int64_t err_row = 0;
    int64_t n_rows = 0;                    // rows of the piece, once its columns have been moved to the table
    std::vector<char> buf;                 // the thread's read buffer (kept with the chunk, see ChunkPool)

    void reset() {                         // empty, capacities kept
        pos.clear(); evmu.clear(); idx.clear(); flags.clear(); seg_begin.clear(); seg_name.clear(); seg_contig.clear();
        names.clear(); unknown.clear(); rc = 0; err.clear(); err_row = 0; n_rows = 0;
    }
    size_t bytes_held() const {
        return pos.capacity() * 4 + evmu.capacity() * 4 + idx.capacity() * 4 + flags.capacity() + buf.capacity();
    }
};

// Chunks are kept between calls.  A file streamed in shards is parsed shard after shard with the same piece sizes: fresh
// vectors every time meant ~1 MB of first-touch page faults per piece and call, from up to 256 threads of one process at
// once (mmap_lock) -- the same parse took anything between 26 and 100 ms (measured, 1.17 GB, 64 threads).  With the
// memory kept nothing is mapped or faulted in after the first shards.
struct ChunkPool {
    std::mutex mu;
    std::vector<Chunk *> idle;
    size_t idle_bytes = 0;
    static constexpr size_t KEEP_BYTES = (size_t)2 << 30, KEEP_CHUNKS = 1024;

    Chunk *get() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!idle.empty()) {
                Chunk *c = idle.back();
                idle.pop_back();
                idle_bytes -= std::min(idle_bytes, c->bytes_held());
                return c;
            }
        }
        return new Chunk();
    }
    void put(Chunk *c) {
        c->reset();
        const size_t b = c->bytes_held();
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() >= KEEP_CHUNKS || idle_bytes + b > KEEP_BYTES) { delete c; return; }
        idle.push_back(c);
        idle_bytes += b;
    }
};
ChunkPool g_chunks;

using ContigMap = std::unordered_map<std::string_view, int32_t, SvHash>;

// One piece [lo, hi) of the file (cut at line starts).  The text is read with pread into a small private block buffer
// (256 KB, reused), not through the mapping: unmapping 1 GB of touched file pages costs ~60 ms of page-table teardown
// (measured) -- four times the parsing itself.
void parse_chunk(int fd, int64_t lo, int64_t hi, const ContigMap &contig_map, Chunk &C) {
    size_t guess = (size_t)((hi - lo) / 100 + 16);
    C.pos.reserve(guess);
    C.evmu.reserve(2 * guess);
    C.idx.reserve(guess);
    C.flags.reserve(guess);
    std::string last_contig_txt;
    int32_t last_contig = -2;
    Tok t[12];
    char msg[256];
    std::vector<char> &buf = C.buf;
    if (buf.size() < (256u << 10) + 4096) buf.resize((256u << 10) + 4096);
    int64_t off = lo;
    size_t have = 0;                       // bytes of an unfinished line carried over to the buffer's start
    while (off < hi || have > 0) {
        const size_t want = (size_t)std::min<int64_t>((int64_t)(buf.size() - have), hi - off);
        size_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, buf.data() + have + got, want - got, off + (int64_t)got);
            if (r < 0) { C.rc = -1; C.err = std::string("read failed: ") + strerror(errno); C.err_row = (int64_t)C.pos.size(); return; }
            if (r == 0) break;
            got += (size_t)r;
        }
        off += (int64_t)got;
        const size_t n = have + got;
        const bool last_block = off >= hi || got == 0;
        size_t usable = n;
        if (!last_block) {                 // complete lines only; the rest is carried over
            const void *lnl = memrchr(buf.data(), '\n', n);
            if (!lnl) {                    // a line longer than the buffer: make room and read on
                buf.resize(buf.size() * 2);
                have = n;
                continue;
            }
            usable = (size_t)((const char *)lnl - buf.data()) + 1;
        }
        const char *p = buf.data(), *end = buf.data() + usable;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        int nt = split12(p, le, t);
        p = nl ? nl + 1 : end;
        if (nt < 12) continue;  // :149-152

        // contig (:154-160)
        int32_t contig;
        if (last_contig != -2 && t[0].n == last_contig_txt.size() && memcmp(t[0].p, last_contig_txt.data(), t[0].n) == 0) {
            contig = last_contig;
        } else {
            auto it = contig_map.find(std::string_view(t[0].p, t[0].n));
            contig = it == contig_map.end() ? -1 : it->second;
            last_contig_txt.assign(t[0].p, t[0].n);
            last_contig = contig;
        }
        if (contig < 0) {
            C.unknown.emplace_back(t[0].p, t[0].n);
            continue;
        }
        int64_t pos, idx;
        if (!parse_int(t[1], &pos) || !parse_int(t[5], &idx)) {
            snprintf(msg, sizeof(msg), "invalid literal for int(): '%.*s' / '%.*s'", (int)t[1].n, t[1].p, (int)t[5].n, t[5].p);
            C.rc = -2; C.err = msg; C.err_row = (int64_t)C.pos.size();
            return;
        }
        if (pos < 0 || pos > 0x7fffffff || idx < -0x7fffffff || idx > 0x7fffffff) {
            C.rc = -2; C.err = "position/event index out of range"; C.err_row = (int64_t)C.pos.size();
            return;
        }
        int64_t e4, m4;
        if (!(parse_e4_fast(t[6], &e4) && parse_e4_fast(t[10], &m4) && e4 > -1000000000LL && e4 < 1000000000LL &&
              m4 > -1000000000LL && m4 < 1000000000LL)) {
            double e, m;
            if (!parse_double_slow(t[6], &e) || !parse_double_slow(t[10], &m)) {
                snprintf(msg, sizeof(msg), "could not convert string to float: '%.*s' / '%.*s'", (int)t[6].n, t[6].p,
                         (int)t[10].n, t[10].p);
                C.rc = -2; C.err = msg; C.err_row = (int64_t)C.pos.size();
                return;
            }
            double r = std::nearbyint((e - m) * 10000.0);  // np.round(x,4) numerator (:286)
            if (!(std::fabs(r) < 2000000000.0)) {
                C.rc = -2; C.err = "current difference not representable"; C.err_row = (int64_t)C.pos.size();
                return;
            }
            e4 = (int64_t)r;
            m4 = 0;
        }
        uint8_t fl = 0;
        if (t[2].n == t[9].n && memcmp(t[2].p, t[9].p, t[2].n) == 0) fl |= MC_F_KMER_EQ;
        if (t[9].n == 6 && memcmp(t[9].p, "NNNNNN", 6) == 0) fl |= MC_F_MODEL_N;

        const bool new_name = C.names.empty() || t[3].n != C.names.back().size() ||
                              memcmp(t[3].p, C.names.back().data(), t[3].n) != 0;
        if (new_name) {
            C.names.emplace_back(t[3].p, t[3].n);
            fl |= MC_F_NAME_START;
        }
        if (new_name || C.seg_contig.empty() || C.seg_contig.back() != contig) {
            fl |= MC_F_SEG_START;
            C.seg_begin.push_back((int64_t)C.pos.size());
            C.seg_name.push_back((int32_t)C.names.size() - 1);
            C.seg_contig.push_back(contig);
        }
        C.pos.push_back((int32_t)pos);
        C.idx.push_back((int32_t)idx);
        C.evmu.push_back((int32_t)e4);
        C.evmu.push_back((int32_t)m4);
        C.flags.push_back(fl);
    }
        have = n - usable;
        if (have) memmove(buf.data(), buf.data() + usable, have);
        if (last_block) break;
    }
}

}  // namespace

static int parse_file(const char *path, int64_t startline, int64_t endline, bool exact_range,
                      const char *const *contig_names, int32_t n_contigs, int32_t n_threads, mc_parsed **out);

extern "C" int mc_parse_eventalign(const char *path, int64_t startline, int64_t endline,
                                   const char *const *contig_names, int32_t n_contigs, int32_t n_threads,
                                   mc_parsed **out) {
    return parse_file(path, startline, endline, false, contig_names, n_contigs, n_threads, out);
}

extern "C" int mc_parse_eventalign_range(const char *path, int64_t byte_begin, int64_t byte_end,
                                         const char *const *contig_names, int32_t n_contigs, int32_t n_threads,
                                         mc_parsed **out) {
    return parse_file(path, byte_begin, byte_end, true, contig_names, n_contigs, n_threads, out);
}

// Byte offsets that cut the file into n_parts pieces of similar size, each cut at the start of a line whose read name
// (column 4) differs from the line before it: a window never spans two reads (extract_contexts.py:179,242), so the pieces
// can be scanned independently (one GPU each).  cuts[0] = 0, cuts[n_parts] = file size; pieces may be empty.
extern "C" int mc_eventalign_read_cuts(const char *path, int32_t n_parts, int64_t *cuts) {
    return mc_eventalign_read_cuts_range(path, 0, INT64_MAX, n_parts, cuts);
}

// The same for the byte range [lo, hi) of the file (lo at a line start; hi is clamped to the file size): cuts[0] = lo,
// cuts[n_parts] = hi.
// cuts[0] = lo, cuts[i] = the first line of a read at or behind want[i - 1] (i = 1 .. n_parts - 1; want == nullptr: pieces of equal
// size), cuts[n_parts] = min(hi, file size); in order (a cut that would lie before the one in front of it collapses onto it)
static int read_cuts(const char *path, int64_t lo, int64_t hi, int32_t n_parts, const int64_t *want, int64_t *cuts) {
    if (n_parts < 1) {
        mc_set_error("mc_eventalign_read_cuts: n_parts %d", n_parts);
        return -12;
    }
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        mc_set_error("cannot open %s: %s", path, strerror(errno));
        return -1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        mc_set_error("cannot stat %s", path);
        close(fd);
        return -1;
    }
    const int64_t file_size = (int64_t)st.st_size;
    const char *base = nullptr;
    if (file_size > 0) {
        base = (const char *)mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) {
            mc_set_error("cannot mmap %s: %s", path, strerror(errno));
            close(fd);
            return -1;
        }
    }
    close(fd);
    const int64_t fsize = std::min(std::max<int64_t>(hi, 0), file_size);     // (everything below works on [lo, fsize))
    lo = std::min(std::max<int64_t>(lo, 0), fsize);
    auto name_of = [&](int64_t line, int64_t line_end, Tok *name) -> bool {      // column 4 of a line, if it has one
        Tok t[12];
        const int n = split12(base + line, base + line_end, t);
        if (n < 4) return false;
        *name = t[3];
        return true;
    };
    auto line_end_of = [&](int64_t line) -> int64_t {
        const void *nl = memchr(base + line, '\n', (size_t)(fsize - line));
        return nl ? (int64_t)((const char *)nl - base) + 1 : fsize;
    };
    cuts[0] = lo;
    // every cut is searched for on its own (a read is ~10^4 lines: the searches are the time), then put in order: a cut that
    // would lie before the one in front of it collapses onto it (an empty piece)
    std::vector<int64_t> found((size_t)n_parts, lo);
    mc_parallel_for(n_parts - 1, [&](int task) {
        const int32_t i = task + 1;
        int64_t c = want ? std::min(want[i - 1], fsize) : lo + (fsize - lo) * i / n_parts;
        if (c <= lo) { found[(size_t)i] = lo; return; }
        if (c >= fsize) { found[(size_t)i] = fsize; return; }
        // the line that contains byte c-1 ends at `line`: start there, remember the name of the line before it
        int64_t line = c;
        {
            const void *nl = memchr(base + c - 1, '\n', (size_t)(fsize - (c - 1)));
            line = nl ? (int64_t)((const char *)nl - base) + 1 : fsize;
        }
        int64_t prev_begin = line - 1;                       // start of the line that ends at `line`
        while (prev_begin > lo && base[prev_begin - 1] != '\n') --prev_begin;
        Tok prev;
        bool have_prev = name_of(prev_begin, line, &prev);
        // A read is ~10^4 lines: before the lines are walked one by one, leaps -- the line behind a probe twice as far each time,
        // as long as it still carries prev's name (lines of one name are taken for one run here: if the name comes back behind
        // another read, the walk below still stops at a first line of a read, which is all a cut has to be)
        if (have_prev) {
            int64_t safe = line;                              // a line start: everything in [line, safe) carries prev's name ... as far as probed
            for (int64_t step = 1 << 16; safe + step < fsize; step *= 2) {
                const void *nl = memchr(base + safe + step, '\n', (size_t)(fsize - (safe + step)));
                if (!nl) break;
                const int64_t probe = (int64_t)((const char *)nl - base) + 1;
                if (probe >= fsize) break;
                Tok cur;
                if (!name_of(probe, line_end_of(probe), &cur) || cur.n != prev.n || memcmp(cur.p, prev.p, cur.n) != 0) {
                    // the name changes in (safe, probe]: halve the distance
                    int64_t a = safe, b = probe;              // a: a line start with prev's name (or `line`), b: a line start without it
                    while (b - a > 4096) {
                        const int64_t mid = a + (b - a) / 2;
                        const void *nm = memchr(base + mid, '\n', (size_t)(b - mid));
                        const int64_t ml = nm ? (int64_t)((const char *)nm - base) + 1 : b;
                        if (ml >= b) break;
                        Tok t2;
                        if (name_of(ml, line_end_of(ml), &t2) && t2.n == prev.n && memcmp(t2.p, prev.p, t2.n) == 0) a = ml; else b = ml;
                    }
                    safe = a;
                    break;
                }
                safe = probe;
            }
            line = std::max(line, safe);
        }
        int64_t cut = fsize;
        while (line < fsize) {
            const int64_t le = line_end_of(line);
            Tok cur;
            if (name_of(line, le, &cur)) {
                if (have_prev && (cur.n != prev.n || memcmp(cur.p, prev.p, cur.n) != 0)) { cut = line; break; }
                prev = cur;
                have_prev = true;
            }
            line = le;
        }
        found[(size_t)i] = cut;
    });
    for (int32_t i = 1; i < n_parts; ++i) cuts[i] = std::max(found[(size_t)i], cuts[i - 1]);
    cuts[n_parts] = fsize;
    if (base) munmap((void *)base, (size_t)file_size);
    return 0;
}

extern "C" int mc_eventalign_read_cuts_range(const char *path, int64_t lo, int64_t hi, int32_t n_parts, int64_t *cuts) {
    return read_cuts(path, lo, hi, n_parts, nullptr, cuts);
}

extern "C" int mc_eventalign_read_cuts_at(const char *path, int64_t lo, int64_t hi, const int64_t *want, int32_t n_want, int64_t *cuts) {
    if (n_want < 0 || (n_want > 0 && !want)) {
        mc_set_error("mc_eventalign_read_cuts_at: bad arguments");
        return -12;
    }
    return read_cuts(path, lo, hi, n_want + 1, want, cuts);
}

// The byte range [*lo, *hi) the reference's batch loop consumes for (startline, endline) (:141-148): seek to
// max(startline-500, 0), readlines(8000000) batches while linepos <= endline-500 -- the last < 500 bytes of a file can stay
// unread.  What mc_parse_eventalign parses; exported so that a caller who cuts the range into shards cuts the same range.
extern "C" int mc_eventalign_consumed_range(const char *path, int64_t startline, int64_t endline, int64_t *lo, int64_t *hi) {
    *lo = *hi = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        mc_set_error("cannot open %s: %s", path, strerror(errno));
        return -1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        mc_set_error("cannot stat %s", path);
        close(fd);
        return -1;
    }
    const int64_t fsize = (int64_t)st.st_size;
    if (fsize > 0) {
        const char *base = (const char *)mmap(nullptr, (size_t)fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) {
            mc_set_error("cannot mmap %s: %s", path, strerror(errno));
            close(fd);
            return -1;
        }
        consumed_range(base, fsize, startline, endline, lo, hi);
        munmap((void *)base, (size_t)fsize);
    }
    close(fd);
    return 0;
}

static int parse_file(const char *path, int64_t startline, int64_t endline, bool exact_range,
                      const char *const *contig_names, int32_t n_contigs, int32_t n_threads, mc_parsed **out) {
    *out = nullptr;
    const bool trace = getenv("MCALLER_TRACE_HOST") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "  parse +%8.1f ms %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(), what);
    };
    if (n_threads <= 0) { if (const char *e = getenv("MCALLER_PARSE_THREADS")) n_threads = atoi(e); }
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        mc_set_error("cannot open %s: %s", path, strerror(errno));
        return -1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        mc_set_error("cannot stat %s", path);
        close(fd);
        return -1;
    }
    int64_t fsize = (int64_t)st.st_size;
    const char *base = nullptr;
    if (fsize > 0) {
        base = (const char *)mmap(nullptr, (size_t)fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) {
            mc_set_error("cannot mmap %s: %s", path, strerror(errno));
            close(fd);
            return -1;
        }
    }
    int64_t lo = 0, hi = 0;
    if (exact_range) {                  // [startline, endline) as given (a piece cut by mc_eventalign_read_cuts)
        lo = std::min(std::max<int64_t>(startline, 0), fsize);
        hi = std::min(std::max<int64_t>(endline, lo), fsize);
    } else if (fsize > 0) {
        consumed_range(base, fsize, startline, endline, &lo, &hi);
    }

    ContigMap contig_map;
    std::vector<std::string> contig_store(contig_names, contig_names + n_contigs);
    for (int32_t i = 0; i < n_contigs; ++i)
        contig_map.emplace(std::string_view(contig_store[i]), i);  // first id wins, like the FASTA scan :77-81

    // ---- cut [lo, hi) at line starts, one piece per thread ----
    // n_threads > 0: that many pieces; <= 0: one per core, but no piece smaller than 4 MB
    int nt = n_threads > 0 ? n_threads : mc_host_cores();      // the cores this process may use, not the machine's
    if (nt < 1) nt = 1;
    const int64_t min_piece = 4 << 20;
    if (n_threads <= 0 && (hi - lo) / nt < min_piece) nt = (int)std::max<int64_t>(1, (hi - lo) / min_piece);
    std::vector<int64_t> cuts;
    cuts.push_back(lo);
    for (int i = 1; i < nt; ++i) {
        int64_t c = lo + (hi - lo) * i / nt;
        if (c <= cuts.back()) continue;
        const void *nl = memchr(base + c, '\n', (size_t)(hi - c));
        if (!nl) break;
        c = (int64_t)((const char *)nl - base) + 1;
        if (c > cuts.back() && c < hi) cuts.push_back(c);
    }
    cuts.push_back(hi);
    const int np = (int)cuts.size() - 1;
    lap("mapped, cut");
    struct Borrowed {                  // chunks from the pool, handed back when parse_file returns (whichever way)
        std::vector<Chunk *> v;
        ~Borrowed() { for (Chunk *c : v) g_chunks.put(c); }
        Chunk &operator[](size_t i) { return *v[i]; }
    } chunks;
    for (int i = 0; i < np; ++i) chunks.v.push_back(g_chunks.get());
    std::vector<double> t_begin(trace ? (size_t)np : 0), t_end(trace ? (size_t)np : 0);
    std::vector<int> on_cpu(trace ? (size_t)np : 0);
    mc_parallel_for(np, [&](int i) {
        if (trace) { t_begin[(size_t)i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(); on_cpu[(size_t)i] = sched_getcpu(); }
        parse_chunk(fd, cuts[(size_t)i], cuts[(size_t)i + 1], contig_map, chunks[(size_t)i]);
        if (trace) t_end[(size_t)i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count();
    });
    lap("pieces parsed");
    if (trace && np > 0) {              // stragglers?  when the pieces started, how long they took, where they ran
        double b_max = 0, d_min = 1e9, d_max = 0, d_sum = 0;
        int n0 = 0;
        for (int i = 0; i < np; ++i) {
            const double d = t_end[(size_t)i] - t_begin[(size_t)i];
            b_max = std::max(b_max, t_begin[(size_t)i]); d_min = std::min(d_min, d); d_max = std::max(d_max, d); d_sum += d;
            n0 += on_cpu[(size_t)i] % 128 < 64 ? 1 : 0;
        }
        fprintf(stderr, "  pieces: last start +%.1f ms; duration min %.1f mean %.1f max %.1f ms; %d of %d on NUMA node 0\n", b_max, d_min,
                d_sum / np, d_max, n0, np);
    }
    if (base) munmap((void *)base, (size_t)fsize);     // (only the few pages around the cuts were touched)
    close(fd);
    lap("unmapped");

    // ---- stitch ----
    int64_t total = 0;
    for (int i = 0; i < np; ++i) {
        if (chunks[(size_t)i].rc != 0) {     // the first error in file order (earlier pieces are complete)
            mc_set_error("%s in eventalign row %lld", chunks[(size_t)i].err.c_str(), (long long)(total + chunks[(size_t)i].err_row));
            return chunks[(size_t)i].rc;
        }
        total += (int64_t)chunks[(size_t)i].pos.size();
    }
    mc_parsed *P = new mc_parsed();
    P->n_pieces = np;
    if (!P->pos.alloc((size_t)total) || !P->evmu.alloc((size_t)total * 2) ||
        !P->idx.alloc((size_t)total) || !P->flags.alloc((size_t)total)) {
        delete P;
        mc_set_error("out of memory for %lld rows", (long long)total);
        return -10;
    }
    // the pieces' columns go to their place in parallel (one thread per piece)
    {
        std::vector<int64_t> offs((size_t)np + 1, 0);
        for (int i = 0; i < np; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + (int64_t)chunks[(size_t)i].pos.size();
        auto place = [&](int i) {
            Chunk &C = chunks[(size_t)i];
            const size_t n = C.pos.size();
            const int64_t o = offs[(size_t)i];
            if (n) {
                memcpy(P->pos.data() + o, C.pos.data(), n * 4);
                memcpy(P->evmu.data() + 2 * o, C.evmu.data(), n * 8);
                memcpy(P->idx.data() + o, C.idx.data(), n * 4);
                memcpy(P->flags.data() + o, C.flags.data(), n);
            }
            C.n_rows = (int64_t)n;
        };
        mc_parallel_for(np, place);
    }
    lap("columns placed");
    std::unordered_map<std::string, int32_t> read_map;
    int64_t off = 0;
    std::string prev_name;
    bool have_prev = false;
    for (int i = 0; i < np; ++i) {
        Chunk &C = chunks[(size_t)i];
        const size_t n = (size_t)C.n_rows;
        std::vector<int32_t> name_id(C.names.size());
        for (size_t j = 0; j < C.names.size(); ++j) {
            auto it = read_map.find(C.names[j]);
            if (it == read_map.end()) {
                name_id[j] = (int32_t)P->read_names.size();
                read_map.emplace(C.names[j], name_id[j]);
                P->read_names.push_back(C.names[j]);
            } else {
                name_id[j] = it->second;
            }
        }
        for (size_t sgi = 0; sgi < C.seg_begin.size(); ++sgi) {
            const int32_t rid = name_id[(size_t)C.seg_name[sgi]];
            const int32_t contig = C.seg_contig[sgi];
            const int64_t row = off + C.seg_begin[sgi];
            if (sgi == 0 && have_prev && C.names[0] == prev_name) {
                // the piece starts inside a name block of the previous piece
                P->flags[(size_t)row] &= (uint8_t)~MC_F_NAME_START;
                if (!P->seg_contig.empty() && P->seg_contig.back() == contig) {
                    P->flags[(size_t)row] &= (uint8_t)~MC_F_SEG_START;
                    continue;                      // same (name, contig) segment continues
                }
            }
            P->seg_begin.push_back(row);
            P->seg_read.push_back(rid);
            P->seg_contig.push_back(contig);
        }
        if (!C.names.empty()) {
            prev_name = C.names.back();
            have_prev = true;
        }
        for (auto &u : C.unknown) P->unknown.push_back(std::move(u));
        off += (int64_t)n;
    }
    P->seg_begin.push_back(total);
    *out = P;
    lap("segments stitched");
    return 0;
}

// A byte range of a file into caller memory (pinned, for the device parser): pread from as many threads as the process may
// run on, 4 MB at a time (measured on the 2 x 64-core host: > 100 GB/s out of the page cache).
extern "C" int mc_read_file_range(const char *path, int64_t lo, int64_t hi, char *dst, int32_t n_threads) {
    if (!path || !dst || lo < 0 || hi < lo) {
        mc_set_error("mc_read_file_range: bad arguments");
        return -12;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        mc_set_error("cannot open %s: %s", path, strerror(errno));
        return -1;
    }
    const int64_t n = hi - lo, piece = 4 << 20;
    const int np = (int)std::max<int64_t>(1, (n + piece - 1) / piece);
    int nt = n_threads > 0 ? n_threads : mc_host_cores();
    nt = std::max(1, std::min(nt, np));
    std::vector<int> rc((size_t)nt, 0);
    std::atomic<int> next{0};
    mc_parallel_for(nt, [&](int w) {
        for (int i; (i = next.fetch_add(1)) < np;) {
            int64_t off = (int64_t)i * piece;
            const int64_t end = std::min(n, off + piece);
            while (off < end) {
                const ssize_t r = pread(fd, dst + off, (size_t)(end - off), lo + off);
                if (r <= 0) { rc[(size_t)w] = r < 0 ? errno : -1; return; }
                off += r;
            }
        }
    });
    close(fd);
    for (int e : rc)
        if (e) {
            mc_set_error("read of %s failed: %s", path, e > 0 ? strerror(e) : "file shorter than the range");
            return -1;
        }
    return 0;
}

extern "C" int mc_parsed_view(const mc_parsed *p, mc_table_view *out) {
    out->n_rows = (int64_t)p->pos.size();
    out->pos = p->pos.data();
    out->event_model_e4 = p->evmu.data();
    out->event_idx = p->idx.data();
    out->flags = p->flags.data();
    out->n_seg = (int32_t)p->seg_read.size();
    out->seg_row_begin = p->seg_begin.data();
    out->seg_read = p->seg_read.data();
    out->seg_contig = p->seg_contig.data();
    out->n_reads = (int32_t)p->read_names.size();
    return 0;
}

extern "C" const char *mc_parsed_read_name(const mc_parsed *p, int32_t read_id) {
    if (read_id < 0 || (size_t)read_id >= p->read_names.size()) return nullptr;
    return p->read_names[(size_t)read_id].c_str();
}

extern "C" int64_t mc_parsed_n_unknown(const mc_parsed *p) { return (int64_t)p->unknown.size(); }

extern "C" const char *mc_parsed_unknown_name(const mc_parsed *p, int64_t i) {
    if (i < 0 || (size_t)i >= p->unknown.size()) return nullptr;
    return p->unknown[(size_t)i].c_str();
}

extern "C" int32_t mc_parsed_n_pieces(const mc_parsed *p) { return p->n_pieces; }

extern "C" void mc_parsed_free(mc_parsed *p) { delete p; }
