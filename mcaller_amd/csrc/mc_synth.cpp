// Synthetic eventalign text for file-to-file timing (bench.py, tools/file_to_file.py): a columnar table written as the
// 13-column TSV nanopolish produces, by all host cores.  Measurement plumbing -- the hot path never
// calls it.  The Python statement of the same format is mcaller_amd.synth.write_tsv (tests compare the two).
#include "../../include/mcaller_hip.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

void mc_set_error(const char *fmt, ...);
void mc_parallel_for(int n, const std::function<void(int)> &f);   // mc_common.cpp: tasks 0..n-1 on the kept worker threads

namespace {

inline char *put_uint(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char *put_int(char *p, int64_t v) {
    if (v < 0) { *p++ = '-'; return put_uint(p, (uint64_t)(-v)); }
    return put_uint(p, (uint64_t)v);
}

// a current stored in 1e-4 pA as "%.2f" (the tables synth.py makes hold multiples of 0.01 pA), else with four decimals
inline char *put_e4(char *p, int32_t v) {
    int64_t a = v;
    if (a < 0) { *p++ = '-'; a = -a; }
    p = put_uint(p, (uint64_t)(a / 10000));
    *p++ = '.';
    const int frac = (int)(a % 10000);
    *p++ = (char)('0' + frac / 1000);
    *p++ = (char)('0' + frac / 100 % 10);
    if (frac % 100) { *p++ = (char)('0' + frac / 10 % 10); *p++ = (char)('0' + frac % 10); }
    return p;
}

inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

}  // namespace

extern "C" int mc_synth_write_tsv(const char *path, const mc_table_view *t, const char *seq, int64_t seq_len,
                                  const char *contig, const char *const *read_names, int32_t n_threads, int64_t *n_bytes) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        mc_set_error("cannot create %s: %s", path, strerror(errno));
        return -1;
    }
    int nt = n_threads > 0 ? n_threads : mc_host_cores();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, t->n_seg));
    // pieces of whole segments, balanced by rows; two passes: sizes, then text written at its offset
    std::vector<int32_t> cut((size_t)nt + 1, 0);
    for (int i = 1; i < nt; ++i) {
        const int64_t target = t->n_rows * i / nt;
        cut[(size_t)i] = (int32_t)(std::lower_bound(t->seg_row_begin, t->seg_row_begin + t->n_seg, target) - t->seg_row_begin);
        cut[(size_t)i] = std::max(cut[(size_t)i], cut[(size_t)i - 1]);
    }
    cut[(size_t)nt] = t->n_seg;
    const size_t clen = strlen(contig);
    std::vector<std::string> text((size_t)nt);
    std::vector<int> rc((size_t)nt, 0);
    auto work = [&](int i) {
        std::string &out = text[(size_t)i];
        const int64_t r0 = t->seg_row_begin[cut[(size_t)i]], r1 = t->seg_row_begin[cut[(size_t)i + 1]];
        out.resize((size_t)(r1 - r0) * 160 + 256);
        char *p = &out[0];
        for (int32_t sg = cut[(size_t)i]; sg < cut[(size_t)i + 1]; ++sg) {
            const char *name = read_names[t->seg_read[sg]];
            const size_t nlen = strlen(name);
            for (int64_t r = t->seg_row_begin[sg]; r < t->seg_row_begin[sg + 1]; ++r) {
                if ((size_t)(p - &out[0]) + clen + nlen + 128 > out.size()) {    // (long names: grow)
                    const size_t used = (size_t)(p - &out[0]);
                    out.resize(out.size() * 2 + clen + nlen + 128);
                    p = &out[0] + used;
                }
                const int64_t pos = t->pos[r];
                if (pos < 0 || pos + 6 > seq_len) { rc[(size_t)i] = -12; return; }
                const uint8_t fl = t->flags[r];
                memcpy(p, contig, clen); p += clen; *p++ = '\t';
                p = put_int(p, pos); *p++ = '\t';
                memcpy(p, seq + pos, 6); p += 6; *p++ = '\t';
                memcpy(p, name, nlen); p += nlen;
                memcpy(p, "\tt\t", 3); p += 3;
                p = put_int(p, t->event_idx[r]); *p++ = '\t';
                p = put_e4(p, t->event_model_e4[2 * r]);
                memcpy(p, "\t1.500\t0.00200\t", 15); p += 15;
                if (fl & MC_F_MODEL_N) { memcpy(p, "NNNNNN", 6); }
                else if (fl & MC_F_KMER_EQ) { memcpy(p, seq + pos, 6); }
                else { for (int j = 0; j < 6; ++j) p[j] = comp(seq[pos + 5 - j]); }
                p += 6; *p++ = '\t';
                p = put_e4(p, t->event_model_e4[2 * r + 1]);
                memcpy(p, "\t1.50\t0.10\n", 11); p += 11;
            }
        }
        out.resize((size_t)(p - &out[0]));
    };
    mc_parallel_for(nt, work);
    for (int i = 0; i < nt; ++i)
        if (rc[(size_t)i]) {
            close(fd);
            mc_set_error("mc_synth_write_tsv: a row's position leaves the sequence");
            return rc[(size_t)i];
        }
    std::vector<int64_t> off((size_t)nt + 1, 0);
    for (int i = 0; i < nt; ++i) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)text[(size_t)i].size();
    std::vector<int> wrc((size_t)nt, 0);
    auto put = [&](int i) {
        const std::string &s = text[(size_t)i];
        size_t done = 0;
        while (done < s.size()) {
            const ssize_t w = pwrite(fd, s.data() + done, s.size() - done, off[(size_t)i] + (int64_t)done);
            if (w <= 0) { wrc[(size_t)i] = -1; return; }
            done += (size_t)w;
        }
    };
    mc_parallel_for(nt, put);
    close(fd);
    for (int i = 0; i < nt; ++i)
        if (wrc[(size_t)i]) {
            mc_set_error("write to %s failed: %s", path, strerror(errno));
            return -1;
        }
    if (n_bytes) *n_bytes = off[(size_t)nt];
    return 0;
}
