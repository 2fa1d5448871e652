// Native `.diffs.<k>` row formatter (host side of libmcaller_hip.so).
//
// Replaces the per-observation text assembly of the reference's flush -- extract_contexts.py:186-216: the k slot
// means printed with str(np.float64) (shortest round-trip repr, literal `0` for an empty slot :186), the read quality,
// the 2k-1 context sliced from the marked strand and reverse-complemented for '-' reads (:194), the label from
// p >= 0.5 (:200-206), str(np.round(p, 2)) (:207) and the row layout (:216) -- for flush records that are already
// in host memory.  Records the host must look at itself (a context that leaves the contig, an unscored record, a
// sub-model key that does not exist, a base without a complement: the reference's exit/crash paths) stop the run:
// *stop_at names the first such record and the rows before it are returned.
#include "mc_rowtext.h"
#include "../../include/mcaller_hip.h"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

void mc_set_error(const char *fmt, ...);
void mc_parallel_for(int n, const std::function<void(int)> &f);   // mc_common.cpp: tasks 0..n-1 on the kept worker threads

namespace {

// repr(float) / str(np.float64): shortest digits that round-trip, Python's layout rules (float_repr_style 'short':
// exponent form iff decpt > 16 or decpt < -3; at least two exponent digits; ".0" appended to integers).
inline char *put_repr(char *o, double v) {
    if (std::isnan(v)) { memcpy(o, "nan", 3); return o + 3; }
    if (std::isinf(v)) { if (v < 0) *o++ = '-'; memcpy(o, "inf", 3); return o + 3; }
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (v == 0.0) { memcpy(o, "0.0", 3); return o + 3; }
    // A mean of 2, 4, 5, 8, 10 ... values of four decimals IS fl(D / 10^8) for an integer D (seven slot means in ten that are not
    // fl(d / 10^4) themselves): then the decimal D / 10^8, at most fifteen significant digits, trailing zeros dropped, is the shortest
    // string that round-trips (two decimals of up to 15 digits never share a double) -- an integer printed instead of the search
    // for the shortest digits (25 ns against 80-95).  Checked, not assumed: the division is done again.
    static const bool no_dec = getenv("MCALLER_FMT_NO_DEC") != nullptr;      // (probe)
    if (!no_dec && v >= 1e-4 && v < 1e7) {
        const long long D = (long long)std::nearbyint(v * 1e8);
        if ((double)D / 1e8 == v) {
            const unsigned long long ip = (unsigned long long)D / 100000000ull;
            unsigned fp = (unsigned)((unsigned long long)D % 100000000ull);
            o = std::to_chars(o, o + 24, ip).ptr;
            *o++ = '.';
            if (fp == 0) { *o++ = '0'; return o; }
            char f8[8];
            for (int i = 7; i >= 0; --i) { f8[i] = (char)('0' + fp % 10u); fp /= 10u; }
            int nf = 8;
            while (f8[nf - 1] == '0') --nf;
            memcpy(o, f8, (size_t)nf);
            return o + nf;
        }
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci) - 1, v, std::chars_format::scientific);   // d[.ddd]e[+-]XX
    *r.ptr = 0;
    char digits[24];
    int nd = 0;
    const char *p = sci;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    const int exp10 = (int)strtol(p + 1, nullptr, 10);
    const int decpt = exp10 + 1;
    if (decpt > 16 || decpt < -3) {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, (size_t)nd - 1); o += nd - 1; }
        *o++ = 'e';
        int e = decpt - 1;
        *o++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *o++ = eb[--ne];
        return o;
    }
    if (decpt <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, digits, (size_t)nd); o += nd;
        return o;
    }
    if (decpt >= nd) {
        memcpy(o, digits, (size_t)nd); o += nd;
        for (int i = nd; i < decpt; ++i) *o++ = '0';
        *o++ = '.'; *o++ = '0';
        return o;
    }
    memcpy(o, digits, (size_t)decpt); o += decpt;
    *o++ = '.';
    memcpy(o, digits + decpt, (size_t)(nd - decpt)); o += nd - decpt;
    return o;
}

// repr(d / 1e4) for a 32-bit integer d without a floating-point conversion: a decimal of at most ten significant digits IS the
// shortest string that round-trips to its double (two decimals of up to 15 digits never share a double), so the digits of d with
// the point four places from the right, trailing zeros dropped, ".0" for an integer -- |d / 1e4| < 214748.4 and >= 0.0001 keep
// Python out of the exponent form.  53 % of the slot means of the headline workload and every one-event slot travel as such a d
// (k_pack, mc_calls_view.feats_lo32): six of these per row were half of the formatter's time.
inline char *put_fixed4(char *o, int32_t d) {
    uint32_t a = d < 0 ? (uint32_t)(-(int64_t)d) : (uint32_t)d;
    if (d < 0) *o++ = '-';
    const uint32_t ip = a / 10000u, fp = a % 10000u;
    char tmp[12];
    int n = 0;
    uint32_t v = ip;
    do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n) *o++ = tmp[--n];
    *o++ = '.';
    if (fp == 0) { *o++ = '0'; return o; }
    char f4[4] = {(char)('0' + fp / 1000u), (char)('0' + fp / 100u % 10u), (char)('0' + fp / 10u % 10u), (char)('0' + fp % 10u)};
    int nf = 4;
    while (f4[nf - 1] == '0') --nf;
    memcpy(o, f4, (size_t)nf);
    return o + nf;
}

inline char *put_int(char *o, long long v) {
    const auto r = std::to_chars(o, o + 24, v);
    return r.ptr;
}

inline char *put_str(char *o, const char *s, size_t n) {
    memcpy(o, s, n);
    return o + n;
}

inline int comp_of(int c) {          // base_comps, extract_contexts.py:11; -1: KeyError there
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'N': return 'N';
        case 'M': return 'M';
        default: return -1;
    }
}

// Memory that is written once per call and thrown away -- the pieces the threads format into, the block the rows leave in -- is
// KEPT between calls: a shard of a one-base motif is 16 MB of rows, and fresh memory of that size comes from mmap, every page of
// it a fault taken under the process's one address-space lock by sixteen threads at once (the formatter scaled 3.9x on 8
// threads; per row and thread it took twice as long on 16 as on one).
struct PartBuf {                        // a growable piece of text that keeps its memory
    char *p = nullptr;
    size_t cap = 0, used = 0;
    void need(size_t more) {
        if (used + more <= cap) return;
        const size_t ncap = std::max(cap * 2, used + more + (1u << 20));
        char *q = (char *)realloc(p, ncap);
        if (!q) throw std::bad_alloc();
        p = q; cap = ncap;
    }
};
std::mutex g_fmt_mu;                    // one formatting call at a time owns the pieces below
std::vector<PartBuf> g_parts;
struct OutBlock { char *p; size_t cap; bool busy; };
std::mutex g_out_mu;
std::vector<OutBlock> g_out;            // blocks handed out by mc_format_diffs, kept when mc_free gets them back (at most four)

char *out_alloc(size_t n) {
    std::lock_guard<std::mutex> lk(g_out_mu);
    int best = -1;
    for (size_t i = 0; i < g_out.size(); ++i)
        if (!g_out[i].busy && g_out[i].cap >= n && (best < 0 || g_out[i].cap < g_out[(size_t)best].cap)) best = (int)i;
    if (best >= 0) { g_out[(size_t)best].busy = true; return g_out[(size_t)best].p; }
    const size_t cap = std::max<size_t>(n + n / 4, 1u << 20);
    char *p = (char *)malloc(cap);
    if (!p) return nullptr;
    for (size_t i = 0; i < g_out.size(); ++i)          // (a block that is too small makes room for this one)
        if (!g_out[i].busy) { free(g_out[i].p); g_out[i] = OutBlock{p, cap, true}; return p; }
    if (g_out.size() < 4) { g_out.push_back(OutBlock{p, cap, true}); return p; }
    return p;                                           // (more than four in use at once: a plain block, freed as such)
}

bool out_release(void *p) {
    std::lock_guard<std::mutex> lk(g_out_mu);
    for (auto &b : g_out)
        if (b.p == p) { b.busy = false; return true; }
    return false;
}

struct Job {
    const mc_format_args *a;
    std::vector<size_t> name_len, qual_len, contig_len_txt;
    size_t tail_len = 0, lab_pos_len = 0, lab_neg_len = 0;
};

// segment that holds row r: searchsorted(seg_row_begin, r, 'right') - 1
inline int32_t seg_of_row(const mc_table_view *t, int64_t r) {
    const int64_t *b = t->seg_row_begin, *e = b + t->n_seg + 1;
    return (int32_t)(std::upper_bound(b, e, r) - b) - 1;
}

// marked context of record j into ctx[0..2k-1); false: the host has to handle this record
inline bool build_context(const mc_format_args *a, int64_t j, char *ctx) {
    const mc_calls_view *rec = a->rec;
    const mc_ref_view *ref = a->ref;
    const int k = a->k;
    const uint32_t info = rec->info[j];
    if (info & MC_I_EDGE) return false;
    const int32_t seg = rec->site_seg[j];
    if (seg < 0 || seg >= a->table->n_seg) return false;
    const int32_t cid = a->table->seg_contig[seg];
    if (cid < 0 || cid >= ref->n_contigs) return false;
    const int64_t L = ref->contig_len[cid], m = rec->site_pos[j];
    const int64_t lo = m - k + 1, hi = m + k;           // last_ref[mpos-k+1 : mpos+k]  (:194)
    if (lo < 0 || hi > L) return false;
    const bool rev = info & MC_I_REV;
    const uint8_t *seq = ref->seq + ref->seq_off[cid];
    const uint32_t *bits = (rev ? ref->mbits_rev : ref->mbits_fwd) + ref->word_off[cid];
    const int n = 2 * k - 1;
    for (int i = 0; i < n; ++i) {
        const int64_t p = lo + i;
        const int c = ((bits[p >> 5] >> (p & 31)) & 1u) ? 'M' : seq[p];
        if (!rev) ctx[i] = (char)c;
        else {
            const int cc = comp_of(c);
            if (cc < 0) return false;
            ctx[n - 1 - i] = (char)cc;
        }
    }
    if (ctx[k - 1] != 'M') return false;                 // :224-228
    const unsigned char nxt = (unsigned char)ctx[k];
    if (a->submodel_of_char[nxt] == 255) return false;   // :218-223
    if (((info >> MC_I_NEXT_SHIFT) & 0xFFu) != nxt) return false;   // device and host disagree: let the host say so
    return true;
}

inline int64_t close_row_of(const mc_calls_view *rec, int64_t j) {
    return rec->close_row ? rec->close_row[j] : (int64_t)rec->close_row32[j];
}

// records [lo, hi) without MC_I_TOO_MANY
int64_t kept_in(const mc_calls_view *rec, int64_t lo, int64_t hi) {
    int64_t kept = 0;
    for (int64_t j = lo; j < hi; ++j) kept += (rec->info[j] & MC_I_TOO_MANY) ? 0 : 1;
    return kept;
}

// Row of feats / prob that belongs to record j, for a walk over consecutive records: compacted views without a call_row
// column count the records without MC_I_TOO_MANY as they go (mc_calls_view)
// wide slots (mc_calls_view.feats_wide) of call rows [0, row)
int64_t wide_before(const mc_calls_view *rec, int64_t row) {
    int64_t w = 0;
    for (int64_t r = 0; r < row; ++r) w += __builtin_popcount(rec->feats_wide[r]);
    return w;
}

struct RowCursor {
    const mc_calls_view *rec;
    int64_t kept;           // records without MC_I_TOO_MANY before the current one (compacted views without call_row)
    int64_t wide;           // packed slot means: wide slots before the current call row
    bool wide_known;
    RowCursor(const mc_calls_view *r, int64_t first)
        : rec(r), kept(r->compacted && !r->call_row ? kept_in(r, 0, first) : 0), wide(0), wide_known(false) {}
    // (the caller has counted: kept records before `first`, wide slots of the call rows before the first one of this walk)
    RowCursor(const mc_calls_view *r, int64_t kept_before, int64_t wide_before_)
        : rec(r), kept(kept_before), wide(wide_before_), wide_known(true) {}
    // (call for every record in order; info = rec->info[j])
    int64_t row(int64_t j, uint32_t info) {
        if (!rec->compacted) return j;
        if (rec->call_row) return rec->call_row[j];
        return (info & MC_I_TOO_MANY) ? -1 : kept++;
    }
    // the k slot means of call row `row` (rows in ascending order) for printing: bit s of the result set <=> slot s travels as the
    // integer d[s] = value x 1e4 (printed by put_fixed4), else f[s] holds the double
    unsigned feats_for_print(int64_t row, int k, double *f, int32_t *d) {
        if (rec->feats) { memcpy(f, rec->feats + row * k, (size_t)k * sizeof(double)); return 0u; }
        if (!wide_known) { wide = wide_before(rec, row); wide_known = true; }
        const unsigned mask = rec->feats_wide[row];
        const int32_t *lo = rec->feats_lo32 + row * k;
        for (int s = 0; s < k; ++s) {
            if ((mask >> s) & 1u) {
                const uint64_t bits = ((uint64_t)rec->feats_hi32[wide++] << 32) | (uint32_t)lo[s];
                memcpy(&f[s], &bits, 8);
            } else {
                d[s] = lo[s];
            }
        }
        return ~mask & ((1u << k) - 1u);
    }
    // the k slot means of call row `row` (rows must come in ascending order) into f[]
    void feats(int64_t row, int k, double *f) {
        if (rec->feats) { memcpy(f, rec->feats + row * k, (size_t)k * sizeof(double)); return; }
        if (!wide_known) { wide = wide_before(rec, row); wide_known = true; }
        const unsigned mask = rec->feats_wide[row];
        const int32_t *lo = rec->feats_lo32 + row * k;
        for (int s = 0; s < k; ++s) {
            if ((mask >> s) & 1u) {
                const uint64_t bits = ((uint64_t)rec->feats_hi32[wide++] << 32) | (uint32_t)lo[s];
                memcpy(&f[s], &bits, 8);
            } else {
                f[s] = (double)lo[s] / 1e4;
            }
        }
    }
};

// rows of records [j0, j1) into `out`; -> the first record the host must handle itself (a context that leaves the contig, an
// unscored record, ...: the rows before it are in `out`), or j1
// (what a thread writes per row lives in ITS locals: the pieces' buffers and row counts sit side by side in memory, and sixteen
// threads updating neighbouring words per row pass the cache line around -- the formatter took the same 20 ms on 1, 4 and 8 threads)
int64_t format_range(const Job &J, int64_t j0, int64_t j1, PartBuf &out_shared, int64_t &n_rows_shared, const RowCursor &start) {
    PartBuf out = out_shared;
    int64_t n_rows = 0;
    struct WriteBack {                  // (also when a piece runs out of memory: the shared entry must not keep a pointer realloc gave up)
        PartBuf &shared, &mine;
        int64_t &rows_shared, &rows_mine;
        ~WriteBack() { shared = mine; rows_shared = rows_mine; }
    } write_back{out_shared, out, n_rows_shared, n_rows};
    const mc_format_args *a = J.a;
    const mc_calls_view *rec = a->rec;
    const mc_table_view *t = a->table;
    const int k = a->k;
    out.used = 0;
    RowCursor cur = start;
    int32_t cseg = -1;                  // (the closing rows of consecutive records ascend: the segment is carried along)
    int64_t j = j0;
    for (; j < j1; ++j) {
        const uint32_t info = rec->info[j];
        const int64_t row = cur.row(j, info);                                      // (compacted views: mc_wait_records)
        if (info & MC_I_TOO_MANY) continue;
        if (std::isnan(rec->prob[row])) break;                                     // (scored by the host)
        const int32_t seg = rec->site_seg[j];
        const int32_t rid = t->seg_read[seg];
        const int64_t crow = close_row_of(rec, j);
        if (cseg < 0 || crow < t->seg_row_begin[cseg] || (cseg < t->n_seg && crow >= t->seg_row_begin[cseg + 1])) {
            if (cseg >= 0 && cseg + 1 < t->n_seg && crow >= t->seg_row_begin[cseg + 1] && crow < t->seg_row_begin[cseg + 2]) ++cseg;
            else cseg = seg_of_row(t, crow);
        }
        const char *chrom;
        size_t chrom_len;
        if (cseg >= t->n_seg) { chrom = a->tail_chrom; chrom_len = J.tail_len; }     // R8: the closing row's contig
        else { const int32_t cc = t->seg_contig[cseg]; chrom = a->contig_names[cc]; chrom_len = J.contig_len_txt[(size_t)cc]; }
        const size_t need = chrom_len + J.name_len[(size_t)rid] + J.qual_len[(size_t)rid] + (size_t)k * 32 + 160;
        out.need(need);
        char *o = out.p + out.used;
        o = put_str(o, chrom, chrom_len); *o++ = '\t';
        o = put_str(o, a->read_names[rid], J.name_len[(size_t)rid]); *o++ = '\t';
        o = put_int(o, rec->site_pos[j]); *o++ = '\t';
        if (!build_context(a, j, o)) break;                                        // (the host's own handling: nothing of this row is kept)
        o += 2 * k - 1;
        *o++ = '\t';
        const uint32_t empty = info & MC_I_EMPTY_MASK;
        double f[MC_MAX_K];
        int32_t d[MC_MAX_K];
        const unsigned as_int = cur.feats_for_print(row, k, f, d);
        for (int i = 0; i < k; ++i) {
            if ((empty >> i) & 1u) *o++ = '0';                                     // literal int 0  (:186)
            else if ((as_int >> i) & 1u) o = put_fixed4(o, d[i]);
            else o = put_repr(o, f[i]);
            *o++ = ',';
        }
        o = put_str(o, a->read_qual_txt[rid], J.qual_len[(size_t)rid]); *o++ = '\t';
        *o++ = (info & MC_I_REV) ? '-' : '+'; *o++ = '\t';
        const double p1 = rec->prob[row];
        if (p1 >= 0.5) o = put_str(o, a->label_meth, J.lab_pos_len);               // :200-206
        else o = put_str(o, a->label_unmeth, J.lab_neg_len);
        *o++ = '\t';
        // np.round(p, 2) (:207): an integer number of hundredths, printed as such
        const double h = std::nearbyint(p1 * 100.0);
        if (h >= 0.0 && h <= 100.0) {
            const int hi = (int)h;
            if (hi == 100) { memcpy(o, "1.0", 3); o += 3; }
            else { *o++ = '0'; *o++ = '.'; *o++ = (char)('0' + hi / 10); if (hi % 10) *o++ = (char)('0' + hi % 10); }
        } else o = put_repr(o, h / 100.0);
        *o++ = '\n';
        out.used = (size_t)(o - out.p);
        ++n_rows;
    }
    return j;
}

}  // namespace

extern "C" int mc_format_diffs(const mc_format_args *a, int64_t first, int32_t n_threads, char **text, int64_t *n_bytes,
                               int64_t *n_rows, int64_t *stop_at) {
    *text = nullptr;
    *n_bytes = 0;
    *n_rows = 0;
    *stop_at = first;
    if (!a || !a->rec || !a->table || !a->ref || a->k < 1 || a->k > MC_MAX_K || first < 0 || first > a->n_records) {
        mc_set_error("mc_format_diffs: bad arguments");
        return -12;
    }
    const int64_t n = a->n_records;
    const int threads = n_threads > 0 ? n_threads : mc_host_cores();     // the cores this process may use
    // four pieces per thread, handed out as the threads get to them: on a host that is shared, some of the cores the workers are
    // bound to run at half the others' pace, and with one piece each the slowest one is the call's time
    const int nt = (int)std::min<int64_t>(threads > 1 ? 4 * (int64_t)threads : 1, std::max<int64_t>(1, (n - first) / 2048));

    // every thread formats its piece of [first, n) and stops at the first record the host must handle itself; the rows before
    // the first such record of all count, what later pieces formatted is dropped (a handful of records per file end a run)
    Job J;
    J.a = a;
    const mc_table_view *t = a->table;
    J.name_len.resize((size_t)std::max(t->n_reads, 0));
    J.qual_len.resize((size_t)std::max(t->n_reads, 0));
    for (int32_t i = 0; i < t->n_reads; ++i) {
        J.name_len[(size_t)i] = a->read_names[i] ? strlen(a->read_names[i]) : 0;
        J.qual_len[(size_t)i] = a->read_qual_txt[i] ? strlen(a->read_qual_txt[i]) : 0;
    }
    J.contig_len_txt.resize((size_t)std::max(a->ref->n_contigs, 0));
    for (int32_t i = 0; i < a->ref->n_contigs; ++i) J.contig_len_txt[(size_t)i] = strlen(a->contig_names[i]);
    J.tail_len = a->tail_chrom ? strlen(a->tail_chrom) : 0;
    J.lab_pos_len = strlen(a->label_meth);
    J.lab_neg_len = strlen(a->label_unmeth);
    std::lock_guard<std::mutex> own_the_pieces(g_fmt_mu);
    if (g_parts.size() < (size_t)nt) g_parts.resize((size_t)nt);
    std::vector<PartBuf> &parts = g_parts;
    std::vector<int64_t> rows((size_t)nt, 0), stops((size_t)nt, n);
    std::vector<int> failed((size_t)nt, 0);
    static const bool fmt_timing = getenv("MCALLER_FMT_TIMING") != nullptr;      // (per-piece milliseconds on stderr)
    std::vector<double> t_piece((size_t)nt, 0.0);
    auto piece_lo = [&](int w) { return first + (n - first) * w / nt; };
    // where every piece begins in the compacted rows (records without MC_I_TOO_MANY before it, wide slots before its first row)
    const mc_calls_view *rec0 = a->rec;
    const bool counted = rec0->compacted && !rec0->call_row;
    std::vector<int64_t> kept0((size_t)nt + 1, 0), wide0((size_t)nt + 1, 0);
    if (counted) {
        kept0[0] = kept_in(rec0, 0, first);
        mc_parallel_for(nt, [&](int w) { kept0[(size_t)w + 1] = kept_in(rec0, piece_lo(w), piece_lo(w + 1)); });
        for (int w = 0; w < nt; ++w) kept0[(size_t)w + 1] += kept0[(size_t)w];
        if (!rec0->feats && rec0->feats_wide) {
            wide0[0] = wide_before(rec0, kept0[0]);
            mc_parallel_for(nt, [&](int w) {
                int64_t c = 0;
                for (int64_t r = kept0[(size_t)w], e = kept0[(size_t)w + 1]; r < e; ++r) c += __builtin_popcount(rec0->feats_wide[r]);
                wide0[(size_t)w + 1] = c;
            });
            for (int w = 0; w < nt; ++w) wide0[(size_t)w + 1] += wide0[(size_t)w];
        }
    }
    auto work = [&](int w) {
        const int64_t lo = piece_lo(w), hi = piece_lo(w + 1);
        const RowCursor start = counted ? RowCursor(rec0, kept0[(size_t)w], wide0[(size_t)w]) : RowCursor(rec0, lo);
        const auto t_a = std::chrono::steady_clock::now();
        try {
            stops[(size_t)w] = format_range(J, lo, hi, parts[(size_t)w], rows[(size_t)w], start);
        } catch (const std::bad_alloc &) { failed[(size_t)w] = 1; }
        if (fmt_timing) t_piece[(size_t)w] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
    };
    const auto t_0 = std::chrono::steady_clock::now();
    mc_parallel_for(nt, work);
    if (fmt_timing) {
        fprintf(stderr, "[mc_format_diffs] %d pieces in %.2f ms:", nt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_0).count() * 1e3);
        for (int w = 0; w < nt; ++w) fprintf(stderr, " %.2f", t_piece[(size_t)w] * 1e3);
        fprintf(stderr, "\n");
    }
    for (int w = 0; w < nt; ++w)
        if (failed[(size_t)w]) { mc_set_error("mc_format_diffs: out of memory"); return -10; }
    int64_t stop = n;
    int last = nt;                        // pieces [0, last) count
    for (int w = 0; w < nt; ++w)
        if (stops[(size_t)w] < piece_lo(w + 1)) { stop = stops[(size_t)w]; last = w + 1; break; }
    *stop_at = stop;
    size_t total = 0;
    std::vector<size_t> at((size_t)last + 1, 0);
    for (int w = 0; w < last; ++w) { at[(size_t)w] = total; total += parts[(size_t)w].used; }
    char *outp = out_alloc(std::max<size_t>(total, 1));
    if (!outp) {
        mc_set_error("mc_format_diffs: out of memory (%zu bytes)", total);
        return -10;
    }
    mc_parallel_for(last, [&](int w) { memcpy(outp + at[(size_t)w], parts[(size_t)w].p, parts[(size_t)w].used); });
    for (int w = 0; w < last; ++w) *n_rows += rows[(size_t)w];
    *text = outp;
    *n_bytes = (int64_t)total;
    return 0;
}

// What a shard's records add to the reference's counters (:184-185, :234-239, :247-248 -- sets of (read, site) pairs) and to the
// set of positions (:235), in ONE pass without the interpreter: a streamed shard of a one-base motif has 150 000 records, and the
// dozen numpy passes this replaces were 1 ms of interpreter lock per shard, taken from the other helper and from the main thread.
extern "C" int mc_count_records(const mc_calls_view *rec, int64_t n, const int32_t *seg_read, int64_t n_seg, int64_t *counts3,
                                int32_t *ascending, uint8_t *pos_marks, int64_t n_marks, int64_t *pos_min, int64_t *pos_top) {
    if (!rec || n < 0 || (n > 0 && (!rec->info || !rec->site_pos || !rec->site_seg)) || (counts3 && !seg_read)) {
        mc_set_error("mc_count_records: bad arguments");
        return -12;
    }
    int64_t too = 0, wskips = 0, multi = 0, lo_pos = INT64_MAX, top = 0;
    bool asc = true;
    int64_t prev_key = INT64_MIN;
    for (int64_t j = 0; j < n; ++j) {
        const uint32_t info = rec->info[j];
        const int32_t pos = rec->site_pos[j];
        const bool is_too = (info & MC_I_TOO_MANY) != 0;
        if (counts3) {
            const int32_t seg = rec->site_seg[j];
            if (seg < 0 || seg >= n_seg) { mc_set_error("mc_count_records: record %lld names segment %d of %lld", (long long)j, seg, (long long)n_seg); return -12; }
            const int64_t key = ((int64_t)seg_read[seg] << 32) | (int64_t)(uint32_t)pos;
            asc = asc && key > prev_key;
            prev_key = key;
            too += is_too;
            wskips += !is_too && (info & MC_I_EMPTY_MASK) != 0;
            multi += (info & MC_I_MULTI) != 0;
        }
        if (!is_too) {
            lo_pos = std::min<int64_t>(lo_pos, pos);
            top = std::max<int64_t>(top, (int64_t)pos + 1);
            if (pos_marks && pos >= 0 && pos < n_marks) pos_marks[pos] = 1;
        }
    }
    if (counts3) { counts3[0] = too; counts3[1] = wskips; counts3[2] = multi; }
    if (ascending) *ascending = asc ? 1 : 0;
    if (pos_min) *pos_min = lo_pos == INT64_MAX ? 0 : lo_pos;
    if (pos_top) *pos_top = top;
    return 0;
}

extern "C" int mc_calls_expand(const mc_calls_view *rec, int64_t n, int32_t k, int32_t *call_row_out, int64_t *close_row_out,
                               double *feats_out) {
    if (!rec || n < 0 || (n > 0 && !rec->info) || (close_row_out && n > 0 && !rec->close_row && !rec->close_row32) ||
        (feats_out && (k < 1 || k > MC_MAX_K || (!rec->feats && !(rec->feats_lo32 && rec->feats_wide))))) {
        mc_set_error("mc_calls_expand: bad arguments");
        return -12;
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(mc_host_cores(), n / 16384));
    auto piece = [&](int w, int64_t &lo, int64_t &hi) { lo = n * w / nt; hi = n * (w + 1) / nt; };
    const bool count_rows = rec->compacted && !rec->call_row && (call_row_out || feats_out);
    std::vector<int64_t> kept((size_t)nt + 1, 0), wide((size_t)nt + 1, 0);
    if (count_rows)
        mc_parallel_for(nt, [&](int w) { int64_t lo, hi; piece(w, lo, hi); kept[(size_t)w + 1] = kept_in(rec, lo, hi); });
    for (int w = 0; w < nt; ++w) kept[(size_t)w + 1] += kept[(size_t)w];
    const bool packed = feats_out && !rec->feats;
    // (call rows are numbered in record order: the rows of piece w are [first_row(w), first_row(w + 1)))
    auto first_row = [&](int w) -> int64_t {
        if (!rec->compacted) { int64_t lo, hi; piece(std::min(w, nt - 1), lo, hi); return w >= nt ? n : lo; }
        if (rec->call_row) {
            if (w >= nt) return rec->n_call_rows;
            int64_t lo, hi; piece(w, lo, hi);
            for (int64_t j = lo; j < n; ++j) if (rec->call_row[j] >= 0) return rec->call_row[j];
            return rec->n_call_rows;
        }
        return kept[(size_t)w];
    };
    if (packed) {
        mc_parallel_for(nt, [&](int w) {
            int64_t c = 0;
            for (int64_t r = first_row(w), e = first_row(w + 1); r < e; ++r) c += __builtin_popcount(rec->feats_wide[r]);
            wide[(size_t)w + 1] = c;
        });
        for (int w = 0; w < nt; ++w) wide[(size_t)w + 1] += wide[(size_t)w];
    }
    mc_parallel_for(nt, [&](int w) {
        int64_t lo, hi;
        piece(w, lo, hi);
        if (close_row_out)
            for (int64_t j = lo; j < hi; ++j) close_row_out[j] = close_row_of(rec, j);
        if (call_row_out) {
            int64_t row = kept[(size_t)w];
            for (int64_t j = lo; j < hi; ++j) {
                if (!rec->compacted) call_row_out[j] = (int32_t)j;
                else if (rec->call_row) call_row_out[j] = rec->call_row[j];
                else call_row_out[j] = (rec->info[j] & MC_I_TOO_MANY) ? -1 : (int32_t)row++;
            }
        }
        if (feats_out) {
            const int64_t r0 = first_row(w), r1 = first_row(w + 1);
            if (rec->feats) {
                if (r1 > r0) memcpy(feats_out + r0 * k, rec->feats + r0 * k, (size_t)(r1 - r0) * (size_t)k * sizeof(double));
            } else {
                RowCursor cur(rec, lo);
                cur.wide = wide[(size_t)w];
                cur.wide_known = true;
                for (int64_t r = r0; r < r1; ++r) cur.feats(r, k, feats_out + r * k);
            }
        }
    });
    return 0;
}

extern "C" void mc_free(void *p) {
    if (p && !out_release(p)) free(p);          // (a block of rows goes back to the formatter's cache)
}

// repr(d / 1e4) from the integer alone (tests pin it against Python's)
extern "C" int mc_repr_fixed4(int32_t d, char *out32) {
    char *e = put_fixed4(out32, d);
    *e = 0;
    return (int)(e - out32);
}

// repr(float) as the device row writer makes it (mc_rowtext.h, host build; tests pin it against Python's and against mc_repr_double)
extern "C" int mc_repr_double_rowtext(double v, char *out32) {
    // (as the kernels do it: the digits made, packed into their 16 bytes, unpacked, laid out -- and the length the counting pass would say)
    RtStore w{out32};
    uint64_t lo;
    uint32_t meta;
    rt_num_pack(rt_num_of(v), &lo, &meta);
    const RtNum n = rt_num_unpack(lo, meta);
    if (!n.ok) { out32[0] = 0; return -1; }
    rt_put_num(w, n);
    *w.p = 0;
    if ((int)(w.p - out32) != rt_num_length(n)) { out32[0] = 0; return -2; }
    return (int)(w.p - out32);
}

// repr(float) alone (tests pin it against Python's)
extern "C" int mc_repr_double(double v, char *out32) {
    char *e = put_repr(out32, v);
    *e = 0;
    return (int)(e - out32);
}
