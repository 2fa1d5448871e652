/*
 * mc_oracle.c -- CPU oracle (plain C) for the mCaller hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A literal, row-at-a-time restatement of the reference's window machine
 * (extract_contexts.py:147-291) and of the MLP forward it calls (:199), operating on the columnar
 * event table of include/mcaller_hip.h instead of text.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it (as the checker / the timed CPU baseline); the product
 * (mcaller_amd/) never does.
 *
 * Pinning: identical flush records to oracle/py_oracle.py on every committed micro-case and on the
 * reference's testdata (tests/test_oracle_pin.py, tests/test_host_pipeline.py); py_oracle.py itself is pinned against the reference run
 * in the build container (tests/golden/PIN_REPORT.json: 2000 micro-cases + 6 testdata runs, 0
 * differences).
 */
#include "../include/mcaller_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double *v;
    int64_t n, cap;
} slot_t;

static void slot_push(slot_t *s, double x) {
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 8;
        s->v = (double *)realloc(s->v, (size_t)s->cap * sizeof(double));
    }
    s->v[s->n++] = x;
}

/* NumPy DOUBLE pairwise_sum (np.mean at extract_contexts.py:186) */
static double pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

static double np_mean(const slot_t *s) { return (0.0 + pairwise_sum(s->v, s->n)) / (double)s->n; }

/* offset of the first 'M' in meth_ref[pos:pos+k] (:176, :250, :270), or -1 */
static int first_m(const mc_ref_view *R, int contig, int rev, int64_t pos, int k) {
    const uint32_t *bits = (rev ? R->mbits_rev : R->mbits_fwd) + R->word_off[contig];
    int64_t L = R->contig_len[contig];
    for (int i = 0; i < k; ++i) {
        int64_t p = pos + i;
        if (p >= L) break;
        if ((bits[p >> 5] >> (p & 31)) & 1u) return i;
    }
    return -1;
}

static int is_m(const mc_ref_view *R, int contig, int rev, int64_t p) {
    const uint32_t *bits = (rev ? R->mbits_rev : R->mbits_fwd) + R->word_off[contig];
    return (int)((bits[p >> 5] >> (p & 31)) & 1u);
}

static unsigned char comp_char(unsigned char c) {
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        case 'N': return 'N';
        case 'M': return 'M';
        default: return 0xFF; /* comp() raises KeyError (:15) */
    }
}

typedef struct {
    int has_mpos;
    int64_t mpos;
    slot_t slots[MC_MAX_K];
    int last_read, last_rev, last_seg;
    int64_t first_idx;
} machine_t;

static void clear_slots(machine_t *m, int k) {
    for (int i = 0; i < k; ++i) m->slots[i].n = 0;
}

static int emit_record(const mc_table_view *T, const mc_ref_view *R, machine_t *m, int k, int skip_thresh,
                       int64_t close_row, const mc_calls_view *out, int64_t *n_out) {
    (void)T;
    int64_t j = *n_out;
    if (j >= out->capacity) return -1;
    int nskip = 0;
    for (int i = 0; i < k; ++i) nskip += (m->slots[i].n == 0);
    uint32_t info = m->last_rev ? MC_I_REV : 0;
    int contig = T->seg_contig[m->last_seg];
    int64_t L = R->contig_len[contig];
    for (int i = 0; i < k; ++i) out->feats[j * k + i] = 0.0;
    if (nskip <= skip_thresh) {
        for (int i = 0; i < k; ++i) {
            int dst = m->last_rev ? i : k - 1 - i; /* :187-188 */
            if (m->slots[i].n == 0)
                info |= (1u << dst);
            else
                out->feats[j * k + dst] = np_mean(&m->slots[i]);
        }
        if (m->mpos - k + 1 < 0 || m->mpos + k > L || m->mpos < 1 || m->mpos + 1 >= L) {
            info |= MC_I_EDGE;
        } else {
            unsigned char c;
            if (!m->last_rev) {
                int64_t q = m->mpos + 1;
                c = is_m(R, contig, 0, q) ? 'M' : R->seq[R->seq_off[contig] + q];
            } else {
                int64_t q = m->mpos - 1;
                c = is_m(R, contig, 1, q) ? 'M' : comp_char(R->seq[R->seq_off[contig] + q]);
            }
            info |= ((uint32_t)c) << MC_I_NEXT_SHIFT;
        }
    } else {
        info |= MC_I_TOO_MANY;
    }
    out->site_pos[j] = (int32_t)m->mpos;
    out->site_seg[j] = m->last_seg;
    out->close_row[j] = close_row;
    out->info[j] = info;
    out->prob[j] = NAN;
    *n_out = j + 1;
    return 0;
}

/* The literal machine.  Returns 0, or -1 if the record buffer is too small. */
int mco_extract_features(const mc_table_view *T, const mc_ref_view *R, const double *qual, const mc_params *prm,
                         const mc_calls_view *out, int64_t *n_records) {
    const int k = prm->k, skip_thresh = prm->skip_thresh;
    machine_t m;
    memset(&m, 0, sizeof(m));
    m.last_read = prm->entry_read;
    m.first_idx = prm->entry_first_idx;
    m.last_seg = -1;
    int64_t n_out = 0;
    int rc = 0;

    for (int32_t seg = 0; seg < T->n_seg && rc == 0; ++seg) {
        const int name = T->seg_read[seg], contig = T->seg_contig[seg];
        for (int64_t r = T->seg_row_begin[seg]; r < T->seg_row_begin[seg + 1]; ++r) {
            const int64_t idx = T->event_idx[r];
            if (name != m.last_read) m.first_idx = idx;                                   /* :161-162 */
            if (qual[name] < prm->qual_thresh || (T->flags[r] & MC_F_MODEL_N)) continue;  /* :167-168 */
            int rev;
            if ((name != m.last_read && (T->flags[r] & MC_F_KMER_EQ)) || (name == m.last_read && idx > m.first_idx))
                rev = 0;                                                                  /* :169-174 */
            else
                rev = 1;
            const int64_t pos = T->pos[r];
            const int off = first_m(R, contig, rev, pos, k);                             /* :176 */

            if (m.has_mpos && m.mpos != 0 &&
                ((pos >= m.mpos + 1 && name == m.last_read) || name != m.last_read)) {   /* :179 */
                int64_t j = n_out;
                if (emit_record(T, R, &m, k, skip_thresh, r, out, &n_out) != 0) { rc = -1; break; }
                if (off < 0 || name != m.last_read || pos > m.mpos + skip_thresh + 1) {  /* :242-245 */
                    clear_slots(&m, k);
                    m.has_mpos = 0;
                } else {                                                                  /* :246-256 */
                    if (off != 0) out->info[j] |= MC_I_MULTI;
                    int64_t old = m.mpos;
                    m.mpos = pos + off;
                    int64_t s = m.mpos - old < k ? m.mpos - old : k;
                    slot_t prev[MC_MAX_K];                 /* new[i] = old[i-s] (i>=s), [] below (:255) */
                    memcpy(prev, m.slots, sizeof(prev));
                    for (int i = 0; i < k; ++i) m.slots[i] = prev[(int)((i - s + k) % k)];
                    for (int i = 0; i < s && i < k; ++i) m.slots[i].n = 0;
                }
            }

            if (off >= 0) {                                                               /* :269-287 */
                if (m.has_mpos && m.mpos != 0) {
                    if (name != m.last_read) {
                        m.has_mpos = 0;
                        clear_slots(&m, k);
                    } else if (rev != m.last_rev) {
                        m.has_mpos = 0; /* slots kept */
                    }
                }
                if (!(m.has_mpos && m.mpos != 0)) {
                    m.has_mpos = 1;
                    m.mpos = pos + off;
                }
                m.last_read = name;
                m.last_rev = rev;
                m.last_seg = seg;
                int64_t d = (int64_t)T->event_model_e4[2 * r] - (int64_t)T->event_model_e4[2 * r + 1];
                slot_push(&m.slots[off], (double)d / 10000.0);                            /* :286 */
            } else if (m.has_mpos && m.mpos != 0) {                                       /* :289-291 */
                m.has_mpos = 0;
                clear_slots(&m, k);
            }
        }
    }
    /* the first unfiltered row of the next shard (a new read) closes the last window; at EOF it is lost */
    if (rc == 0 && prm->tail_contig >= 0 && m.has_mpos && m.mpos != 0) {
        if (emit_record(T, R, &m, k, skip_thresh, T->n_rows, out, &n_out) != 0) rc = -1;
    }
    for (int i = 0; i < MC_MAX_K; ++i) free(m.slots[i].v);
    *n_records = n_out;
    return rc;
}

/* MLP forward (sklearn MLPClassifier.predict_proba for tanh hidden / logistic output; :199) */
int mco_mlp_forward(int32_t n_models, int32_t n_in, int32_t n_hidden, const double *W1, const double *b1,
                    const double *W2, const double *b2, const double *X, const uint8_t *submodel, int64_t n,
                    double *p) {
    for (int64_t r = 0; r < n; ++r) {
        int mi = submodel[r];
        if (mi >= n_models) {
            p[r] = NAN;
            continue;
        }
        const double *w1 = W1 + (size_t)mi * n_in * n_hidden, *bb1 = b1 + (size_t)mi * n_hidden;
        const double *w2 = W2 + (size_t)mi * n_hidden;
        const double *x = X + r * n_in;
        double z = 0.0;
        for (int j = 0; j < n_hidden; ++j) {
            double a = 0.0;
            for (int i = 0; i < n_in; ++i) a += x[i] * w1[i * n_hidden + j];
            z += tanh(a + bb1[j]) * w2[j];
        }
        z += b2[mi];
        p[r] = 1.0 / (1.0 + exp(-z));
    }
    return 0;
}

/* Random-forest predict_proba (scikit-learn RandomForestClassifier, binary; call site extract_contexts.py:199 with
 * classifier RF): float32 inputs, x[feature] <= threshold, mean over trees of v1/(v0+v1). */
int mco_forest_forward(int32_t n_models, int32_t n_in, const int32_t *model_tree_off, const int32_t *tree_node_off,
                       const int32_t *left, const int32_t *right, const int32_t *feature, const double *threshold,
                       const double *value, const double *X, const uint8_t *submodel, int64_t n, double *p) {
    for (int64_t r = 0; r < n; ++r) {
        int mi = submodel[r];
        if (mi >= n_models) {
            p[r] = NAN;
            continue;
        }
        double x[MC_MAX_K + 2];
        for (int i = 0; i < n_in; ++i) x[i] = (double)(float)X[r * n_in + i];
        double sum = 0.0;
        for (int t = model_tree_off[mi]; t < model_tree_off[mi + 1]; ++t) {
            int node = tree_node_off[t];
            while (left[node] >= 0) node = (x[feature[node]] <= threshold[node]) ? left[node] : right[node];
            double v0 = value[2 * (size_t)node], v1 = value[2 * (size_t)node + 1];
            double norm = (-0.0 + v0) + v1;
            if (norm == 0.0) norm = 1.0;
            sum += v1 / norm;
        }
        p[r] = sum / (double)(model_tree_off[mi + 1] - model_tree_off[mi]);
    }
    return 0;
}
