"""Shared test plumbing: golden fixtures, the C oracle (the checker), case materialisation."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
MODELS = os.path.join(REPO, 'mcaller_amd', 'models')       # the weight exports the package ships
MODEL_STEMS = {'r95': 'r95_twobase_model_NN_6_m6A', 'r94': 'r94_model_NN_6_m6A',
               'CAAY': 'CAAYNNNNNRTAC_model_6_m6A', 'CRAA': 'CRAANNNNNNNTGC_model_6_m6A'}

_oracle = None


def oracle_lib():
    """Build (if stale) and load oracle/libmc_oracle.so -- the CPU checker, never the product."""
    global _oracle
    if _oracle is None:
        src = os.path.join(REPO, 'oracle', 'mc_oracle.c')
        so = os.path.join(REPO, 'oracle', 'libmc_oracle.so')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-ffp-contract=off', '-o', so, src, '-lm'])
        L = C.CDLL(so)
        from mcaller_amd import _lib
        L.mco_extract_features.argtypes = [C.POINTER(_lib.TableView), C.POINTER(_lib.RefView), C.c_void_p,
                                           C.POINTER(_lib.Params), C.POINTER(_lib.CallsView), C.POINTER(C.c_int64)]
        L.mco_mlp_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mco_forest_forward.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 9 + [C.c_int64, C.c_void_p]
        _oracle = L
    return _oracle


def model_meta():
    return json.load(open(os.path.join(GOLDEN, 'models', 'models_meta.json')))


def load_modelset(tag):
    from mcaller_amd.model_io import load_npz_weights
    stem = MODEL_STEMS[tag]
    return load_npz_weights(os.path.join(MODELS, stem + '.npz'), model_meta()[stem]['is_dict'])


def oracle_records(table, ref_arrays, qual, k, skip_thresh, qual_thresh, tail_contig=-1, entry_read=-1,
                   entry_first_idx=0):
    """Run the C oracle's literal machine on a host table -> mcaller_amd._lib.Records."""
    from mcaller_amd import _lib
    L = oracle_lib()
    tv = table.view()
    rv = _lib.make_ref_view(ref_arrays)
    q = np.ascontiguousarray(qual, dtype=np.float64)
    cap = table.n_rows + 2
    rec = _lib.Records(cap, k)
    cv = rec.view()
    prm = _lib.Params(k, skip_thresh, float(qual_thresh), tail_contig, 0, entry_read, entry_first_idx)
    n = C.c_int64(0)
    rc = L.mco_extract_features(C.byref(tv), C.byref(rv), q.ctypes.data_as(C.c_void_p), C.byref(prm), C.byref(cv),
                                C.byref(n))
    assert rc == 0
    rec.n = n.value
    return rec


def oracle_forest_forward(forests, X, submodel):
    from mcaller_amd.device import forest_arrays
    L = oracle_lib()
    a = forest_arrays(forests)
    X = np.ascontiguousarray(X, dtype=np.float64)
    sm = np.ascontiguousarray(submodel, dtype=np.uint8)
    p = np.full(len(X), np.nan)
    P_ = lambda z: z.ctypes.data_as(C.c_void_p)
    L.mco_forest_forward(len(forests), forests[0].n_in, P_(a['model_tree_off']), P_(a['tree_node_off']), P_(a['left']),
                         P_(a['right']), P_(a['feature']), P_(a['threshold']), P_(a['value']), P_(X), P_(sm), len(X), P_(p))
    return p


def load_rf_modelset():
    from mcaller_amd.model_io import load_model_file
    return load_model_file(os.path.join(GOLDEN, 'models', 'rf_twobase_model_RF_6_m6A.pkl'))


def oracle_score(rec, table, qual, weights, soc, k):
    """Fill rec.prob with the C oracle's classifier for records that are scored on the device too."""
    from mcaller_amd import _lib
    L = oracle_lib()
    n = rec.n
    if n == 0:
        return
    info = rec.info[:n]
    scored = (info & (_lib.I_TOO_MANY | _lib.I_EDGE)) == 0
    sub = soc[(info >> _lib.I_NEXT_SHIFT) & 0xFF].astype(np.uint8)
    sub[~scored] = 255
    X = np.zeros((n, k + 1), dtype=np.float64)
    X[:, :k] = rec.feats[:n * k].reshape(n, k)
    X[:, k] = np.asarray(qual, dtype=np.float64)[table.seg_read[rec.site_seg[:n]]]
    if weights[0].kind == 'forest':
        rec.prob[:n] = oracle_forest_forward(weights, X, sub)
        return
    if weights[0].kind in ('logistic', 'gnb'):
        from oracle import clf_oracle
        rec.prob[:n] = clf_oracle.forward(weights, X, sub)
        return
    W1 = np.ascontiguousarray(np.stack([w.W1 for w in weights]))
    b1 = np.ascontiguousarray(np.stack([w.b1 for w in weights]))
    W2 = np.ascontiguousarray(np.stack([w.W2 for w in weights]))
    b2 = np.ascontiguousarray(np.concatenate([w.b2 for w in weights]))
    p = np.full(n, np.nan)
    P_ = lambda a: a.ctypes.data_as(C.c_void_p)
    L.mco_mlp_forward(len(weights), k + 1, weights[0].n_hidden, P_(W1), P_(b1), P_(W2), P_(b2), P_(X), P_(sub), n, P_(p))
    rec.prob[:n] = p


def micro_cases():
    with gzip.open(os.path.join(GOLDEN, 'micro_cases.json.gz')) as fh:
        return json.loads(fh.read())


def materialise(case, d):
    """Write a micro-case's files into directory d -> dict of paths."""
    paths = dict(tsv=os.path.join(d, 'case.eventalign.tsv'), fasta=os.path.join(d, 'ref.fasta'),
                 fastq=os.path.join(d, 'reads.fastq'), positions=None)
    open(paths['tsv'], 'w').write(case['tsv'])
    open(paths['fasta'], 'w').write(case['fasta'])
    open(paths['fastq'], 'w').write(case['fastq'])
    if case['positions'] is not None:
        paths['positions'] = os.path.join(d, 'positions.txt')
        open(paths['positions'], 'w').write(case['positions'])
    return paths


def testdata_paths(d):
    """Materialise the reference's testdata (TSV gunzipped, FASTA rebuilt from the committed span)."""
    td = os.path.join(GOLDEN, 'testdata')
    out = dict(tsv=os.path.join(d, 'masonread1.eventalign.tsv'), fasta=os.path.join(d, 'pb_ecoli_polished_assembly.fasta'),
               fastq=os.path.join(td, 'masonread1.fastq'))
    if not os.path.exists(out['tsv']):
        with gzip.open(os.path.join(td, 'masonread1.eventalign.tsv.gz')) as src, open(out['tsv'], 'wb') as dst:
            dst.write(src.read())
        span = json.load(open(os.path.join(td, 'rebuilt_fasta_span.json')))
        seq = bytearray(b'N' * span['length'])
        s0 = span['span_start']
        seq[s0:s0 + len(span['span'])] = span['span'].encode()
        with open(out['fasta'], 'w') as fa:
            fa.write('>%s\n' % span['contig'])
            s = seq.decode()
            fa.write('\n'.join(s[i:i + 60] for i in range(0, len(s), 60)) + '\n')
    for name in ('test_positions.txt', 'test_positions_A.txt', 'test_positions_m6A.txt'):
        out[name] = os.path.join(td, name)
    return out


def pos2label(path):
    out = {}
    for line in open(path).read().split('\n'):
        t = line.split()
        if len(t) > 1:
            out[(t[0], int(t[1]), t[2])] = t[3]
    return out


def plain_signals(sig):
    return {k: {lab: [[(0 if isinstance(x, int) else repr(float(x))) for x in row] for row in rows]
                for lab, rows in v.items()} for k, v in sig.items()}


def assert_records_equal(got, want, k, prob_tol=1e-6):
    """Flush records from the HIP path vs the oracle: integers and slot means bit-for-bit, p within tol -- and what the
    reference PRINTS of p (the label p >= 0.5, np.round(p, 2): extract_contexts.py:200-207) the same for every record.
    (1e-6: the MLP's fast forward -- fp32, fp64 wherever a printed digit could depend on it; MCALLER_MLP_FP64=1 and the other
    classifiers are held to tighter bounds by their own tests.)"""
    assert got.n == want.n, 'record count %d vs oracle %d' % (got.n, want.n)
    n = got.n
    if getattr(got, 'call_row', None) is not None:
        # a pipelined pass (mc_wait_records): slot means / probabilities only for the records that are calls, compacted
        from mcaller_amd import _lib
        kept = (want.info[:n] & _lib.I_TOO_MANY) == 0
        rows = np.where(kept, np.cumsum(kept) - 1, -1)
        assert np.array_equal(got.call_row[:n], rows), 'call_row is not the running count of the records that are calls'
        assert got.n_calls == int(kept.sum())
        got = got.by_record()                 # zeros / NaN for the other records, which is what the oracle stores for them
    for name in ('site_pos', 'site_seg', 'close_row', 'info'):
        a, b = getattr(got, name)[:n], getattr(want, name)[:n]
        if not np.array_equal(a, b):
            i = int(np.nonzero(a != b)[0][0])
            raise AssertionError('%s differs first at record %d: %r vs %r (site_pos %d)' % (name, i, a[i], b[i], want.site_pos[i]))
    fa = got.feats[:n * k].view(np.uint64)
    fb = want.feats[:n * k].view(np.uint64)
    if not np.array_equal(fa, fb):
        i = int(np.nonzero(fa != fb)[0][0])
        raise AssertionError('slot mean differs at record %d slot %d: %r vs %r' % (i // k, i % k, got.feats[i], want.feats[i]))
    pa, pb = got.prob[:n], want.prob[:n]
    assert np.array_equal(np.isnan(pa), np.isnan(pb)), 'scored/unscored sets differ'
    ok = ~np.isnan(pa)
    if ok.any():
        err = np.abs(pa[ok] - pb[ok]).max()
        assert err <= prob_tol, 'probability differs by %g' % err
        assert np.array_equal(pa[ok] >= 0.5, pb[ok] >= 0.5), 'a label differs'
        assert np.array_equal(np.rint(pa[ok] * 100.0), np.rint(pb[ok] * 100.0)), 'a printed probability differs'
    assert_records_equal.last_prob_err = float(err) if ok.any() else 0.0
