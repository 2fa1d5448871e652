#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE in the build container.

Only works where /root/reference is mounted (never on the GPU box; no test imports this file).
It (1) builds a scratch directory with stand-ins for the two third-party packages the image lacks
(Biopython's SeqIO -- a ~40-line FASTA/FASTQ reader of our own -- and an empty `seaborn`), aliases
the pre-0.22 scikit-learn module paths the py2 pickles name, and rebuilds the missing test FASTA
from column 3 of the test TSV; (2) runs the reference's own CLI / functions on
its testdata and on random micro-cases from oracle/casegen.py; (3) compares every output with
oracle/py_oracle.py (the pin) and writes the captured reference outputs under tests/golden/.

usage: make_golden.py [n_micro_cases=2000] [--out DIR] [--rf-only | --simple-only]
(--out DIR: write the fixtures under DIR/tests/golden and DIR/mcaller_amd/models instead of into the repository.)

Outputs (all DATA: inputs + expected outputs + exported weight arrays, no reference source):
  tests/golden/testdata/          the reference's test inputs (TSV gz, FASTQ, position lists, the
                                  rebuilt span of the FASTA) and its own golden outputs
  tests/golden/ref_outputs/       outputs of the reference run here (diffs.6 for -p, -m GATC, -m A,
                                  --train; make_bed BEDs; stdout counter lines)
  mcaller_amd/models/*.npz        W1,b1,W2,b2 of every shipped estimator (meta: tests/golden/models/models_meta.json)
  tests/golden/micro_cases.json.gz  micro-cases with the reference's output for each
  tests/golden/PIN_REPORT.json    how many cases were compared, how many differed (must be 0)
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import runpy
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
# where the fixtures go: the repository, or a copy of its layout under --out DIR (tests/test_pin_recipe.py regenerates them
# into a scratch directory and compares with the committed bytes)
OUT_GOLDEN = HERE                                          # tests/golden/
OUT_PKG_MODELS = os.path.join(REPO, 'mcaller_amd', 'models')   # the weight exports ship with the package


def golden_models_dir():
    """models_meta.json, the RF fixture and its meta (the .npz weight exports live in OUT_PKG_MODELS)."""
    return os.path.join(OUT_GOLDEN, 'models')

SEQIO_STANDIN = '''
"""Minimal stand-in for Bio.SeqIO.parse (FASTA / FASTQ) -- ours, for running the reference here."""
import io

class _Rec(object):
    def __init__(self, rid, seq, phred=None):
        self.id = rid
        self.seq = seq
        self.letter_annotations = {'phred_quality': phred} if phred is not None else {}

def parse(handle, fmt):
    close = False
    if isinstance(handle, str):
        handle = open(handle, 'r')
        close = True
    try:
        if fmt == 'fasta':
            rid, chunks = None, []
            for line in handle:
                if line.startswith('>'):
                    if rid is not None:
                        yield _Rec(rid, ''.join(chunks))
                    title = line[1:].rstrip()
                    rid = title.split(None, 1)[0] if title.split() else ''
                    chunks = []
                elif rid is not None:
                    chunks.append(line.strip().replace(' ', '').replace('\\r', ''))
            if rid is not None:
                yield _Rec(rid, ''.join(chunks))
        elif fmt == 'fastq':
            while True:
                title = handle.readline()
                if not title:
                    break
                if not title.strip():
                    continue
                seq = handle.readline().strip()
                handle.readline()
                qual = handle.readline().rstrip('\\n').rstrip('\\r')
                yield _Rec(title[1:].split(None, 1)[0], seq, [ord(c) - 33 for c in qual])
        else:
            raise ValueError(fmt)
    finally:
        if close:
            handle.close()
'''


def install_shims(scratch):
    shim = os.path.join(scratch, 'shim')
    os.makedirs(os.path.join(shim, 'Bio'))
    os.makedirs(os.path.join(shim, 'seaborn'))
    open(os.path.join(shim, 'Bio', '__init__.py'), 'w').close()
    open(os.path.join(shim, 'Bio', 'SeqIO.py'), 'w').write(SEQIO_STANDIN)
    open(os.path.join(shim, 'seaborn', '__init__.py'), 'w').write('def set_style(*a, **k):\n    pass\n')
    sys.path.insert(0, shim)
    sys.path.insert(0, REF)
    os.environ['MPLBACKEND'] = 'Agg'
    import sklearn.neural_network._multilayer_perceptron as mlp
    import sklearn.preprocessing._label as lab
    sys.modules['sklearn.neural_network.multilayer_perceptron'] = mlp
    sys.modules['sklearn.preprocessing.label'] = lab
    import warnings
    warnings.filterwarnings('ignore')


def rebuild_fasta(tsv_path, out_path, only_span):
    """SURVEY.md §8(c) item 3.  `only_span=True` writes a compact fixture: the same contig name and
    length semantics are NOT needed by the tests (positions are absolute), so the fixture keeps the
    full length with N outside the read's span, gz-compressed."""
    length = 4734145
    seq = bytearray(b'N' * length)
    with open(tsv_path) as fh:
        for line in fh:
            t = line.split()
            p = int(t[1])
            seq[p:p + 6] = t[2].encode()
    seq[13153:13154] = b'A'
    with open(out_path, 'w') as out:
        out.write('>ecoli\n')
        s = seq.decode()
        for i in range(0, length, 60):
            out.write(s[i:i + 60] + '\n')
    return s


def run_cli(script, argv):
    """Run a reference script's __main__ with argv; capture stdout; tolerate sys.exit."""
    buf = io.StringIO()
    old = sys.argv
    sys.argv = [os.path.join(REF, script)] + argv
    code = None
    try:
        with contextlib.redirect_stdout(buf):
            try:
                runpy.run_path(os.path.join(REF, script), run_name='__main__')
            except SystemExit as e:
                code = e.code
    finally:
        sys.argv = old
    return buf.getvalue(), code


def export_models():
    import pickle
    import numpy as np
    outdir = golden_models_dir()
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(OUT_PKG_MODELS, exist_ok=True)
    meta = {}
    for fn in ['r95_twobase_model_NN_6_m6A.pkl', 'r94_model_NN_6_m6A.pkl',
               'CAAYNNNNNRTAC_model_6_m6A.pkl', 'CRAANNNNNNNTGC_model_6_m6A.pkl']:
        raw = open(os.path.join(REF, fn), 'rb').read()
        obj = pickle.loads(raw, encoding='latin')
        is_dict = isinstance(obj, dict)
        ests = obj if is_dict else {'general': obj}
        arrays = {}
        for key, est in ests.items():
            assert est.activation == 'tanh' and est.out_activation_ == 'logistic'
            assert [c.shape for c in est.coefs_] == [(7, 100), (100, 1)]
            arrays[key + '.W1'] = np.ascontiguousarray(est.coefs_[0], dtype=np.float64)
            arrays[key + '.b1'] = np.ascontiguousarray(est.intercepts_[0], dtype=np.float64)
            arrays[key + '.W2'] = np.ascontiguousarray(est.coefs_[1], dtype=np.float64)
            arrays[key + '.b2'] = np.ascontiguousarray(est.intercepts_[1], dtype=np.float64)
        if is_dict:
            arrays['__is_dict__'] = np.array([1], dtype=np.uint8)     # (a dict with the single key 'general' stays a dict)
        stem = fn[:-4]
        np.savez(os.path.join(OUT_PKG_MODELS, stem + '.npz'), **arrays)
        # known answers: predict_proba on fixed probe vectors, per sub-model
        rng = np.random.default_rng(7)
        probes = np.concatenate([rng.normal(0, 2.5, size=(64, 6)), rng.uniform(6, 12, size=(64, 1))], axis=1)
        probes[0] = [-0.4066666666666667, 1.6099999999999999, -1.6866666666666665, 6.67, -4.55, 1.775,
                     7.055265349382997]
        ka = {key: [float(v) for v in est.predict_proba(probes)[:, 1]] for key, est in ests.items()}
        meta[stem] = dict(sha256=hashlib.sha256(raw).hexdigest(), is_dict=is_dict,
                          submodels=sorted(ests.keys()),
                          classes=[c.decode() if isinstance(c, bytes) else str(c) for c in
                                   list(ests.values())[0].classes_],
                          probes=[[float(v) for v in row] for row in probes], known_answers=ka)
    json.dump(meta, open(os.path.join(outdir, 'models_meta.json'), 'w'), indent=1)
    return meta


def export_rf_fixture():
    """BASELINE config 5's alternative classifier: a scikit-learn RandomForestClassifier with the reference's
    hyper-parameters (train_model.py:40-45; `min_impurity_split` dropped, scikit-learn >= 1.0 rejects it) and a fixed
    random_state, fitted HERE on seeded synthetic vectors labelled by the shipped r95 MLP.  The pickle (data, written by
    this script) and predict_proba on probe vectors are the fixture for the forest kernel."""
    import pickle
    import numpy as np
    from sklearn.ensemble import RandomForestClassifier
    outdir = golden_models_dir()
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(11)
    ref = pickle.loads(open(os.path.join(REF, 'r95_twobase_model_NN_6_m6A.pkl'), 'rb').read(), encoding='latin')
    models, ka = {}, {}
    probes = np.concatenate([rng.normal(0, 2.5, size=(256, 6)), rng.uniform(6, 12, size=(256, 1))], axis=1)
    probes = np.round(probes, 4)
    for key in ('MG', 'MH'):
        X = np.concatenate([rng.normal(0, 2.5, size=(300, 6)), rng.uniform(6, 12, size=(300, 1))], axis=1)
        y = np.where(ref[key].predict_proba(X)[:, 1] + rng.normal(0, 0.15, size=300) >= 0.5, 'm6A', 'A')
        rf = RandomForestClassifier(bootstrap=True, criterion='entropy', max_depth=10, max_features=4,
                                    min_samples_leaf=2, min_samples_split=3, n_estimators=50, random_state=3)
        rf.fit(X, y)
        models[key] = rf
        ka[key] = [float(v) for v in rf.predict_proba(probes)[:, 1]]
    with open(os.path.join(outdir, 'rf_twobase_model_RF_6_m6A.pkl'), 'wb') as fh:
        pickle.dump(models, fh, protocol=4)
    import sklearn
    json.dump(dict(sklearn=sklearn.__version__, probes=[[float(v) for v in r] for r in probes], known_answers=ka,
                   classes=[str(c) for c in models['MG'].classes_]),
              open(os.path.join(outdir, 'rf_meta.json'), 'w'))


def export_simple_fixture():
    """The reference's other two closed-form classifiers (`-c LR`, `-c NBC`; train_model.py:55-60): scikit-learn
    LogisticRegression(solver='liblinear', penalty='l1') -- `multi_class='ovr'` is what that solver does for two classes anyway,
    and scikit-learn >= 1.5 warns about the argument; a fixed random_state so that the fixture can be regenerated byte for byte -- and
    GaussianNB(), fitted HERE on seeded synthetic vectors labelled by the
    shipped r95 MLP (as the RF fixture).  The pickles (dicts keyed by sub-model, the reference's format) and predict_proba on
    probe vectors are the fixture for the device's closed-form classifiers."""
    import pickle
    import numpy as np
    import sklearn
    from sklearn.linear_model import LogisticRegression
    from sklearn.naive_bayes import GaussianNB
    outdir = golden_models_dir()
    os.makedirs(outdir, exist_ok=True)
    ref = pickle.loads(open(os.path.join(REF, 'r95_twobase_model_NN_6_m6A.pkl'), 'rb').read(), encoding='latin')
    meta = {'sklearn': sklearn.__version__}
    for tag, make in (('LR', lambda: LogisticRegression(solver='liblinear', penalty='l1', random_state=5)), ('NBC', lambda: GaussianNB())):
        rng = np.random.default_rng(13)
        probes = np.round(np.concatenate([rng.normal(0, 2.5, size=(256, 6)), rng.uniform(6, 12, size=(256, 1))], axis=1), 4)
        probes[0, :6] = 40.0                               # (far out: the ends of expit / logsumexp)
        probes[1, :6] = -40.0
        models, ka = {}, {}
        for key in ('MG', 'MH'):
            X = np.concatenate([rng.normal(0, 2.5, size=(400, 6)), rng.uniform(6, 12, size=(400, 1))], axis=1)
            y = np.where(ref[key].predict_proba(X)[:, 1] + rng.normal(0, 0.15, size=400) >= 0.5, 'm6A', 'A')
            est = make()
            est.fit(X, y)
            assert list(est.classes_) == ['A', 'm6A']
            models[key] = est
            ka[key] = [float(v) for v in est.predict_proba(probes)[:, 1]]
        with open(os.path.join(outdir, 'simple_twobase_model_%s_6_m6A.pkl' % tag), 'wb') as fh:
            pickle.dump(models, fh, protocol=4)
        meta[tag] = dict(probes=[[float(v) for v in r] for r in probes], known_answers=ka, classes=[str(c) for c in models['MG'].classes_])
    json.dump(meta, open(os.path.join(outdir, 'simple_meta.json'), 'w'))


def load_weights(stem):
    """The arrays export_models() wrote: weights from OUT_PKG_MODELS, the sub-model list from models_meta.json."""
    import numpy as np
    z = np.load(os.path.join(OUT_PKG_MODELS, stem + '.npz'))
    meta = json.load(open(os.path.join(golden_models_dir(), 'models_meta.json')))[stem]
    out = {'__twobase__': meta['is_dict']}
    for key in meta['submodels']:
        out[key] = (z[key + '.W1'], z[key + '.b1'], z[key + '.W2'], z[key + '.b2'])
    return out


MODEL_FILES = {'r95': 'r95_twobase_model_NN_6_m6A', 'r94': 'r94_model_NN_6_m6A',
               'CAAY': 'CAAYNNNNNRTAC_model_6_m6A', 'CRAA': 'CRAANNNNNNNTGC_model_6_m6A'}


def run_reference_case(ec, rq, pos2label, case, d):
    """Run the reference's extract_features on a micro-case laid out in directory d."""
    a = case['args']
    tsv = os.path.join(d, 'case.eventalign.tsv')
    fasta = os.path.join(d, 'ref.fasta')
    fastq = os.path.join(d, 'reads.fastq')
    posf = os.path.join(d, 'positions.txt') if case['positions'] is not None else None
    open(tsv, 'w').write(case['tsv'])
    open(fasta, 'w').write(case['fasta'])
    open(fastq, 'w').write(case['fastq'])
    if posf:
        open(posf, 'w').write(case['positions'])
    stem = os.path.join(d, 'case.eventalign')
    for f in os.listdir(d):
        if '.tmp' in f:
            os.remove(os.path.join(d, f))
    read2qual = rq.extract_read_quality(fastq)
    pos_label = pos2label(posf) if (a['train'] and posf) else None
    modelfile = os.path.join(REF, MODEL_FILES[a['model']] + '.pkl')
    buf = io.StringIO()
    outcome, ret = 'ok', None
    try:
        with contextlib.redirect_stdout(buf):
            ret = ec.extract_features(tsv, fasta, read2qual, a['k'], a['skip_thresh'], a['qual_thresh'],
                                      modelfile, 'NN', 0, endline=os.path.getsize(tsv), train=a['train'],
                                      pos_label=pos_label, base=a['base'], motif=a['motif'],
                                      positions_list=posf)
    except SystemExit:
        outcome = 'exit'
    except Exception as e:                                   # noqa
        outcome = 'crash:' + type(e).__name__
    suffix = '.diffs.%d%s.tmp0' % (a['k'], '.train' if a['train'] else '')
    out_path = stem + suffix
    text = open(out_path).read() if os.path.exists(out_path) else None
    return dict(outcome=outcome, text=text, stdout=buf.getvalue().split('\n'), ret=ret)


def run_oracle_case(case, d):
    from oracle import py_oracle as po
    a = case['args']
    tsv = os.path.join(d, 'case.eventalign.tsv')
    fasta = os.path.join(d, 'ref.fasta')
    fastq = os.path.join(d, 'reads.fastq')
    posf = os.path.join(d, 'positions.txt') if case['positions'] is not None else None
    read2qual = po.read_fastq_quality(fastq)
    pos_label = None
    if a['train'] and posf:
        pos_label = {}
        for line in open(posf).read().split('\n'):
            t = line.split()
            if len(t) > 1:
                pos_label[(t[0], int(t[1]), t[2])] = t[3]
    models = None if a['train'] else load_weights(MODEL_FILES[a['model']])
    outcome, res = 'ok', None
    try:
        res = po.extract_features_oracle(tsv, fasta, read2qual, a['k'], a['skip_thresh'], a['qual_thresh'],
                                         models, 0, os.path.getsize(tsv), train=a['train'],
                                         pos_label=pos_label, base=a['base'], motif=a['motif'],
                                         positions_list=posf)
        if res['exit']:
            outcome = 'exit'
    except Exception as e:                                   # noqa
        outcome = 'crash:' + type(e).__name__
    text = None
    if res is not None:
        text = ''.join('\t'.join(r) + '\n' for r in res['written'])
    return dict(outcome=outcome, text=text, stdout=res['stdout'] if res else None, res=res)


def compare_case(ref, orc):
    """-> list of difference descriptions (empty = identical)."""
    diffs = []
    ref_bad = ref['outcome'] != 'ok'
    orc_bad = orc['outcome'] != 'ok'
    if ref_bad != orc_bad:
        diffs.append('outcome %s vs %s' % (ref['outcome'], orc['outcome']))
        return diffs
    rt = ref['text'] or ''
    ot = orc['text'] or ''
    if ref_bad:
        if orc['outcome'].startswith('crash'):
            return diffs      # a crash inside our restatement loses its buffered rows; class matched
        if rt != ot:
            diffs.append('partial output differs')
        return diffs
    if rt != ot:
        diffs.append('output text differs')
    rs = [l for l in ref['stdout'] if l.strip()]
    os_ = [l for l in orc['stdout'] if l.strip()]
    if rs != os_:
        diffs.append('stdout differs')
    if ref['ret'] is not None:
        sig, ctx = ref['ret']
        res = orc['res']
        if json.dumps(_plain(sig), sort_keys=True) != json.dumps(_plain(res['signals']), sort_keys=True):
            diffs.append('train signals differ')
        if json.dumps(ctx, sort_keys=True) != json.dumps(res['contexts'], sort_keys=True):
            diffs.append('train contexts differ')
    return diffs


def _plain(sig):
    return {k: {lab: [[(0 if isinstance(x, int) else repr(float(x))) for x in row] for row in rows]
                for lab, rows in v.items()} for k, v in sig.items()}


def main():
    global OUT_GOLDEN, OUT_PKG_MODELS
    n_micro = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
    if '--out' in sys.argv:                                  # a scratch copy of the repository's layout
        root = os.path.abspath(sys.argv[sys.argv.index('--out') + 1])
        OUT_GOLDEN = os.path.join(root, 'tests', 'golden')
        OUT_PKG_MODELS = os.path.join(root, 'mcaller_amd', 'models')
        os.makedirs(OUT_GOLDEN, exist_ok=True)
    scratch = tempfile.mkdtemp(prefix='mcaller_golden_')
    install_shims(scratch)
    import extract_contexts as ec
    import read_qual as rq
    from train_model import pos2label
    from oracle import casegen

    if '--rf-only' in sys.argv:
        export_rf_fixture()
        return
    if '--simple-only' in sys.argv:
        export_simple_fixture()
        return
    meta = export_models()
    export_rf_fixture()
    export_simple_fixture()
    report = {'models': {k: v['sha256'] for k, v in meta.items()}}

    # ------------------------------------------------------------------ testdata ------------
    td = os.path.join(scratch, 'testdata')
    shutil.copytree(os.path.join(REF, 'testdata'), td)
    for f in os.listdir(td):
        os.chmod(os.path.join(td, f), 0o644)
    os.chmod(td, 0o755)
    genome = rebuild_fasta(os.path.join(td, 'masonread1.eventalign.tsv'),
                           os.path.join(td, 'pb_ecoli_polished_assembly.fasta'), False)
    fix_td = os.path.join(OUT_GOLDEN, 'testdata')
    os.makedirs(fix_td, exist_ok=True)
    with open(os.path.join(td, 'masonread1.eventalign.tsv'), 'rb') as src, \
            gzip.GzipFile(os.path.join(fix_td, 'masonread1.eventalign.tsv.gz'), 'wb', mtime=0) as dst:
        dst.write(src.read())
    for f in ['masonread1.fastq', 'test_positions.txt', 'test_positions_A.txt', 'test_positions_m6A.txt',
              'masonread1.eventalign.diffs.6', 'masonread1.eventalign.diffs.6.train',
              'masonread1.methylation.summary.bed', 'pb_ecoli_polished_assembly.fasta.fai']:
        shutil.copy(os.path.join(REF, 'testdata', f), os.path.join(fix_td, f))
    # the span of the rebuilt FASTA that carries information (everything else is 'N')
    json.dump({'contig': 'ecoli', 'length': 4734145, 'span_start': 13100, 'span': genome[13100:26500]},
              open(os.path.join(fix_td, 'rebuilt_fasta_span.json'), 'w'))

    ref_out = os.path.join(OUT_GOLDEN, 'ref_outputs')
    os.makedirs(ref_out, exist_ok=True)
    model = os.path.join(REF, 'r95_twobase_model_NN_6_m6A.pkl')
    common = ['-r', os.path.join(td, 'pb_ecoli_polished_assembly.fasta'),
              '-e', os.path.join(td, 'masonread1.eventalign.tsv'),
              '-f', os.path.join(td, 'masonread1.fastq')]
    diffs = os.path.join(td, 'masonread1.eventalign.diffs.6')
    bed = os.path.join(td, 'masonread1.methylation.summary.bed')

    def bed_runs(tag):
        for extra, name in [([], 'bed'), (['--vo'], 'vo.bed')]:
            out, _ = run_cli('make_bed.py', ['-f', diffs, '-d', '1', '-t', '0.5'] + extra)
            shutil.copy(bed, os.path.join(ref_out, '%s.%s' % (tag, name)))

    cli_cases = [
        ('config1_positions_m6A', ['-p', os.path.join(td, 'test_positions_m6A.txt'), '-d', model]),
        ('motif_GATC', ['-m', 'GATC', '-d', model]),
        ('motif_A', ['-m', 'A', '-d', model]),
        ('positions_all', ['-p', os.path.join(td, 'test_positions.txt'), '-d', model]),
        ('motif_GATC_s1', ['-m', 'GATC', '-d', model, '-s', '1']),
        ('motif_A_r94', ['-m', 'A', '-d', os.path.join(REF, 'r94_model_NN_6_m6A.pkl')]),
    ]
    for tag, extra in cli_cases:
        if os.path.exists(diffs):
            os.remove(diffs)
        out, code = run_cli('mCaller.py', extra + common)
        shutil.copy(diffs, os.path.join(ref_out, tag + '.diffs.6'))
        open(os.path.join(ref_out, tag + '.stdout'), 'w').write(out.replace(td + '/', '<DIR>/'))
        if tag in ('config1_positions_m6A', 'motif_GATC'):
            bed_runs(tag)
    # the reference's own golden diffs -> its own golden BED (README.md:139)
    shutil.copy(os.path.join(REF, 'testdata', 'masonread1.eventalign.diffs.6'), diffs)
    bed_runs('reference_golden_diffs')
    os.remove(diffs)

    # --train: capture the labelled rows and the returned dicts (extract_features directly; the fit is
    # stochastic and out of scope)
    read2qual = rq.extract_read_quality(os.path.join(td, 'masonread1.fastq'))
    posf = os.path.join(td, 'test_positions.txt')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sig, ctx = ec.extract_features(os.path.join(td, 'masonread1.eventalign.tsv'),
                                       os.path.join(td, 'pb_ecoli_polished_assembly.fasta'), read2qual, 6, 0, 0,
                                       None, 'NN', 0, endline=os.path.getsize(os.path.join(td, 'masonread1.eventalign.tsv')),
                                       train=True, pos_label=pos2label(posf), base='A', motif=None,
                                       positions_list=posf)
    shutil.copy(os.path.join(td, 'masonread1.eventalign.diffs.6.train.tmp0'),
                os.path.join(ref_out, 'train_positions_all.diffs.6.train'))
    open(os.path.join(ref_out, 'train_positions_all.stdout'), 'w').write(buf.getvalue())
    json.dump({'signals': _plain(sig), 'contexts': ctx},
              open(os.path.join(ref_out, 'train_positions_all.dicts.json'), 'w'), indent=0)
    json.dump({'read2qual': {k: repr(float(v)) for k, v in read2qual.items()}},
              open(os.path.join(ref_out, 'read2qual.json'), 'w'))

    # oracle vs reference on the testdata runs (pin)
    from oracle import py_oracle as po
    pin_td = {}
    tsvp = os.path.join(td, 'masonread1.eventalign.tsv')
    fap = os.path.join(td, 'pb_ecoli_polished_assembly.fasta')
    r2q = po.read_fastq_quality(os.path.join(td, 'masonread1.fastq'))
    w95 = load_weights(MODEL_FILES['r95'])
    w94 = load_weights(MODEL_FILES['r94'])
    specs = [('config1_positions_m6A', dict(positions_list=os.path.join(td, 'test_positions_m6A.txt')), w95, 0),
             ('motif_GATC', dict(motif='GATC'), w95, 0), ('motif_A', dict(motif='A'), w95, 0),
             ('positions_all', dict(positions_list=posf), w95, 0),
             ('motif_GATC_s1', dict(motif='GATC'), w95, 1), ('motif_A_r94', dict(motif='A'), w94, 0)]
    for tag, kw, w, s in specs:
        res = po.extract_features_oracle(tsvp, fap, r2q, 6, s, 0, w, 0, os.path.getsize(tsvp), base='A', **kw)
        mine = ''.join('\t'.join(r) + '\n' for r in res['written'])
        theirs = open(os.path.join(ref_out, tag + '.diffs.6')).read()
        ref_stdout = [l for l in open(os.path.join(ref_out, tag + '.stdout')).read().split('\n')]
        counters_ok = all(l in ref_stdout for l in res['stdout'])
        pin_td[tag] = dict(rows=len(res['written']), identical=(mine == theirs), counters_ok=counters_ok)
    report['testdata_pin'] = pin_td

    # ------------------------------------------------------------------ micro-cases -----------
    kept, n_diff, by_flavour, outcomes, kept_by_flavour = [], 0, {}, {}, {}
    d = os.path.join(scratch, 'micro')
    os.makedirs(d)
    first_diffs = []
    for seed in range(n_micro):
        case = casegen.gen_case(seed)
        ref = run_reference_case(ec, rq, pos2label, case, d)
        orc = run_oracle_case(case, d)
        df = compare_case(ref, orc)
        fl = case['flavour']
        by_flavour.setdefault(fl, [0, 0])
        by_flavour[fl][0] += 1
        outcomes[ref['outcome']] = outcomes.get(ref['outcome'], 0) + 1
        if df:
            n_diff += 1
            by_flavour[fl][1] += 1
            if len(first_diffs) < 20:
                first_diffs.append(dict(seed=seed, flavour=fl, diffs=df))
        case['expected'] = dict(outcome=ref['outcome'], text=ref['text'],
                                stdout=[l for l in ref['stdout'] if l.strip()] if ref['outcome'] == 'ok' else None,
                                train=(dict(signals=_plain(ref['ret'][0]), contexts=ref['ret'][1])
                                       if ref['ret'] is not None else None))
        n_kept_fl = kept_by_flavour.get(fl, 0)
        if len(case['tsv']) < 24000 and n_kept_fl < 16:
            kept_by_flavour[fl] = n_kept_fl + 1
            kept.append(case)
    report['micro'] = dict(cases=n_micro, different=n_diff, by_flavour=by_flavour, outcomes=outcomes,
                           first_differences=first_diffs, kept=len(kept))
    with gzip.GzipFile(os.path.join(OUT_GOLDEN, 'micro_cases.json.gz'), 'wb', mtime=0) as fh:
        fh.write(json.dumps(kept).encode())
    json.dump(report, open(os.path.join(OUT_GOLDEN, 'PIN_REPORT.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: report[k] for k in ('testdata_pin',)}, indent=1))
    print(json.dumps({k: v for k, v in report['micro'].items() if k != 'by_flavour'}, indent=1))
    print(json.dumps(report['micro']['by_flavour']))
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
