"""oracle/py_oracle.py against the committed captures of the reference (tests/golden/, made by make_golden.py)."""
import os

from tests import helpers as H
from oracle import py_oracle as po


def weights(tag):
    ms = H.load_modelset(tag)
    out = {'__twobase__': ms.twobase}
    for key, w in ms.models.items():
        out[key] = (w.W1, w.b1, w.W2.reshape(-1, 1), w.b2)
    return out


def test_testdata_runs(tmp_path):
    td = H.testdata_paths(str(tmp_path))
    r2q = po.read_fastq_quality(td['fastq'])
    size = os.path.getsize(td['tsv'])
    for tag, kw, model, skip in [('config1_positions_m6A', dict(positions_list=td['test_positions_m6A.txt']), 'r95', 0),
                                 ('motif_GATC', dict(motif='GATC'), 'r95', 0),
                                 ('motif_GATC_s1', dict(motif='GATC'), 'r95', 1)]:
        res = po.extract_features_oracle(td['tsv'], td['fasta'], r2q, 6, skip, 0, weights(model), 0, size, base='A', **kw)
        text = ''.join('\t'.join(r) + '\n' for r in res['written'])
        assert text == open(os.path.join(H.GOLDEN, 'ref_outputs', tag + '.diffs.6')).read()


def test_micro_cases(tmp_path):
    bad = 0
    cases = H.micro_cases()
    for case in cases:
        d = tmp_path / ('c%d' % case['seed'])
        d.mkdir()
        paths = H.materialise(case, str(d))
        a = case['args']
        exp = case['expected']
        try:
            r2q = po.read_fastq_quality(paths['fastq'])
            res = po.extract_features_oracle(paths['tsv'], paths['fasta'], r2q, a['k'], a['skip_thresh'], a['qual_thresh'],
                                             None if a['train'] else weights(a['model']), 0,
                                             os.path.getsize(paths['tsv']), train=a['train'],
                                             pos_label=H.pos2label(paths['positions']) if (a['train'] and paths['positions']) else None,
                                             base=a['base'], motif=a['motif'], positions_list=paths['positions'])
            outcome = 'exit' if res['exit'] else 'ok'
            text = ''.join('\t'.join(r) + '\n' for r in res['written'])
        except Exception as e:              # noqa
            outcome, text = 'crash', None
        if (exp['outcome'] == 'ok') != (outcome == 'ok'):
            bad += 1
        elif outcome == 'ok' and (text != (exp['text'] or '') or [l for l in res['stdout'] if l.strip()] != exp['stdout']):
            bad += 1
    assert bad == 0, '%d of %d' % (bad, len(cases))
