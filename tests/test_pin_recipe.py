"""The pin recipes run as committed: tests/golden/make_golden*.py import the reference from /root/reference, run it, and must
reproduce every committed fixture byte for byte (build container only: the reference never travels to the GPU box).

The scripts write into a scratch copy of the repository's layout (--out), so a run never touches the committed files."""
import gzip
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/testdata'), reason='needs the reference (build container only)')


def run_recipe(script, args, out):
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script)] + args + ['--out', str(out)],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def same_bytes(out, rel_dir, skip=()):
    made = os.path.join(str(out), rel_dir)
    names = sorted(os.listdir(made))
    assert names, rel_dir
    for name in names:
        if name in skip:
            continue
        a, b = os.path.join(made, name), os.path.join(REPO, rel_dir, name)
        assert os.path.exists(b), 'the recipe writes %s/%s, which is not committed' % (rel_dir, name)
        assert open(a, 'rb').read() == open(b, 'rb').read(), '%s/%s differs from the committed fixture' % (rel_dir, name)
    return names


def test_make_golden_reproduces_the_committed_fixtures(tmp_path):
    run_recipe('make_golden.py', ['50'], tmp_path)
    report = json.load(open(tmp_path / 'tests' / 'golden' / 'PIN_REPORT.json'))
    assert report['micro']['cases'] == 50 and report['micro']['different'] == 0, report['micro']
    assert all(v['identical'] and v['counters_ok'] for v in report['testdata_pin'].values()), report['testdata_pin']
    assert len(report['testdata_pin']) == 6
    assert report['models'] == json.load(open(os.path.join(GOLDEN, 'PIN_REPORT.json')))['models']
    assert len(same_bytes(tmp_path, 'tests/golden/ref_outputs')) == len(os.listdir(os.path.join(GOLDEN, 'ref_outputs')))
    assert len(same_bytes(tmp_path, 'mcaller_amd/models')) == 4
    same_bytes(tmp_path, 'tests/golden/models')
    same_bytes(tmp_path, 'tests/golden/testdata')
    # the micro-cases of the first 50 seeds that the committed file (made from 2000) also kept: the same cases, the same captures
    mine = {c['seed']: c for c in json.loads(gzip.open(tmp_path / 'tests' / 'golden' / 'micro_cases.json.gz').read())}
    committed = {c['seed']: c for c in json.loads(gzip.open(os.path.join(GOLDEN, 'micro_cases.json.gz')).read())}
    both = sorted(set(mine) & set(committed))
    assert len(both) >= 10
    for seed in both:
        assert mine[seed] == committed[seed], seed


def test_make_golden_bed_reproduces_the_committed_fixtures(tmp_path):
    run_recipe('make_golden_bed.py', [], tmp_path)
    assert len(same_bytes(tmp_path, 'tests/golden/bed_cases')) == len(os.listdir(os.path.join(GOLDEN, 'bed_cases')))


def test_make_golden_train_reproduces_the_committed_fixtures(tmp_path):
    pytest.importorskip('sklearn')
    run_recipe('make_golden_train.py', [], tmp_path)
    assert len(same_bytes(tmp_path, 'tests/golden/train')) == len(os.listdir(os.path.join(GOLDEN, 'train')))
