"""make_bed-compatible writer against the reference's BEDs (its own golden + the captures made in the build container)."""
import contextlib
import io
import os
import shutil

import pytest

from tests import helpers as H


@pytest.mark.parametrize('diffs,bed,vo', [
    ('testdata/masonread1.eventalign.diffs.6', 'testdata/masonread1.methylation.summary.bed', False),
    ('testdata/masonread1.eventalign.diffs.6', 'ref_outputs/reference_golden_diffs.vo.bed', True),
    ('ref_outputs/config1_positions_m6A.diffs.6', 'ref_outputs/config1_positions_m6A.bed', False),
    ('ref_outputs/motif_GATC.diffs.6', 'ref_outputs/motif_GATC.bed', False),
    ('ref_outputs/motif_GATC.diffs.6', 'ref_outputs/motif_GATC.vo.bed', True),
])
def test_bed_bytes(tmp_path, diffs, bed, vo):
    from mcaller_amd import make_bed
    src = str(tmp_path / 'masonread1.eventalign.diffs.6')
    shutil.copy(os.path.join(H.GOLDEN, diffs), src)
    with contextlib.redirect_stdout(io.StringIO()):
        make_bed.main(['-f', src, '-d', '1', '-t', '0.5'] + (['--vo'] if vo else []))
    out = str(tmp_path / 'masonread1.methylation.summary.bed')
    assert open(out).read() == open(os.path.join(H.GOLDEN, bed)).read()


def _bed_cases():
    import json
    return json.load(open(os.path.join(H.GOLDEN, 'bed_cases', 'manifest.json')))['cases']


@pytest.mark.parametrize('tag', sorted(_bed_cases()))
def test_bed_option_matrix(tmp_path, tag):
    """-d/-t, --control, --vo, --gff (+ --vo), --ref and -p (t-test columns) on a multi-read diffs file, against the outputs
    of the reference's make_bed.py (tests/golden/make_golden_bed.py)."""
    import json
    import warnings
    from mcaller_amd import make_bed
    case = _bed_cases()[tag]
    src = str(tmp_path / 'multi.eventalign.diffs.6')
    shutil.copy(os.path.join(H.GOLDEN, 'bed_cases', 'multi.eventalign.diffs.6'), src)
    span = json.load(open(os.path.join(H.GOLDEN, 'testdata', 'rebuilt_fasta_span.json')))
    fasta = str(tmp_path / 'ref.fasta')
    open(fasta, 'w').write('>%s\n%s\n' % (span['contig'], 'N' * span['span_start'] + span['span'] + 'N' * 200))
    args = [{'<POS>': os.path.join(H.GOLDEN, 'bed_cases', 'bed_positions.txt'), '<REF>': fasta}.get(a, a) for a in case['args']]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        make_bed.main(['-f', src] + args)
    assert open(str(tmp_path / case['stem'])).read() == open(os.path.join(H.GOLDEN, 'bed_cases', case['output'])).read()
    if case['summary_line']:
        assert case['summary_line'] in buf.getvalue()
