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
