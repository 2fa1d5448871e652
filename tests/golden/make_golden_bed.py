#!/usr/bin/env python3
"""More make_bed fixtures, made by RUNNING THE REFERENCE's make_bed.py in the build container (same stand-ins as
make_golden.py: our own SeqIO reader, an empty seaborn).  Never runs on the GPU box; no test imports this file.

Input: a multi-read `.diffs.6` built here from the committed `-m A` capture (three pseudo-reads per row, features and
labels perturbed deterministically) so that depth > 1, both labels occur at a site and the t-tests of `-p` mode have
more than one sample.  Outputs under tests/golden/bed_cases/: the input, a positions file, a small FASTA for --ref,
and the reference's output for every option combination the build offers.
"""
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg      # noqa: E402


def build_input(path):
    src = open(os.path.join(HERE, 'ref_outputs', 'motif_A.diffs.6')).read().strip().split('\n')
    rows = []
    for i, line in enumerate(src[:400]):
        chrom, read, pos, ctx, vals, strand, label, prob = line.split('\t')
        v = [float(x) for x in vals.split(',')]
        for r in range(3 if i % 5 else 1):                  # every fifth site keeps depth 1
            p = min(0.99, max(0.01, float(prob) + 0.17 * r - 0.1))
            vv = [repr(round(x + 0.37 * r * ((j % 3) - 1) + 0.011 * r * r * (j + 1) + 0.05 * r, 4)) for j, x in enumerate(v[:-1])] + [repr(v[-1])]
            rows.append('\t'.join([chrom, '%s_r%d' % (read, r), pos, ctx, ','.join(vv), strand,
                                   'm6A' if p >= 0.5 else 'A', repr(round(p, 2))]))
    open(path, 'w').write('\n'.join(rows) + '\n')
    return rows


def main():
    scratch = tempfile.mkdtemp(prefix='mcaller_golden_bed_')
    mg.install_shims(scratch)
    # --out DIR: write under DIR/tests/golden/bed_cases instead of into the repository (tests/test_pin_recipe.py)
    out_root = os.path.join(os.path.abspath(sys.argv[sys.argv.index('--out') + 1]), 'tests', 'golden') if '--out' in sys.argv else HERE
    out_dir = os.path.join(out_root, 'bed_cases')
    os.makedirs(out_dir, exist_ok=True)
    work = os.path.join(scratch, 'w')
    os.makedirs(work)
    diffs = os.path.join(work, 'multi.eventalign.diffs.6')
    rows = build_input(diffs)
    shutil.copy(diffs, os.path.join(out_dir, 'multi.eventalign.diffs.6'))
    # positions file of -p mode: chrom, start, end, strand (tab-separated, make_bed.py:13-19)
    sites = []
    for line in rows:
        t = line.split('\t')
        key = (t[0], t[2], str(int(t[2]) + 1), t[5])
        if key not in sites:
            sites.append(key)
    posf = os.path.join(work, 'bed_positions.txt')
    open(posf, 'w').write(''.join('\t'.join(s) + '\n' for s in sites[10:60:2]) + 'ecoli\t5\t6\t+\n')
    shutil.copy(posf, os.path.join(out_dir, 'bed_positions.txt'))
    # --ref: a FASTA whose coordinates cover the sites (N outside the span the test TSV describes)
    span = json.load(open(os.path.join(HERE, 'testdata', 'rebuilt_fasta_span.json')))
    fasta = os.path.join(work, 'ref.fasta')
    seq = 'N' * span['span_start'] + span['span'] + 'N' * 200
    open(fasta, 'w').write('>%s\n' % span['contig'] + '\n'.join(seq[i:i + 60] for i in range(0, len(seq), 60)) + '\n')

    cases = {
        'default_d1': ['-d', '1', '-t', '0.5'],
        'default_d3': ['-d', '3', '-t', '0.5'],
        'thresh_d2_t0.7': ['-d', '2', '-t', '0.7'],
        'control_d2': ['-d', '2', '-t', '0.5', '--control'],
        'vo_d2': ['-d', '2', '-t', '0.5', '--vo'],
        'gff_d2': ['-d', '2', '-t', '0.5', '--gff'],
        'gff_vo_d2': ['-d', '2', '-t', '0.5', '--gff', '--vo'],
        'ref_d2': ['-d', '2', '-t', '0.5', '--ref', fasta],
        'positions': ['-p', posf],
        'positions_vo': ['-p', posf, '--vo'],
    }
    cwd = os.getcwd()
    os.chdir(work)                                           # make_bed creates --plotdir in the cwd
    manifest = {}
    try:
        for tag, extra in cases.items():
            stdout, code = mg.run_cli('make_bed.py', ['-f', diffs] + extra)
            stem = 'multi.methylation' + ('.positions' if '-p' in extra else ('.control' if '--control' in extra else '')) + '.summary'
            produced = os.path.join(work, stem + ('.gff' if '--gff' in extra else '.bed'))
            name = tag + ('.gff' if '--gff' in extra else '.bed')
            shutil.move(produced, os.path.join(out_dir, name))
            tail = [l for l in stdout.strip().split('\n') if 'loci found' in l]
            manifest[tag] = {'args': [a if a not in (posf, fasta) else ('<POS>' if a == posf else '<REF>') for a in extra],
                             'output': name, 'stem': os.path.basename(produced), 'summary_line': tail[-1] if tail else None}
    finally:
        os.chdir(cwd)
    json.dump({'cases': manifest, 'ref_span': 'tests/golden/testdata/rebuilt_fasta_span.json'},
              open(os.path.join(out_dir, 'manifest.json'), 'w'), indent=1)
    print('wrote', len(manifest), 'make_bed cases to', out_dir)


if __name__ == '__main__':
    main()
