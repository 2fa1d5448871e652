"""Where the host time of a file-to-file run goes: cProfile of the CLI on a synthetic file (after two warm-up runs).
  python tools/f2f_cprofile.py [rows] [motif]"""
import os, sys, tempfile, contextlib, io, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth, mCaller
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
d = tempfile.mkdtemp(prefix='mc_f2f_')
codes = synth.genome()
table, qual = synth.make_table(n, seed=5, codes=codes)
paths = synth.write_inputs(table, qual, codes, d)
model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
motif = sys.argv[2] if len(sys.argv) > 2 else 'GATC'
argv = ['-m', motif, '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model]
out = paths['tsv'][:-4] + '.diffs.6'
for rep in range(3):
    if os.path.exists(out):
        os.remove(out)
    with contextlib.redirect_stdout(io.StringIO()):
        if rep < 2:
            mCaller.main(argv)
        else:
            pr = cProfile.Profile()
            pr.enable()
            mCaller.main(argv)
            pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
