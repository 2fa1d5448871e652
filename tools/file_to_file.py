"""File-to-file timing (third timing): synthetic eventalign TSV -> .diffs.6 through the CLI."""
import os, sys, time, tempfile, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, mCaller

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
d = tempfile.mkdtemp(prefix='mc_f2f_')
codes = synth.genome()
table, qual = synth.make_table(n, seed=5, codes=codes)
t = time.time()
tsv = os.path.join(d, 'syn.eventalign.tsv')
synth.write_tsv(table, codes, tsv)
with open(os.path.join(d, 'ref.fasta'), 'w') as fa:
    s = synth.codes_to_str(codes)
    fa.write('>ecoli_syn\n' + '\n'.join(s[i:i + 60] for i in range(0, len(s), 60)) + '\n')
with open(os.path.join(d, 'reads.fastq'), 'w') as fq:
    for i, name in enumerate(table.read_names):
        q = int(round(qual[i]))
        fq.write('@%s\nACGTACGTAC\n+\n%s\n' % (name, chr(33 + q) * 10))
print('inputs written in %.1f s: %.1f MB of TSV, %d rows' % (time.time() - t, os.path.getsize(tsv) / 1e6, table.n_rows))
os.environ['MCALLER_TIMING'] = os.environ.get('MCALLER_TIMING', '1')
model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
for rep in range(6):
    out = tsv[:-4] + '.diffs.6'
    if os.path.exists(out):
        os.remove(out)
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        mCaller.main(['-m', 'GATC', '-r', os.path.join(d, 'ref.fasta'), '-e', tsv, '-f', os.path.join(d, 'reads.fastq'), '-d', model])
    dt = time.perf_counter() - t
    calls = sum(1 for _ in open(out))
    print('run %d: %.3f s wall, %d calls -> %.3g events/s, %.3g calls/s (file to file)' % (rep, dt, calls, table.n_rows / dt, calls / dt))
