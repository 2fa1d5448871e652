"""The streamed file-to-file path over and over: the .diffs.6 bytes of every run hashed and compared with the first run's and
with the one-table path's (MCALLER_NO_STREAM).   python tools/stream_soak.py [rows] [runs] [motif]"""
import contextlib, hashlib, io, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcaller_amd import synth, mCaller
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
motif = sys.argv[3] if len(sys.argv) > 3 else 'GATC'
d = tempfile.mkdtemp(prefix='mc_soak_')
codes = synth.genome()
table, qual = synth.make_table(n, seed=17, codes=codes)
paths = synth.write_inputs(table, qual, codes, d)
model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
argv = ['-m', motif, '-r', paths['fasta'], '-e', paths['tsv'], '-f', paths['fastq'], '-d', model]
out = paths['tsv'][:-4] + '.diffs.6'


def run_once(env):
    for k in ('MCALLER_NO_STREAM', 'MCALLER_HOST_PARSER', 'MCALLER_STREAM_SHARDS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    if os.path.exists(out):
        os.remove(out)
    with contextlib.redirect_stdout(io.StringIO()):
        mCaller.main(argv)
    return hashlib.sha1(open(out, 'rb').read()).hexdigest(), os.path.getsize(out)


want = run_once({'MCALLER_NO_STREAM': '1'})
bad = 0
for i in range(runs):
    env = {'MCALLER_STREAM_SHARDS': str(3 + i % 11)}
    if i % 7 == 3:
        env['MCALLER_HOST_PARSER'] = '1'
    got = run_once(env)
    bad += got != want
print('%d streamed runs of %d rows, -m %s (3..13 shards, every seventh through the host parser): %d differ from the one-table path (%d bytes)'
      % (runs, n, motif, bad, want[1]))
