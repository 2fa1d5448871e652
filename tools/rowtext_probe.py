"""The device's row writer (mc_rowtext.hip) alone: ONE shard of a synthetic dense file parsed on the device, then the same pass
over it again and again with nothing else on the GPU -- what rocprofv3 --stats reports for the k_rt_* kernels under this script is
what they take by themselves (in a streamed run they share the GPU with the parser's kernels of the next shards and the copies).
  python tools/rowtext_probe.py [rows] [passes] [motif]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcaller_amd import synth, _lib
from mcaller_amd import extract_contexts as ec
from mcaller_amd.device import get_device
from mcaller_amd.read_qual import extract_read_quality
from mcaller_amd.refmark import MarkedReference

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
motif = sys.argv[3] if len(sys.argv) > 3 else 'A'
d = tempfile.mkdtemp(prefix='mc_rtp_')
codes = synth.genome()
table, qual = synth.make_table(rows, seed=5, codes=codes)
paths = synth.write_inputs(table, qual, codes, d)
model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mcaller_amd', 'models', 'r95_twobase_model_NN_6_m6A.npz')
modelset = ec.load_model_file(model)
r2q = extract_read_quality(paths['fastq'])
dev = get_device()
ref = MarkedReference(paths['fasta'], 'A', motif, None)
ref.quiet = True
_, weights, _, soc = ec.submodel_setup(modelset, 'A')
dev.set_classifier(weights, soc)
size = os.path.getsize(paths['tsv'])
dev.reserve_tables(size // 48 + 65536, (size // 48 + 65536) // 16, (size // 48 + 65536) // 16)
ref.mark(0)
dev.set_reference(ref.device_arrays())
text = _lib.TextBlock(paths['tsv'], 0, size)
slot = dev.parse_begin(text, ref.names, size // 48 + 65536)
t = dev.parse_end(slot, text)
assert t is not None, dev.parse_fallback_reason
P = ec.prepare_table(ec.Prepared(), t, ref, r2q, quiet=True)
dev.upload_table_async(P.table, P.qual)
dev.row_text(True, 'm6A', 'A')
got, t0 = None, time.perf_counter()
for i in range(passes):
    dev.run_async(6, 0, 0.0, tail_contig=0, score=True)
    rec = dev.wait()
    rt = getattr(rec, 'row_text', None)
    assert rt is not None, 'the pass came without rows'
    if got is None:
        got = (rt.n, rt.n_rows, bytes(rt.view[:200]))
    rt.release()
dt = (time.perf_counter() - t0) / passes
dev.row_text(False)
print('%d rows, -m %s: %d records -> %d rows of text, %d bytes; %.3f ms per pass with its rows (host clock, one pass at a time)' % (
    rows, motif, rec.n, got[1], got[0], dt * 1e3))
print(got[2].decode().splitlines()[0])
