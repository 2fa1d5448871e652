"""Campaign for the rows written on the GPU (mc_rowtext.hip): many small eventalign files (tests/test_gpu_rowtext.write_case: three
contigs, both strands, reads at the contigs' ends, skipped positions, NNNNNN rows, two to four decimals, qualities with many digits),
each through the CLI as ONE table (the host formatter) and streamed in 2..9 shards with the device's row writer, under four motifs;
the bytes must be the same.   python tests/tools/fuzz_rowtext.py [cases] [first seed]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_rowtext as T                                                     # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 41000000
bad = with_text = without = rows_total = 0
for i in range(cases):
    seed = seed0 + i
    d = tempfile.mkdtemp(prefix='mc_rt_fuzz_')
    decimals = [(2,), (2, 2, 4), (4,), (2, 3)][i % 4]
    paths, rows = T.write_case(d, seed, n_reads=12 + (seed % 5) * 9, decimals=decimals, edge_reads=(i % 3 != 0))
    for motif in ('A', 'GATC', 'AT', 'AA'):
        want, _, _ = T.run_cli(paths, motif, {'MCALLER_NO_STREAM': '1'})
        got, n_dev, n = T.run_cli(paths, motif, {'MCALLER_STREAM_SHARDS': str(2 + (seed + len(motif)) % 8)})
        with_text += n_dev
        without += n - n_dev
        rows_total += want.count(b'\n')
        if got != want:
            bad += 1
            print('DIFFERENT: seed %d motif %s (%d shards, %d with text)' % (seed, motif, n, n_dev))
    for f in paths.values():
        os.remove(f)
print('%d cases x 4 motifs (A, GATC, AT, AA -- the masks of the last made by the host, contig by contig), %d rows of output: %d differ from the one-table path; %d shards had their rows written on the GPU, %d came '
      'from the host formatter (a context that leaves its contig)' % (cases, rows_total, bad, with_text, without))
sys.exit(1 if bad else 0)
