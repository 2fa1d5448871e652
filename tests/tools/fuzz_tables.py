"""One-off campaign on the GPU box for the one-base-motif kernels (k1_scan<CG_DENSE>, k1_emit_runs): random synthetic TABLES --
sizes around the multiples of a chunk / a piece (1024 rows) and a tile (2048), reads from a dozen to thousands of events, every k
and skip_thresh, a quality threshold that filters reads -- each as a table's first pass (synchronous, validating), second, third
and, declared new, the validating pass again with two in flight, against the C oracle.  (The micro-cases of fuzz_gpu.py are a
few hundred rows: they never cross a chunk.)  usage: fuzz_tables.py [n_tables] [first seed] [dense|sparse]
("sparse": the same tables under motifs of three to five bases -- k1_scan<64>'s candidate lists, k1_emit;
"stalls": dense and sparse motifs over tables with stalls -- rows repeated 40-300 times: windows beyond what the emit looks back,
slots of more than 128 events -- and gaps of the model -- 64-400 filtered rows in a row --, SCORED, pipelined first: what the side
stream's one kernel finishes row by row for its own records, the rows the emit predicted for them, the special closer behind a gap)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcaller_amd import synth
from mcaller_amd.device import Device
from tests import helpers as H

n_tables = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000000
kind = sys.argv[3] if len(sys.argv) > 3 else 'dense'
MOTIFS = {'dense': ['A', 'A', 'A', 'C', 'AT', 'GA'],                                   # one-base motifs, and two-base ones that are dense too
          'sparse': ['GATC', 'GATC', 'CAG', 'TAC', 'GCAGC', 'ACGT', 'CCG'],
          'stalls': ['A', 'A', 'GATC', 'GATC', 'CAG', 'AT']}[kind]
if kind == 'stalls':
    from tests.test_gpu_parity import with_stalls
    from tests.test_gpu_fused import with_model_gaps
    from mcaller_amd.extract_contexts import submodel_setup
    WEIGHTS = submodel_setup(H.load_modelset('r95'), 'A')
dev = Device(0)
bad, done, rows, t0 = [], 0, 0, time.time()
for i in range(n_tables):
    seed = seed0 + i
    rng = np.random.default_rng(seed)
    codes = synth.genome(length=int(rng.integers(30000, 400000)), seed=seed)
    motif = MOTIFS[int(rng.integers(0, len(MOTIFS)))]
    base = 'A' if 'A' in motif else 'C'
    lo = int(rng.choice([8, 30, 120, 600, 3000]))
    read_len = (lo, lo * int(rng.integers(2, 12)))
    around = int(rng.choice([1024, 2048, 3072, 4096, 10240, 50000, 200000]))
    n = max(50, around + int(rng.integers(-40, 41)) if rng.random() < 0.7 else int(rng.integers(100, 300000)))
    k = int(rng.choice([6, 6, 6, 4, 5, 7, 8]))
    skip = int(rng.integers(0, min(3, (k - 1) // 2 + 1)))
    qthr = float(rng.choice([0.0, 0.0, 8.0, 10.5]))
    try:
        ref = synth.SynthRef(codes, base=base, motif=motif)
        table, qual = synth.make_table(n, seed=seed, codes=codes, read_len=read_len)
        arrays = ref.device_arrays()
        if kind == 'stalls':
            # (k = 6 and base A: the shipped model scores; reads long enough to hold a stall or a gap)
            k, base = 6, 'A'
            if n < 4000 or read_len[0] < 120:
                continue
            table = with_stalls(table, max(4, n // 400), (40, 70, 110, 127, 128, 129, 200, 300), seed=seed)
            if rng.random() < 0.6:
                table = with_model_gaps(table, max(2, n // 5000), seed=seed + 1)
            ref = synth.SynthRef(codes, base='A', motif=motif if 'A' in motif else 'A')
            arrays = ref.device_arrays()
            _, weights, _, soc = WEIGHTS
            orc = H.oracle_records(table, arrays, qual, k, skip, qthr)
            H.oracle_score(orc, table, qual, weights, soc, k)
            dev.set_reference(arrays)
            dev.set_mlp(weights, soc)
            # (few workgroups, forced, in two tables out of three: every one of them walks many stretches -- stretches of nothing among them
            # where long reads lie under the quality threshold)
            g = int(rng.choice([0, 1, 2, 3, 5, 16]))
            if g:
                os.environ['MCALLER_SIDE_GRID'] = str(g)
            else:
                os.environ.pop('MCALLER_SIDE_GRID', None)
            if rng.random() < 0.5:
                qual = qual.copy()
                qual[rng.random(len(qual)) < 0.5] = 5.0
                qthr = 8.0
                orc = H.oracle_records(table, arrays, qual, k, skip, qthr)
                H.oracle_score(orc, table, qual, weights, soc, k)
            slot = dev.upload_table_async(table, qual)
            for again in range(3):                                   # first pass pipelined and validating, a later one, two in flight declared new
                if again == 2:
                    dev.select_table(slot, as_new=True)
                    dev.run_async(k, skip, qthr, score=True)
                dev.run_async(k, skip, qthr, score=True)
                for _ in range(2 if again == 2 else 1):
                    H.assert_records_equal(dev.wait(), orc, k, prob_tol=1e-6)
            done += 1
            rows += table.n_rows
            continue
        orc = H.oracle_records(table, arrays, qual, k, skip, qthr)
        dev.set_reference(arrays)
        dev.upload_table(table)
        dev.set_read_quality(qual)
        rec = dev.extract(k, skip, qthr, score=False)
        rec.prob[:rec.n] = np.nan
        H.assert_records_equal(rec, orc, k)
        slot = dev.current_slot()
        for again in range(4):
            if again == 2:
                dev.select_table(slot, as_new=True)
            dev.run_async(k, skip, qthr, score=False)
            if again == 2:
                continue
            for _ in range(2 if again == 3 else 1):
                rec2 = dev.wait()
                rec2.prob[:rec2.n] = np.nan
                H.assert_records_equal(rec2, orc, k)
        done += 1
        rows += table.n_rows
    except AssertionError as e:
        bad.append((seed, motif, read_len, n, k, skip, qthr, str(e)[:160]))
    except Exception as e:
        bad.append((seed, motif, read_len, n, k, skip, qthr, 'ERROR %s: %s' % (type(e).__name__, str(e)[:160])))
print('%d tables (%d rows) compared in %.0f s, %d differ' % (done, rows, time.time() - t0, len(bad)))
for b in bad[:20]:
    print(b)
