"""One-off fuzz campaign on the GPU box: random micro-cases (oracle/casegen.py, seeds outside every committed fixture) through
the HIP path -- synchronous and pipelined, as a table's first, second and third pass -- against the C oracle.  Prints the seeds that differ (none expected)."""
import contextlib, io, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import casegen
from mcaller_amd import extract_contexts as ec
from mcaller_amd.device import Device
from mcaller_amd.read_qual import extract_read_quality
from tests import helpers as H

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 300
base_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
flavours = ['quirk_pal', 'plain', 'dense', 'skips', 'heavy', 'multi_contig', 'qual', 'quirk_names', 'quirk_flip',
            'quirk_backwards', 'quirk_pos0', 'header', 'n_context', 'positions', 'basec', 'bare_model', 'guppy_model']
dev = Device(0)
bad, done, t0 = [], 0, time.time()
root = tempfile.mkdtemp(prefix='fuzz_')
for fi, fl in enumerate(flavours):
    for i in range(n_per):
        seed = base_seed + fi * 100000 + i
        case = casegen.gen_case(seed, flavour=fl)
        d = os.path.join(root, 'c')
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        paths = H.materialise(case, d)
        a = case['args']
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                P = ec.prepare(paths['tsv'], paths['fasta'], extract_read_quality(paths['fastq']), 0, os.path.getsize(paths['tsv']),
                               a['base'], a['motif'], paths['positions'])
        except BaseException:
            continue
        if P.table.n_rows == 0:
            continue
        train = a['train']
        modelset = None if train else H.load_modelset(a['model'])
        try:
            if not train:
                _, w, _, soc = ec.submodel_setup(modelset, a['base'])
                if w[0].n_in != a['k'] + 1:
                    continue
            orc = H.oracle_records(P.table, P.ref.device_arrays(), P.qual, a['k'], a['skip_thresh'], a['qual_thresh'])
            if not train:
                H.oracle_score(orc, P.table, P.qual, w, soc, a['k'])
            rec = ec.compute(P, a['k'], a['skip_thresh'], a['qual_thresh'], modelset, a['base'], train, device=dev)
            H.assert_records_equal(rec, orc, a['k'])
            # the same table again through the pipelined interface: second pass (positions streamed), third (unit summaries), and
            # -- declared new -- the validating first pass itself (two in flight)
            slot = dev.current_slot()
            for again in range(4):
                if again == 2:
                    dev.select_table(slot, as_new=True)
                dev.run_async(a['k'], a['skip_thresh'], a['qual_thresh'], score=not train)
                if again == 2:
                    continue
                for _ in range(2 if again == 3 else 1):
                    rec2 = dev.wait()                 # (compacted view: the helper checks call_row, then compares by record)
                    H.assert_records_equal(rec2, orc, a['k'])
            done += 1
        except AssertionError as e:
            bad.append((seed, fl, str(e)[:200]))
        except Exception as e:
            bad.append((seed, fl, 'ERROR %s: %s' % (type(e).__name__, str(e)[:200])))
print('%d cases compared in %.0f s, %d differ' % (done, time.time() - t0, len(bad)))
for b in bad[:20]:
    print(b)
