"""Times the per-pass kernels (hipEvents of the synchronous full pass) for every library variant under mcaller_amd/variants/
(tools/variants.sh), one subprocess each: the same 10^8-row table, records checked against the default build's."""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if len(sys.argv) > 1 and sys.argv[1] == '--one':
    sys.path.insert(0, REPO)
    import numpy as np
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    from mcaller_amd.extract_contexts import submodel_setup
    from mcaller_amd.model_io import load_model_file, shipped_model
    n, motif = int(float(sys.argv[2])), sys.argv[3]
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif=motif)
    table, qual = synth.make_table(n, seed=1000, codes=codes)
    _, weights, _, soc = submodel_setup(load_model_file(shipped_model()), 'A')
    dev = Device(0)
    dev.set_reference(ref.device_arrays()); dev.set_mlp(weights, soc)
    slot = dev.upload_table_async(table, qual); dev.wait_upload(slot)
    ts, ts2 = [], []
    for it in range(10):                   # full passes: the table declared new every time (the first pass over a table)
        dev.select_table(slot, as_new=True)
        n_rec = dev.run(6, 0, 0.0)
        ts.append(dev.times_ms())
    rec = dev.fetch()
    import hashlib
    h = hashlib.sha1(rec.feats[:rec.n * 6].tobytes() + rec.info[:rec.n].tobytes() + rec.close_row[:rec.n].tobytes()).hexdigest()[:12]
    for it in range(8):                    # later passes over the validated table (the third builds the unit summaries)
        dev.run(6, 0, 0.0)
        ts2.append(dev.times_ms())
    # pipelined rate of full passes
    import time

    def enqueue():
        dev.select_table(slot, as_new=True)
        dev.run_async(6, 0, 0.0)
    for _ in range(3): enqueue()
    t0 = None
    for i in range(60):
        if i == 10: dev.sync(); t0 = time.perf_counter(); k0 = i
        dev.wait(); enqueue()
    for _ in range(3): dev.wait()
    dev.sync()
    per = (time.perf_counter() - t0) / (60 - 10 + 3) * 1e3
    pr = np.asarray(rec.prob[:rec.n], dtype=np.float64)
    print(json.dumps(dict(pipelined_ms=round(per, 4), records=int(n_rec), sha=h, prob_sum=repr(float(np.nansum(pr))),
                          **{k: round(float(np.median([t[k] for t in ts[3:]])), 4) for k in ts[0]},
                          **{'rescan_' + k: round(float(np.median([t[k] for t in ts2[4:]])), 4) for k in ('strand_resolve', 'window_scan')})))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else '1e8'
motif = sys.argv[2] if len(sys.argv) > 2 else 'GATC'
vdir = os.path.join(REPO, 'mcaller_amd', 'variants')
libs = [('default', None)] + [(f[:-3], os.path.join(vdir, f)) for f in sorted(os.listdir(vdir)) if f.endswith('.so')] if os.path.isdir(vdir) else [('default', None)]
only = os.environ.get('VARIANTS')
for tag, path in libs:
    if only and tag not in only.split(',') and tag != 'default':
        continue
    env = dict(os.environ)
    if path:
        env['MCALLER_LIB'] = path
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--one', n, motif], env=env, capture_output=True, text=True, timeout=600)
    print('%-22s %s' % (tag, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else 'FAILED: ' + r.stderr[-400:]), flush=True)
