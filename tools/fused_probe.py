"""Kernel time of the fused dense pass (k1_fused) for every library variant under mcaller_amd/variants/ whose name starts with fd_
(tools/variants.sh), and for the default build: hipEvents around the pass of pipelined passes, one at a time, a table's first
pass (validating) and later ones; records hashed against the default build's.   python tools/fused_probe.py [rows]"""
import hashlib, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if len(sys.argv) > 1 and sys.argv[1] == '--one':
    sys.path.insert(0, REPO)
    import numpy as np
    from mcaller_amd import synth
    from mcaller_amd.device import Device
    n = int(float(sys.argv[2]))
    codes = synth.genome()
    ref = synth.SynthRef(codes, motif='A')
    table, qual = synth.make_table(n, seed=1000, codes=codes)
    dev = Device(0)
    dev.set_reference(ref.device_arrays())
    slot = dev.upload_table_async(table, qual)
    dev.wait_upload(slot)
    dev.set_pass_timing(1)
    out = {}
    for kind, as_new in (('validating', True), ('later', False)):
        ts = []
        for it in range(8):
            if as_new:
                dev.select_table(slot, as_new=True)
            dev.run_async(6, 0, 0.0, score=False)
            rec = dev.wait()
            ts.append(dev.times_ms()['window_scan'] + dev.times_ms()['emit'])
        out[kind] = float(np.median(ts[2:])) * 1e3
    r = rec.by_record()
    out['hash'] = hashlib.sha1(r.feats[:r.n * 6].tobytes() + r.info[:r.n].tobytes() + r.close_row[:r.n].tobytes()).hexdigest()[:12]
    out['info'] = dev.last_pass_info()
    print(json.dumps(out))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else '1e8'
libs = [None] + sorted(os.path.join(REPO, 'mcaller_amd', 'variants', f) for f in os.listdir(os.path.join(REPO, 'mcaller_amd', 'variants')) if f.startswith('fd_') and f != 'fd_trace.so')
for lib in libs:
    env = dict(os.environ)
    if lib:
        env['MCALLER_LIB'] = lib
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--one', n], capture_output=True, text=True, env=env, timeout=600)
    tag = os.path.basename(lib) if lib else 'default'
    print('%-16s %s' % (tag, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else 'FAILED ' + r.stderr[-300:]))
