"""rocprofv3 --pmc passes -> profiles/rNN_pmc.json.

Usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload text] [commit]

Each pass is `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -d <dir> -- python3 bench.py ...`
(separate passes: the two counters do not fit the TCC slots together, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Correction applied (same guide, HBM section): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming
read -> x2; WRITE_SIZE is taken as reported.  Calibration inside the same run: the validating scan (k1_scan<64,0>) streams
exactly 9 B/row (positions, event indices, flag bytes: 0.9 GB at 10^8 rows) plus the mask words of its units (L2) -- its
corrected figure is stored as `calibration` and must come out slightly above 1.0.
Both counters are in KB.  The file records the hash of the kernel sources it was collected on (`kernel_source_sha16`): bench.py
quotes `roofline.traffic` from it only while the library is built from the same sources."""
import csv
import glob
import json
import os
import sys


def short(name):
    """Kernel name without namespace and arguments; the scan keeps its template arguments (k1_scan<64,0>: the validating
    first pass; <64,1> / <64,2>: repeated passes over a validated table)."""
    name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if name.startswith('k1_scan') or name.startswith('k2_mlp'):    # (k2_mlp<NI, FAST, PACK, ROOMY, TH>: PACK = the side stream's one kernel)
        return name.replace(' ', '')
    return name.split('<')[0]


PER_TABLE_KERNELS = ['k0_first_site', 'k1_scan<64,0>', 'k1_group_scan', 'k1_list', 'k1_emit']     # every kernel that touches a table once
DENSE_PER_TABLE_KERNELS = ['k0_first_site', 'k1_scan<130,0>', 'k1_group_scan', 'k1_list', 'k1_emit_runs']
FUSED_PER_TABLE_KERNELS = ['k0_first_site', 'k1_fused']       # a pipelined pass over a dense reference


def kernel_source_hash():
    import hashlib
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ('mcaller_amd/csrc/mc_dev.h', 'mcaller_amd/csrc/mc_rows.h', 'mcaller_amd/csrc/mc_rowtext.h', 'mcaller_amd/csrc/mc_rowtext.hip', 'mcaller_amd/csrc/mc_k0.hip', 'mcaller_amd/csrc/mc_scan.hip', 'mcaller_amd/csrc/mc_emit.hip',
                'mcaller_amd/csrc/mc_fused.hip', 'mcaller_amd/csrc/mc_literal.hip', 'mcaller_amd/csrc/mc_classify.hip', 'mcaller_amd/csrc/mc_stream.hip',
                'mcaller_amd/csrc/mc_devparse.inc'):     # (bench.KERNEL_SOURCES, in its order)
        with open(os.path.join(repo, rel), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def collect(d, counter):
    per = {}
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                per.setdefault(short(row['Kernel_Name']), {}).setdefault(row['Dispatch_Id'], 0.0)
                per[short(row['Kernel_Name'])][row['Dispatch_Id']] += float(row['Counter_Value'])
    return {k: {'launches': len(v), 'mean': sum(v.values()) / len(v)} for k, v in per.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else ''
    head = sys.argv[5] if len(sys.argv) > 5 else 'unknown'
    fetch, write = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    per = {}
    for kname in fetch:
        fb = fetch[kname]['mean'] * 1024.0 * 2.0
        wb = write.get(kname, {'mean': 0.0})['mean'] * 1024.0
        per[kname] = {'fetch_bytes_corrected': fb, 'write_bytes': wb, 'hbm_bytes': fb + wb}
    names = PER_TABLE_KERNELS if 'k1_scan<64,0>' in per else DENSE_PER_TABLE_KERNELS
    pair_bytes = sum(per[kname]['hbm_bytes'] for kname in DENSE_PER_TABLE_KERNELS if kname in per) if 'k1_fused' in per else None
    if 'k1_fused' in per:
        names = FUSED_PER_TABLE_KERNELS
    fe = sum(per[kname]['hbm_bytes'] for kname in names if kname in per)
    cal = None
    if 'k1_scan<64,0>' in per:
        cal = {'kernel': 'k1_scan<64,0>', 'expected_fetch_bytes': 9.0e8, 'fetch_bytes_corrected': per['k1_scan<64,0>']['fetch_bytes_corrected'],
               'ratio': per['k1_scan<64,0>']['fetch_bytes_corrected'] / 9.0e8,
               'note': 'expected value holds for the 10^8-row workload: 9 B/row streamed; the mask words and the descriptors come on top'}
    json.dump({'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write, 'per_launch_bytes_corrected': per, 'workload': workload,
               'head': head, 'kernel_source_sha16': kernel_source_hash(), 'per_table_kernels': [n for n in names if n in per],
               'calibration': cal, 'per_table_hbm_bytes': fe,
               'per_table_hbm_bytes_of_the_scan_and_emit_pair': pair_bytes}, open(out, 'w'), indent=1)
    print('every kernel that touches a table once: %.1f MB of HBM traffic per table' % (fe / 1e6))


if __name__ == '__main__':
    main()
