"""rocprofv3 --pmc passes -> profiles/rNN_pmc.json.

Usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload text] [commit]

Each pass is `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -d <dir> -- python3 bench.py ...`
(separate passes: the two counters do not fit the TCC slots together, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Correction applied (same guide, HBM section): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming
read -> x2; WRITE_SIZE is taken as reported.  Calibration inside the same run: k_validate streams exactly 8 B/row (the
position and event-index columns, 0.8 GB at 10^8 rows) and writes nothing but a few flag words -- its corrected figure
is stored as `calibration` and must come out at ~1.0.
Both counters are in KB."""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0].split('<')[0]


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
    fe = sum(per[kname]['hbm_bytes'] for kname in ('k1_scan', 'k1_group_scan', 'k1_list', 'k1_emit') if kname in per)
    cal = None
    if 'k_validate' in per:
        cal = {'kernel': 'k_validate', 'expected_fetch_bytes': 8.0e8, 'fetch_bytes_corrected': per['k_validate']['fetch_bytes_corrected'],
               'ratio': per['k_validate']['fetch_bytes_corrected'] / 8.0e8, 'note': 'expected value holds for the 10^8-row workload'}
    json.dump({'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write, 'per_launch_bytes_corrected': per, 'workload': workload,
               'head': head, 'calibration': cal,
               'feature_extraction_hbm_bytes_per_step': fe}, open(out, 'w'), indent=1)
    print('feature extraction: %.1f MB of HBM traffic per step' % (fe / 1e6))


if __name__ == '__main__':
    main()
