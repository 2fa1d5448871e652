"""Per-wave view of tools/sq_counters.sh's output: python tools/sq_show.py k1_scan k1_emit ..."""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/sq/p*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace('k1_dense<true, false>', 'k1_dense_emit').replace('k1_dense<false, false>', 'k1_dense_count').replace('k1_dense<false, true>', 'k1_dense_countv').split('<')[0]
        per[(k, row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
    for (k, d, c), v in per.items():
        agg[k][c].append(v)
for k in sys.argv[1:]:
    if k not in agg:
        print(k, 'not found among', sorted(agg)); continue
    w = max(agg[k]['SQ_WAVES'])
    print(k, 'waves per launch (max)', w)
    for c in sorted(agg[k]):
        vs = agg[k][c]
        print('   %-24s max %14.0f  per wave %.1f' % (c, max(vs), max(vs) / w))
