"""Print the tail of a rocprofv3 --kernel-trace [--memory-copy-trace] CSV set as a timeline (us)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0], 'q' + r.get('Queue_Id', '')))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', ''))), 'dma'))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = min(n, len(rows))
t0 = rows[-n][0]
for s, e, name, q in rows[-n:]:
    print('%9.1f %9.1f %7.1f  %s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
