"""Per-queue activity over time from a rocprofv3 --kernel-trace CSV: for every 5 ms bin the busy milliseconds of each HIP queue.
usage: python tools/queue_overlap.py <kernel_trace.csv> [bin_ms] [last_fraction]"""
import csv
import sys
from collections import defaultdict

bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id'])))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * (float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
rows = [r for r in rows if r[0] >= lo]
t0 = rows[0][0]
qs = sorted(set(r[2] for r in rows))
bins = defaultdict(lambda: defaultdict(float))
w = bin_ms * 1e6
for s, e, q in rows:
    b = int((s - t0) // w)
    while s < e:
        edge = t0 + (b + 1) * w
        seg = min(e, edge) - s
        bins[b][q] += seg / 1e6
        s += seg
        b += 1
print('bin(ms)   ' + '  '.join('q%-5d' % q for q in qs))
for b in sorted(bins):
    print('%7.1f   ' % (b * bin_ms) + '  '.join('%6.2f' % bins[b].get(q, 0.0) for q in qs))
