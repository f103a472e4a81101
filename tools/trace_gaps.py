"""Busy / idle analysis of a rocprofv3 --kernel-trace CSV: per queue busy time, union busy time, idle gaps, top kernels.
usage: trace_gaps.py <kernel_trace.csv> [steps]  -- analyses the last `frac` of the trace window (the timed steps)."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'\(.*', '', n)[:60]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), n))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
rows = [r for r in rows if r[0] >= lo]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print('window %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(rows)))
byq = defaultdict(list)
for r in rows:
    byq[r[2]].append(r)
for q, rs in byq.items():
    busy = sum(e - s for s, e, _, _ in rs)
    print('queue %s: %d kernels busy %.2f ms' % (q, len(rs), busy / 1e6))
# union busy
cur_s, cur_e, union = None, None, 0
gaps = []
for s, e, q, n in rows:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print('union busy %.2f ms, idle %.2f ms' % (union / 1e6, (t1 - t0 - union) / 1e6))
g = defaultdict(lambda: [0, 0])
for d, n in gaps:
    g[n][0] += 1
    g[n][1] += d
print('idle before kernel (top):')
for n, (c, d) in sorted(g.items(), key=lambda kv: -kv[1][1])[:15]:
    print('  %-60s n=%5d idle %.2f ms' % (n, c, d / 1e6))
agg = defaultdict(lambda: [0, 0])
for s, e, q, n in rows:
    agg[n][0] += 1
    agg[n][1] += e - s
print('top kernels:')
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('  %-60s n=%6d  %.2f ms  avg %.1f us' % (n, c, d / 1e6, d / c / 1e3))
