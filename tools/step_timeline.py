"""One training step of a rocprofv3 --kernel-trace CSV as a timeline: the step between the last two optimizer launches, per HIP queue
busy time, when each queue starts / ends relative to the step, the largest kernel families per queue and the tail after the main queue's
last kernel.   usage: python tools/step_timeline.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'\(.*', '', n)
        n = re.sub(r'^void ', '', n)[:52]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), n))
rows.sort()
opt = [i for i, r in enumerate(rows) if r[3].startswith('optim_group_kernel') or r[3].startswith('optim_dyn_kernel')]
# two optimizer launches per step (two groups): step k ends at the second
ends = opt[1::2]
a, b = ends[-2], ends[-1]
step = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
print('step window %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
byq = defaultdict(list)
for r in step:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    busy = sum(e - s for s, e, _, _ in rs)
    print('\nqueue %d: %d kernels, busy %.2f ms, first at +%.2f ms, last ends at +%.2f ms' % (q, len(rs), busy / 1e6, (rs[0][0] - t0) / 1e6, (max(r[1] for r in rs) - t0) / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for s, e, _, n in rs:
        agg[n][0] += 1
        agg[n][1] += e - s
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print('    %-54s n=%5d  %6.2f ms  avg %6.1f us' % (n, c, d / 1e6, d / c / 1e3))
# concurrency profile: time with k queues busy
ev = []
for s, e, q, n in step:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
lvl, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[lvl] += t - last
    last = t
    lvl += d
print('\ntime with k kernels in flight:', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
