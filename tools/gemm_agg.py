"""Aggregate a bench.py --dump-launches CSV by GEMM shape."""
import collections
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['family'] == '0']
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (r['M'], r['N'], r['K'], r['batch'], r['a_mode'], r['b_mode'])
    agg[k][0] += 1; agg[k][1] += float(r['ms']); agg[k][2] += float(r['flops'])
tot = sum(v[1] for v in agg.values())
print('total gemm ms/step %.2f  launched TF/s %.1f' % (tot / steps, sum(v[2] for v in agg.values()) / tot / 1e9))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print('%-40s %5d %9.3f ms/step %8.1f TF/s' % (','.join(k), v[0] // steps, v[1] / steps, v[2] / v[1] / 1e9 if v[1] > 0 else 0))
