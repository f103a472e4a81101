"""Average rocprofv3 --pmc counter values per (kernel, grid): pmc_agg.py <counter_collection.csv> [name-filter]"""
import csv
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:44]
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    k = (n, int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))
    a = agg[k][r['Counter_Name']]
    a[0] += 1
    a[1] += float(r['Counter_Value'])
for k, cs in sorted(agg.items()):
    print('%-44s wgs %7d n=%4d  ' % (k[0], k[1], max(v[0] for v in cs.values())) + '  '.join('%s=%.4g' % (c, v[1] / v[0]) for c, v in sorted(cs.items())))
