"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs per kernel name.
usage: traffic_agg.py <fetch.csv> <write.csv> [steps]   (FETCH_SIZE doubled on gfx950, KB -> bytes; see MI355X_MICROARCH.md)"""
import csv
import re
import sys
from collections import defaultdict


def load(path, scale):
    agg, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            n = re.sub(r'\(.*', '', n)[:70]
            agg[n] += float(r['Counter_Value']) * 1024.0 * scale
            cnt[n] += 1
    return agg, cnt


if __name__ == '__main__':
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    fe, cnt = load(sys.argv[1], 2.0)
    wr, _ = load(sys.argv[2], 1.0)
    names = sorted(set(fe) | set(wr), key=lambda n: -(fe.get(n, 0) + wr.get(n, 0)))
    tot = 0.0
    print('%-70s %8s %10s %10s' % ('kernel', 'calls', 'fetch GB', 'write GB'))
    for n in names:
        tot += fe.get(n, 0) + wr.get(n, 0)
        print('%-70s %8d %10.2f %10.2f' % (n, cnt.get(n, 0) / steps, fe.get(n, 0) / 1e9 / steps, wr.get(n, 0) / 1e9 / steps))
    print('total GB/step %.1f' % (tot / 1e9 / steps))
