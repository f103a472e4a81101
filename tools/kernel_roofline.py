"""Per-kernel roofline table from three rocprofv3 --pmc passes of the same command (FETCH_SIZE, WRITE_SIZE, MfmaUtil):
usage: kernel_roofline.py fetch.csv write.csv mfma.csv steps > table.md
HBM bytes: FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes; duration from the dispatch timestamps of
the MfmaUtil pass; GB/s against 8 TB/s, MfmaUtil as reported (percent of MFMA-busy cycles)."""
import csv
import re
import sys
from collections import defaultdict


def key(r):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    return re.sub(r'\(.*', '', n)[:64]


def load(path, scale):
    agg = defaultdict(float)
    for r in csv.DictReader(open(path)):
        agg[key(r)] += float(r['Counter_Value']) * 1024.0 * scale
    return agg


fe, wr = load(sys.argv[1], 2.0), load(sys.argv[2], 1.0)
steps = float(sys.argv[4])
dur, cnt, mf = defaultdict(float), defaultdict(int), defaultdict(float)
for r in csv.DictReader(open(sys.argv[3])):
    k = key(r)
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    dur[k] += d
    cnt[k] += 1
    mf[k] += float(r['Counter_Value']) * d
names = sorted(dur, key=lambda n: -dur[n])
print('| kernel | launches/step | avg us | ms/step | HBM GB/step | GB/s | % of 8 TB/s | MfmaUtil % |')
print('|---|---|---|---|---|---|---|---|')
tot_t = tot_b = 0.0
for n in names[:36]:
    b = fe.get(n, 0) + wr.get(n, 0)
    tot_t += dur[n]
    tot_b += b
    print('| `%s` | %.0f | %.1f | %.2f | %.2f | %.0f | %.1f | %.1f |' % (n, cnt[n] / steps, 1e6 * dur[n] / cnt[n], 1e3 * dur[n] / steps, b / steps / 1e9,
                                                             b / dur[n] / 1e9, 100 * b / dur[n] / 8e12, mf[n] / dur[n]))
print()
print('listed kernels: %.1f ms/step, %.1f GB/step (%.0f GB/s average while a listed kernel runs)' % (1e3 * tot_t / steps, tot_b / steps / 1e9, tot_b / tot_t / 1e9))
