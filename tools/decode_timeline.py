"""Steady-state picture of the decode workload from a rocprofv3 kernel trace (tools/profile_decode.sh): per queue the busy time, the
time with k kernels in flight, and how long the decode-step kernels take alone / beside another search / beside encoder kernels.
usage: python tools/decode_timeline.py <kernel_trace.csv> [window_ms]   (the window is taken at the END of the trace: the timed batches)"""
import bisect
import collections
import csv
import sys


def short(n, w=40):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:w]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 300e6
    t1 = max(int(r['End_Timestamp']) for r in rows)
    t0 = t1 - win
    ev = collections.defaultdict(list)
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > t0:
            ev[r['Queue_Id'] + '/' + r['Stream_Id']].append((max(s, t0), e, r['Kernel_Name']))
    for k in ev:
        ev[k].sort()
    print('window %.1f ms' % (win / 1e6))
    kinds = {}
    for k, v in sorted(ev.items()):
        names = collections.Counter(short(n) for _, _, n in v)
        dec = sum(c for n, c in names.items() if 'decode_' in n or 'beam_step' in n)
        kinds[k] = 'search' if dec > 50 else ('encoder' if any('stem_fwd' in n for n in names) else 'other')
        print('queue/stream %-6s %-8s n=%6d busy %7.2f ms   top: %s' % (k, kinds[k], len(v), sum(e - s for s, e, _ in v) / 1e6,
                                                                     ', '.join('%s x%d' % nc for nc in names.most_common(3))))
    pts = []
    for k, v in ev.items():
        for s, e, _ in v:
            pts.append((s, 1, kinds[k]))
            pts.append((e, -1, kinds[k]))
    pts.sort()
    live = collections.Counter()
    hist = collections.Counter()
    last = t0
    for t, d, kind in pts:
        key = (min(live['search'], 3), min(live['encoder'], 1))
        hist[key] += t - last
        last = t
        live[kind] += d
    print('time with (search kernels, encoder kernels) in flight:')
    for key in sorted(hist):
        print('   searches %d%s encoder %d : %7.2f ms' % (key[0], '+' if key[0] == 3 else ' ', key[1], hist[key] / 1e6))
    enc = sorted(x for k, v in ev.items() if kinds[k] == 'encoder' for x in v)
    es = [x[0] for x in enc]
    sq = [k for k in ev if kinds[k] == 'search']
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for k in sq:
        others = sorted(x for k2 in sq if k2 != k for x in ev[k2])
        os_ = [x[0] for x in others]
        for s, e, n in ev[k]:
            def ov(lst, starts):
                j = max(0, bisect.bisect_left(starts, s) - 2)
                tot = 0
                while j < len(lst) and lst[j][0] < e:
                    tot += max(0, min(e, lst[j][1]) - max(s, lst[j][0]))
                    j += 1
                return tot
            oe, oo = ov(enc, es), ov(others, os_)
            cls = 'beside encoder' if oe > 0.5 * (e - s) else ('beside a search' if oo > 0.5 * (e - s) else ('alone' if oe == 0 and oo == 0 else 'partial'))
            a = agg[short(n, 34)][cls]
            a[0] += 1
            a[1] += e - s
    print('%-36s %s' % ('decode kernel', '   '.join('%-22s' % c for c in ('alone', 'beside a search', 'beside encoder'))))
    for n, d in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1].values()))[:12]:
        print('%-36s %s' % (n, '   '.join('n=%5d avg %6.1f us ' % (d[c][0], d[c][1] / max(1, d[c][0]) / 1e3) for c in ('alone', 'beside a search', 'beside encoder'))))
    for k in sq:
        v = ev[k]
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        big = [g for g in gaps if g > 30000]
        print('search queue %s: gaps > 30 us: n=%d, %.2f ms in total; all gaps %.2f ms' % (k, len(big), sum(big) / 1e6, sum(g for g in gaps if g > 0) / 1e6))


if __name__ == '__main__':
    main()
