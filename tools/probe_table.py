"""Side-by-side table of tools/conv1x1_probe.py runs concatenated with 'G=<label>' separator lines."""
import re
import sys
blocks, cur = {}, None
for l in open(sys.argv[1]):
    if l.startswith('G='):
        cur = l.strip(); blocks[cur] = []; continue
    m = re.match(r'(.{36})\s+([\d.]+) us', l)
    if m and cur:
        blocks[cur].append((m.group(1).strip(), float(m.group(2))))
keys = list(blocks)
print('%-36s' % 'case', '  '.join('%7s' % k for k in keys))
tot = [0.0] * len(keys)
for i, (n, _) in enumerate(blocks[keys[0]]):
    print('%-36s' % n, '  '.join('%7.1f' % blocks[g][i][1] for g in keys))
    for j, g in enumerate(keys):
        tot[j] += blocks[g][i][1]
print('%-36s' % 'sum', '  '.join('%7.1f' % t for t in tot))
