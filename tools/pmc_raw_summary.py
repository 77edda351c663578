"""Raw per-kernel counter sums from rocprofv3 --pmc passes (rocpd sqlite databases, ROCm 7.2), one row per (kernel, grid):
launches, average duration and every counter found, per launch.  usage: python tools/pmc_raw_summary.py <db> [<db> ...]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return 'igemm<' + re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:110] + '>'
    return re.sub(r'\(.*', '', n).replace('void ', '')[:80]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = []
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute('select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection').fetchall()
    per = {}
    for did, name, grid, cname, val, dur in rows:
        d = per.setdefault(did, {'k': (short(name), grid), 'dur': dur})
        d[cname] = d.get(cname, 0.0) + val
        if cname not in names:
            names.append(cname)
    for d in per.values():
        a = agg[d['k']]
        a['n@' + path] += 1
        a['dur@' + path] += d['dur']
        for k, v in d.items():
            if k not in ('k', 'dur'):
                a[k] += v
                a['cnt:' + k] += 1
print('| kernel | grid | avg us | ' + ' | '.join(names) + ' |')
print('|---|---|---|' + '---|' * len(names))
rowsout = []
for k, a in agg.items():
    durs = [(a['dur@' + p] / a['n@' + p]) for p in sys.argv[1:] if a.get('n@' + p)]
    if not durs or durs[0] < 20e3:
        continue
    rowsout.append((durs[0], '| %s | %d | %.1f | ' % (k[0], k[1], durs[0] / 1e3) +
                    ' | '.join('%.4g' % (a[c] / a['cnt:' + c]) if a.get('cnt:' + c) else '' for c in names) + ' |'))
for _, r in sorted(rowsout, reverse=True):
    print(r)
