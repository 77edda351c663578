"""Per-kernel totals of a rocprofv3 --kernel-trace rocpd database, plus the launch sequence of the last repetition.
usage: python tools/trace_list.py <results.db> [top]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return 'igemm<' + re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:110] + '>'
    return re.sub(r'\(.*', '', n).replace('void ', '')[:90]


db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute('select name, start, end from kernels order by start').fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[short(n)]
    a[0] += 1
    a[1] += (e - s) / 1e3
print('| kernel | launches | total ms | avg us |')
print('|---|---|---|---|')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('| %s | %d | %.3f | %.1f |' % (k, a[0], a[1] / 1e3, a[1] / a[0]))

# idle time between consecutive kernels (end of i -> start of i+1), overall and by the kernel that FOLLOWS the gap
if len(sys.argv) > 3 and sys.argv[3] == 'gaps':
    gaps = collections.defaultdict(lambda: [0, 0.0])
    hist = collections.Counter()
    tot = 0.0
    last_end = rows[0][2]
    for n, s, e in rows[1:]:
        g = max(0.0, (s - last_end) / 1e3)
        last_end = max(last_end, e)
        if g > 2000.0:                       # host-side pauses (warm-up, synchronisation points of the script)
            continue
        tot += g
        a = gaps[short(n)]
        a[0] += 1
        a[1] += g
        hist[min(int(g), 20)] += 1
    busy = sum((e - s) / 1e3 for _, s, e in rows)
    print('\nkernel time %.2f ms, idle between kernels %.2f ms (%.1f %%), %d launches' % (busy / 1e3, tot / 1e3, 100 * tot / (tot + busy), len(rows)))
    print('gap histogram (us: count):', ', '.join('%s%d: %d' % ('>=' if k == 20 else '', k, v) for k, v in sorted(hist.items())))
    print('| idle before kernel | launches | total idle ms | avg us |')
    print('|---|---|---|---|')
    for k, a in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
        print('| %s | %d | %.3f | %.2f |' % (k, a[0], a[1] / 1e3, a[1] / a[0]))

# overlap between queues: sum of kernel durations vs the length of their union (equal when nothing runs concurrently)
if len(sys.argv) > 3 and sys.argv[3] == 'overlap':
    cols = [r[1] for r in db.execute('pragma table_info(kernels)').fetchall()]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    print('\ncolumns:', cols)
    if qcol:
        per = collections.defaultdict(lambda: [0, 0.0])
        for q, s, e in db.execute('select %s, start, end from kernels' % qcol).fetchall():
            per[q][0] += 1
            per[q][1] += (e - s) / 1e6
        print('per %s:' % qcol, {k: (v[0], round(v[1], 2)) for k, v in per.items()})
    iv = sorted((s, e) for _, s, e in rows)
    union, cs, ce = 0.0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    tot = sum(e - s for s, e in iv)
    print('sum of kernel durations %.2f ms, union %.2f ms, overlapped %.2f ms (%.1f %%)' % (tot / 1e6, union / 1e6, (tot - union) / 1e6, 100 * (tot - union) / tot))
