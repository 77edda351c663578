"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) time per step, GPU busy fraction, gaps.
usage: python tools/prof_summary.py <kernel_trace.csv> <steps> [top]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:110]
    return re.sub(r'\(.*', '', n)[:90]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    agg = collections.defaultdict(lambda: [0, 0.0])
    busy, last_end, gaps = 0.0, None, []
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        agg[(short(r['Kernel_Name']), r['Grid_Size_X'] + 'x' + r['Grid_Size_Z'])][0] += 1
        agg[(short(r['Kernel_Name']), r['Grid_Size_X'] + 'x' + r['Grid_Size_Z'])][1] += (e - s) / 1e6
        if last_end is not None and s > last_end:
            gaps.append((s - last_end) / 1e3)
        busy += (e - max(s, last_end or s)) / 1e6 if (last_end is None or e > last_end) else 0.0
        last_end = max(last_end or e, e)
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6
    print('kernels %d | span %.1f ms | busy %.1f ms (%.1f%%) | sum of kernel time %.1f ms/step' % (
        len(rows), span, busy, 100 * busy / span, sum(v[1] for v in agg.values()) / steps))
    gaps.sort()
    if gaps:
        print('gaps: n=%d total %.1f ms | median %.1f us | p90 %.1f us | >100us: %d (%.1f ms)' % (
            len(gaps), sum(gaps) / 1e3, gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)],
            sum(g > 100 for g in gaps), sum(g for g in gaps if g > 100) / 1e3))
    print('| ms/step | launches/step | avg us | kernel | grid (threads x z) |')
    print('|---|---|---|---|---|')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('| %.3f | %.1f | %.1f | %s | %s |' % (v[1] / steps, v[0] / steps, 1000 * v[1] / v[0], k[0], k[1]))


if __name__ == '__main__':
    main()
