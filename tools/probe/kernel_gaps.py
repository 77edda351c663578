"""From a rocprofv3 --kernel-trace CSV: is the GPU idle or busy when the step is slow?  Lists the stage_copy kernels, the idle gaps
longer than a threshold (with the kernels on either side) and busy / idle totals of the last ``tail_ms`` of the trace."""
import csv
import glob
import sys

d = sys.argv[1]
thr_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 700.0
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
t0 = rows[0][0]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - t0
print('last %.0f ms of the trace: %d kernels, busy %.1f ms, span %.1f ms, idle %.1f ms' % (tail_ms, len(rows), busy * 1e-6, span * 1e-6,
                                                                                          (span - busy) * 1e-6))
cp = [(s, e) for s, e, n in rows if 'stage_copy' in n]
print('stage_copy kernels: %d, durations us: %s' % (len(cp), ' '.join('%.0f' % ((e - s) * 1e-3) for s, e in cp)))
print('gaps > %.0f us:' % thr_us)
prev_end, prev_name = rows[0][1], rows[0][2]
small = 0.0
for s, e, n in rows[1:]:
    g = (s - prev_end) * 1e-3
    if g > thr_us:
        print('  +%8.1f ms  gap %8.1f us  after %-60.60s before %-60.60s' % ((prev_end - t0) * 1e-6, g, prev_name, n))
    elif g > 0:
        small += g
    if e > prev_end:
        prev_end, prev_name = e, n
print('sum of gaps <= %.0f us: %.1f ms' % (thr_us, small * 1e-3))
# where the idle time sits: (kernel before -> kernel after) pairs by total gap
import collections
import re
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:70]
pairs = collections.defaultdict(lambda: [0, 0.0, 0.0])
hist = collections.Counter()
prev_end, prev_name = rows[0][1], rows[0][2]
for s_, e, n in rows[1:]:
    g = (s_ - prev_end) * 1e-3
    if g > 0:
        p_ = pairs[(short(prev_name), short(n))]
        p_[0] += 1
        p_[1] += g
        p_[2] = max(p_[2], g)
        hist[min(int(g), 50) if g < 50 else (100 if g < 100 else 1000)] += 1
    if e > prev_end:
        prev_end, prev_name = e, n
print('gap histogram (us -> count): ' + ' '.join('%d:%d' % kv for kv in sorted(hist.items())))
print('| kernel before | kernel after | n | total gap us | mean us | max us |\n|---|---|---|---|---|---|')
for (a, b), (n, tot, mx) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:40]:
    print('| %s | %s | %d | %.0f | %.1f | %.0f |' % (a, b, n, tot, tot / n, mx))
