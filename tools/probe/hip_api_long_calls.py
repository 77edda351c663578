"""Summarise a rocprofv3 --hip-runtime-trace CSV: per API the call count / total / max, and every call longer than a threshold
(which HIP entry point is it that blocks the launching thread?  round 6, host path)."""
import csv
import glob
import sys
from collections import defaultdict

d, thr_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
files = glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)
agg = defaultdict(lambda: [0, 0.0, 0.0])
long_calls = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get('Function') or r.get('Name')
        dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        a = agg[name]
        a[0] += 1
        a[1] += dt
        a[2] = max(a[2], dt)
        if dt >= thr_ms:
            long_calls.append((int(r['Start_Timestamp']), name, dt, r.get('Thread_Id')))
print('| HIP API | calls | total ms | max ms |\n|---|---|---|---|')
for name, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('| %s | %d | %.1f | %.2f |' % (name, n, tot, mx))
print('\ncalls >= %.1f ms (start order): ' % thr_ms)
t0 = min([c[0] for c in long_calls] or [0])
for ts, name, dt, tid in sorted(long_calls)[:80]:
    print('  +%9.1f ms  %-28s %7.1f ms  thread %s' % ((ts - t0) * 1e-6, name, dt, tid))
