"""Per-launch efficiency of the MFMA GEMM launches of a leg: joins a rocprofv3 --kernel-trace database with the library's launch log
(SG_LAUNCH_LOG: grid, grid.z, M, N, K per igemm launch, in launch order; run with SG_GRAPHS=0 so that every launch is logged when it
is issued).  Prints the launches of the LAST step, heaviest first, with 2*M*N*K / time (parity-class and batched launches log the
first class / one batch only: their rate is a lower bound, marked '~').
usage: python tools/launch_eff.py <results.db> <launch.log> <launches of igemm per step, 0 = guess> [top]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    return 'igemm<' + re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:96] + '>'


db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute('select name, start, end, grid_x, grid_z from kernels order by start').fetchall() if 'igemm_kernel' in r[0]]
log = [l.split() for l in open(sys.argv[2]) if l.strip()]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 70
print('igemm dispatches %d, log lines %d' % (len(rows), len(log)))
n = min(len(rows), len(log))
rows, log = rows[len(rows) - n:], log[len(log) - n:]
per = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else 0
if per == 0:
    # period of the log's (M, N, K) sequence
    key = [tuple(l[:5]) for l in log]
    for cand in range(50, n // 2):
        if key[n - cand:] == key[n - 2 * cand:n - cand]:
            per = cand
            break
print('launches per step:', per)
agg = collections.OrderedDict()
steps = 3
for s in range(steps):
    lo = n - (s + 1) * per
    for i in range(per):
        (name, st, en, gx, gz), l = rows[lo + i], log[lo + i]
        k = (i, short(name), int(l[0]), int(l[1]), int(l[2]), int(l[3]), int(l[4]))
        if int(l[0]) != gx:
            k = k + ('GRID MISMATCH %d' % gx,)
        agg.setdefault(k, []).append((en - st) / 1e3)
tot = sum(sum(v) / len(v) for v in agg.values())
print('igemm time per step: %.2f ms' % (tot / 1e3))
print('| # | kernel | workgroups x z | M | N | K | us | TFLOP/s (logged M,N,K) | share |')
print('|---|---|---|---|---|---|---|---|---|')
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:top]:
    us = sum(v) / len(v)
    i, name, thr, gz, M, N, K = k[:7]
    tf = 2.0 * M * N * K / us * 1e-6
    print('| %d | %s | %d x %d | %d | %d | %d | %.1f | %.1f | %.1f %% | %s' % (i, name, thr // 256, gz, M, N, K, us, tf, 100 * us / tot, ' '.join(k[7:])))
