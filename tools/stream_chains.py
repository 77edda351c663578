"""Where the small launches of a step sit: from a rocprofv3 --kernel-trace database (rocpd sqlite), per HIP stream / HSA queue the
launch count and busy time of one training step, and on the busiest stream the RUNS of consecutive short kernels (< LIMIT us each)
-- chains of latency-bound launches that leave the chip mostly idle -- with their duration, launch count and the kernels they
start / end with.  Steps are delimited by smallm_fwd_kernel (once per G+D step), as in prof_db_summary.py.
usage: python tools/stream_chains.py <results.db> [skip_steps=2] [limit_us=14] [min_run=6]"""
import collections
import sqlite3
import sys

from prof_db_summary import short


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    limit = float(sys.argv[3]) if len(sys.argv) > 3 else 14.0
    min_run = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)').fetchall()]
    qcol = next((c for c in ('stream_id', 'stream', 'queue_id', 'queue') if c in cols), None)
    print('columns:', ', '.join(cols))
    print('stream column:', qcol)
    rows = db.execute('select name, start, end, %s from kernels order by start' % (qcol or '0')).fetchall()
    marks = [i for i, r in enumerate(rows) if 'smallm_fwd_kernel' in r[0]]
    assert len(marks) > skip + 1, 'not enough steps in the trace'
    sel = rows[marks[skip]:marks[skip + 1]]                 # ONE step
    t0, t1 = sel[0][1], sel[-1][2]
    print('one step: %d launches, wall %.2f ms' % (len(sel), (t1 - t0) / 1e6))
    per = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, q in sel:
        per[q][0] += 1
        per[q][1] += (e - s) / 1e3
    print('| stream | launches | busy ms |')
    print('|---|---|---|')
    for q, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print('| %s | %d | %.3f |' % (q, c, t / 1e3))
    mainq = max(per.items(), key=lambda kv: kv[1][1])[0]
    m = [r for r in sel if r[3] == mainq]
    others = [r for r in sel if r[3] != mainq]
    # time on which ANY other stream is busy (to tell hidden gaps from exposed ones)
    runs, cur = [], []
    for r in m:
        if (r[2] - r[1]) / 1e3 < limit:
            cur.append(r)
        else:
            if len(cur) >= min_run:
                runs.append(cur)
            cur = []
    if len(cur) >= min_run:
        runs.append(cur)
    print('\nruns of >= %d consecutive launches shorter than %.0f us on stream %s:' % (min_run, limit, mainq))
    print('| start ms | span us | launches | kernel us | other streams busy us | first kernel | last kernel | most frequent |')
    print('|---|---|---|---|---|---|---|---|')
    tot_span = 0.0
    for run in sorted(runs, key=lambda r: -(r[-1][2] - r[0][1])):
        a, b = run[0][1], run[-1][2]
        kt = sum(e - s for _, s, e, _ in run) / 1e3
        ob = sum(max(0, min(e, b) - max(s, a)) for _, s, e, _ in others) / 1e3
        names = collections.Counter(short(n) for n, _, _, _ in run)
        tot_span += (b - a) / 1e3
        print('| %.2f | %.0f | %d | %.0f | %.0f | %s | %s | %s x%d |' % ((a - t0) / 1e6, (b - a) / 1e3, len(run), kt, ob, short(run[0][0])[:40],
                                                                  short(run[-1][0])[:40], names.most_common(1)[0][0][:40], names.most_common(1)[0][1]))
    print('\ntotal span of these runs: %.2f ms of the %.2f ms step' % (tot_span / 1e3, (t1 - t0) / 1e6))
    # gaps on the main stream
    gaps = [(m[i + 1][1] - m[i][2]) / 1e3 for i in range(len(m) - 1)]
    big = sorted(((g, i) for i, g in enumerate(gaps) if g > 20), reverse=True)[:12]
    print('\nmain-stream gaps: %d boundaries, sum %.2f ms, median %.2f us; the largest:' % (len(gaps), sum(gaps) / 1e3, sorted(gaps)[len(gaps) // 2]))
    for g, i in big:
        print('  %.0f us after %s (before %s)' % (g, short(m[i][0])[:50], short(m[i + 1][0])[:50]))


if __name__ == '__main__':
    sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
    main()
