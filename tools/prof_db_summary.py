"""Summarise a rocprofv3 --kernel-trace run stored as a rocpd sqlite database (ROCm 7.2 default output): per kernel
launches / time per training step, launches per step, GPU busy fraction.  Steps are delimited by the generator's RGB-head
forward kernel (smallm_fwd_kernel), which runs exactly once per G+D step.
usage: python tools/prof_db_summary.py <results.db> [skip_steps] [top] [--by-grid]
--by-grid: a second table per (kernel, grid size in workgroups): which launches fill the chip and which leave a partial round."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return 'igemm<' + re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:104] + '>'
    return re.sub(r'\(.*', '', n).replace('void ', '')[:80]


def main():
    argv = [a for a in sys.argv if a != '--by-grid']
    by_grid = '--by-grid' in sys.argv
    path = argv[1]
    skip = int(argv[2]) if len(argv) > 2 else 2
    top = int(argv[3]) if len(argv) > 3 else 40
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if 'smallm_fwd_kernel' in r[0]]
    assert len(marks) > skip + 1, 'not enough steps in the trace'
    sel = rows[marks[skip]:marks[-1]]
    steps = len(marks) - 1 - skip
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in sel:
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(a[1] for a in agg.values())
    span = (sel[-1][2] - sel[0][1]) / 1e3
    print('steps analysed: %d   launches/step: %.1f   kernel time: %.2f ms/step   wall span: %.2f ms/step   GPU busy: %.1f %%'
          % (steps, len(sel) / steps, tot / steps / 1e3, span / steps / 1e3, 100 * tot / span))
    own = sum(a[0] for k, a in agg.items() if not ('at::' in k or 'rocblas' in k.lower() or 'Cijk' in k or 'elementwise' in k))
    print('library kernels/step: %.1f   ATen / rocBLAS kernels/step: %.1f' % (own / steps, (len(sel) - own) / steps))
    print('| kernel | launches/step | ms/step | avg us |')
    print('|---|---|---|---|')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('| %s | %.1f | %.3f | %.1f |' % (k, a[0] / steps, a[1] / steps / 1e3, a[1] / a[0]))
    if by_grid:
        by_grid_table(db, marks, skip, steps, top)


def by_grid_table(db, marks, skip, steps, top):
    if True:
        cols = [r[1] for r in db.execute('pragma table_info(kernels)').fetchall()]
        gx = [c for c in ('grid_x', 'grid_size_x', 'grid_size') if c in cols]
        wx = [c for c in ('workgroup_x', 'workgroup_size_x', 'workgroup_size') if c in cols]
        if not gx or not wx:
            print('\n(no grid columns in this database: %s)' % ', '.join(cols))
            return
        mul = lambda base: '*'.join(c for c in (base, base.replace('x', 'y'), base.replace('x', 'z')) if c in cols and c.endswith(('x', 'y', 'z'))) or base
        q = 'select name, start, end, (%s), (%s) from kernels order by start' % (mul(gx[0]), mul(wx[0]))
        rows2 = db.execute(q).fetchall()[marks[skip]:marks[-1]]
        agg2 = collections.defaultdict(lambda: [0, 0.0])
        for n, s, e, g, w in rows2:
            a = agg2[(short(n), int(g) // max(int(w), 1))]
            a[0] += 1
            a[1] += (e - s) / 1e3
        print('\n| kernel | workgroups | launches/step | ms/step | avg us |')
        print('|---|---|---|---|---|')
        for (k, g), a in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:top]:
            print('| %s | %d | %.1f | %.3f | %.1f |' % (k, g, a[0] / steps, a[1] / steps / 1e3, a[1] / a[0]))


if __name__ == '__main__':
    main()
