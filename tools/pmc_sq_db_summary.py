"""Per-kernel SQ counters from one rocprofv3 --pmc pass stored as a rocpd sqlite database (ROCm 7.2).
Counters expected: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU.  Percentages are of SQ_WAVE_CYCLES; "MFMA busy" = SQ_VALU_MFMA_BUSY_CYCLES /
(duration x clock x 1024 SIMDs) with the clock given on the command line (MHz under load).
usage: python tools/pmc_sq_db_summary.py <results.db> [clock_mhz] [top]"""
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
    path = sys.argv[1]
    mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2400.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    db = sqlite3.connect(path)
    rows = db.execute('select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection').fetchall()
    per = {}
    for did, name, grid, cname, val, dur in rows:
        d = per.setdefault(did, {'k': (short(name), grid), 'dur': dur})
        d[cname] = d.get(cname, 0.0) + val
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in per.values():
        a = agg[d['k']]
        a['n'] += 1
        a['dur'] += d['dur']
        for k, v in d.items():
            if k.startswith('SQ_'):
                a[k] += v
    print('| kernel | grid | launches | avg us | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | WAIT_INST_LDS | ACTIVE_VALU | MFMA busy (at %d MHz) |' % mhz)
    print('|---|---|---|---|---|---|---|---|---|---|')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['dur'])[:top]:
        wc = a['SQ_WAVE_CYCLES'] or 1.0
        busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['dur'] * 1e-9 * mhz * 1e6 * 1024) if a['dur'] else 0.0
        print('| %s | %d | %d | %.1f | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% |' % (
            k[0], k[1], a['n'], a['dur'] / a['n'] / 1e3, 100 * a['SQ_WAIT_ANY'] / wc, 100 * a['SQ_WAIT_INST_ANY'] / wc,
            100 * a['SQ_ACTIVE_INST_ANY'] / wc, 100 * a['SQ_WAIT_INST_LDS'] / wc, 100 * a['SQ_ACTIVE_INST_VALU'] / wc, 100 * busy))


if __name__ == '__main__':
    main()
