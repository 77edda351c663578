"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs: they do not fit one pass)
stored as rocpd sqlite databases.  Prints a table and, with --json, writes {profiler kind: bytes per launch} for bench.py's
roofline.traffic.  Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads (x2); WRITE_SIZE is calibrated on adam_kernel, whose traffic is
known exactly (16 B read + 12 B written per parameter).
With --launch-log <file> (the SG_LAUNCH_LOG of the FETCH pass: one line per igemm launch, in launch order, carrying the
ALGORITHMIC bytes of the conv / GEMM it belongs to -- operands read once + result written once) the table gains an
"algorithmic MB/launch" column and the ratio (fetched + written) / algorithmic: > 1 = operands re-fetched from beyond L2.
usage: python tools/pmc_db_summary.py <fetch.db> <write.db> [--json out.json] [--launch-log log.txt]"""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return 'igemm<' + re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:104] + '>'
    return re.sub(r'\(.*', '', n).replace('void ', '')[:80]


def load(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute('select kernel_name, grid_size, value, duration from counters_collection where counter_name = ?',
                      (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, grid, val, dur in rows:
        a = agg[(short(name), grid)]
        a[0] += 1
        a[1] += val * 1024.0
        a[2] += dur / 1e3
    return agg


def load_alg(path, logpath):
    """{(kernel, grid): mean algorithmic bytes per launch}: igemm dispatches of the FETCH pass in dispatch order, matched
    one-to-one with the lines of the launch log (same process, single stream, graphs off)"""
    db = sqlite3.connect(path)
    rows = db.execute('select dispatch_id, kernel_name, grid_size from counters_collection where counter_name = ? '
                      'order by dispatch_id', ('FETCH_SIZE',)).fetchall()
    seen, disp = set(), []
    for did, name, grid in rows:
        if did in seen or 'igemm_kernel' not in name:
            continue
        seen.add(did)
        disp.append((short(name), grid))
    log = [l.split() for l in open(logpath) if l.strip()]
    out = collections.defaultdict(lambda: [0, 0.0])
    if len(log) != len(disp):
        print('(launch log has %d lines, the database %d igemm dispatches: algorithmic column omitted)' % (len(log), len(disp)))
        return {}
    bad = 0
    for (k, grid), l in zip(disp, log):
        if int(l[0]) * max(int(l[1]), 1) != grid:
            bad += 1
            continue
        a = out[(k, grid)]
        a[0] += 1
        a[1] += float(l[5])
    if bad:
        print('(%d of %d launches did not line up with the log and were skipped)' % (bad, len(log)))
    return {k: v[1] / v[0] for k, v in out.items() if v[0]}


def main():
    f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    # calibration: the largest adam launch (the generator's flat buffer)
    adam = [(k, v) for k, v in f.items() if k[0] == 'adam_kernel']
    k_adam = max(adam, key=lambda kv: kv[0][1])[0]
    n_params = k_adam[1]                              # one thread per parameter
    rd_adam, wr_adam = 2 * f[k_adam][1] / f[k_adam][0], w[k_adam][1] / w[k_adam][0]
    print('calibration on adam_kernel (%d parameters): read %.1f MB vs %.1f expected (x2 applied), written %.1f MB vs %.1f expected'
          % (n_params, rd_adam / 1e6, 16.0 * n_params / 1e6, wr_adam / 1e6, 12.0 * n_params / 1e6))
    wcal = 12.0 * n_params / wr_adam
    alg = load_alg(sys.argv[1], sys.argv[sys.argv.index('--launch-log') + 1]) if '--launch-log' in sys.argv else {}
    print('| kernel | grid | launches | fetch MB/launch (x2) | write MB/launch (x%.3f) | avg us | GB/s | algorithmic MB/launch | traffic / algorithmic |' % wcal)
    print('|---|---|---|---|---|---|---|---|---|')
    out = {}
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k, a in sorted(f.items(), key=lambda kv: -kv[1][1]):
        b = w.get(k, [1, 0.0, 0.0])
        fe, wr, us = 2 * a[1] / a[0], wcal * b[1] / max(b[0], 1), a[2] / a[0]
        t = tot[k[0]]
        t[0] += a[0]; t[1] += 2 * a[1]; t[2] += wcal * b[1] * a[0] / max(b[0], 1); t[3] += a[2]
        if a[1] * 2 > 0.005 * sum(x[1] * 2 for x in f.values()):
            ab = alg.get(k)
            print('| %s | %d | %d | %.1f | %.1f | %.1f | %.0f | %s | %s |' % (
                k[0], k[1], a[0], fe / 1e6, wr / 1e6, us, (fe + wr) / us / 1e3, '%.1f' % (ab / 1e6) if ab else '',
                '%.2f' % ((fe + wr) / ab) if ab else ''))
    if '--json' in sys.argv:
        names = {'igemm<T128x128, LoadKContig<128, true, false>, LoadKContig<128, true, false>, EpRowMajorPlain>': 'wino_bgemm_t128',
                 # F(4x4,3x3): the forward instantiation (the data gradient's -- LoadKContig / LoadXContigS -- has the same grid and bytes)
                 'igemm<T64x64, LoadKContig<64, true, false>, LoadKContig<64, true, false>, EpRowMajorPlain>': 'wino43_bgemm_t64'}
        for kname, kind in names.items():
            # the instantiation also runs the F(2x2,4x4) GEMMs (other grids): the bench line's kernel is the most frequent grid
            grids = [(a[0], k) for k, a in f.items() if k[0] == kname]
            if grids:
                kk = max(grids)[1]
                a, b = f[kk], w.get(kk, [1, 0.0, 0.0])
                t = [a[0], 2 * a[1], wcal * b[1] * a[0] / max(b[0], 1), a[2]]
                out[kind] = {'bytes_per_launch': (t[1] + t[2]) / t[0], 'fetch_bytes_per_launch': t[1] / t[0],
                             'write_bytes_per_launch': t[2] / t[0], 'launches_sampled': t[0],
                             'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 (gfx950), WRITE '
                                       'calibrated on adam_kernel: tools/pmc_db_summary.py, profiles/r05_pmc_traffic.md'}
        json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
