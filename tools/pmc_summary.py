"""Merge rocprofv3 --pmc passes of bench.py into a per-kernel table.
usage: python tools/pmc_summary.py <passA_counter_collection.csv> <passB_FETCH.csv> <passC_WRITE.csv> <steps> [top]
pass A counters: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'void igemm_kernel<(.*)>\(', n)
    if m:
        return re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', m.group(1))[:100]
    return re.sub(r'\(.*', '', n)[:80]


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        did = r['Dispatch_Id']
        if did not in seen[k]:
            seen[k].add(did)
            agg[k]['_ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            agg[k]['_n'] += 1
    return agg


def main():
    A, B, C = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    steps = int(sys.argv[4])
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 25
    tot_ns = sum(v['_ns'] for v in A.values())
    tot_mfma = sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in A.values())
    tot_gui = sum(v['GRBM_GUI_ACTIVE'] for v in A.values())
    print('all kernels: %.1f ms/step under PMC | MFMA busy = %.1f %% of the SIMD-cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 x 1024 SIMDs) | FETCH %.2f GB/step, '
          'WRITE %.2f GB/step (raw counter KB -> bytes, uncorrected)' % (
              tot_ns / 1e6 / steps, 100 * tot_mfma / (tot_gui * 128 + 1e-9),
              sum(v['FETCH_SIZE'] for v in B.values()) * 1024 / 1e9 / steps,
              sum(v['WRITE_SIZE'] for v in C.values()) * 1024 / 1e9 / steps))
    print('| kernel | launches/step | ms/step (pass A) | MFMA busy % of SIMD-cycles | FETCH MB/launch | WRITE MB/launch | (FETCH+WRITE)/time GB/s |')
    print('|---|---|---|---|---|---|---|')
    for k, v in sorted(A.items(), key=lambda kv: -kv[1]['_ns'])[:top]:
        n = v['_n']
        f = B[k]['FETCH_SIZE'] * 1024 / max(B[k]['_n'], 1)
        w = C[k]['WRITE_SIZE'] * 1024 / max(C[k]['_n'], 1)
        t = v['_ns'] / n * 1e-9
        print('| %s | %.1f | %.3f | %.1f | %.1f | %.1f | %.0f |' % (
            k, n / steps, v['_ns'] / 1e6 / steps, 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 128 + 1e-9),
            f / 1e6, w / 1e6, (f + w) / t / 1e9))


if __name__ == '__main__':
    main()
