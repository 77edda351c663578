mkdir -p gpurun_out/r06
( SG_W43_TAIL_SPLIT=2 timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/tests_tail2.log 2>&1; tail -4 gpurun_out/r06/tests_tail2.log
timeout 600 python tools/bench_gemm_classes.py --sweep w43_tail_split=1,2 --iters 20 2>&1 | grep -v amdgpu > gpurun_out/r06/gemm_tail_general.md
python - <<'PY'
import re, collections
rows = collections.defaultdict(dict)
for line in open('gpurun_out/r06/gemm_tail_general.md'):
    c = [x.strip() for x in line.split('|')]
    if len(c) < 10 or not c[1].startswith('w43_tail_split='): continue
    if 'igemm' not in c[4] and 'bgemm' not in c[4] and 'linear' not in c[4]: continue
    rows[(c[2], c[3], c[4])][c[1]] = float(c[6]) * float(c[5])
tot = collections.Counter()
for k, v in sorted(rows.items()):
    a, b = v.get('w43_tail_split=1'), v.get('w43_tail_split=2')
    if a and b:
        print('%-8s %-6s %-22s %8.1f -> %8.1f us  %+5.1f %%' % (k[0], k[1], k[2], a, b, 100 * (b - a) / a))
        tot['a'] += a; tot['b'] += b
print('total %.1f -> %.1f us (%+.1f %%)' % (tot['a'], tot['b'], 100 * (tot['b'] - tot['a']) / tot['a']))
PY
