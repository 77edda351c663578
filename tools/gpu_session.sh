#!/bin/bash
# One GPU-box session of round 5: a parity subset, optional micro-benchmarks, one bench line per variant.
# usage: tools/gpu_session.sh PREFIX "pytest -k expression" "extra commands (eval'ed, output appended to PREFIX_extra.log)" ["VAR=val ..." variants]
out=gpurun_out/$1; kexpr=$2; extra=$3; shift 3
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -15 > ${out}_tests.log
  echo "pytest rc=${PIPESTATUS[0]}" >> ${out}_tests.log
fi
if [ -n "$extra" ]; then eval "$extra" > ${out}_extra.log 2>&1; fi
i=0
for v in "$@"; do
  echo "== variant $i: '$v'" >> ${out}_ab.log
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off 2>${out}_bench_err_$i.log | tail -1 > ${out}_bench_$i.json
  python - ${out}_bench_$i.json >> ${out}_ab.log 2>&1 <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read())
o = {k: d[k] for k in ('value', 'ms_per_step', 'host_issue_isolated_ms_per_step', 'launches_per_step', 'host_calls_per_step') if k in d}
k = d.get('kernels', {})
o['gemms'] = k.get('all_mfma_gemms'); o['timed_ms'] = k.get('timed_kernels_ms_per_step'); o['roofline'] = d.get('roofline', {}).get('frac')
print(json.dumps(o))
print(json.dumps({n: (x['ms_per_step'], x.get('tflops') or x.get('gbs')) for n, x in k.get('top', {}).items()}))
print(json.dumps({n: (x['ms_per_step'], x['frac']) for n, x in d.get('roofline_hbm', {}).get('kernels', {}).items()}))
PY
  tail -c 400 ${out}_bench_err_$i.log > ${out}_bench_err_$i.tail; rm -f ${out}_bench_err_$i.log
  i=$((i+1))
done
