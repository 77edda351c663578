#!/bin/bash
# A/B of env-switched kernel variants on the GPU box: a parity subset, the conv micro-benchmark and one short bench line per
# variant.  usage: tools/gpu_ab.sh OUT_PREFIX "VAR=val ..." "VAR=val ..." ...   ("" = defaults)
out=gpurun_out/$1; shift
python -m pytest tests -m gpu -x -q -k "${SG_AB_TESTS:-conv2d or linear or gconv or vgg_full or two_rank or graphed or fast_paths}" 2>&1 | tail -6 > ${out}_tests.log
echo "pytest rc=${PIPESTATUS[0]}" >> ${out}_tests.log
i=0
for v in "$@"; do
  echo "== variant $i: '$v'" >> ${out}_ab.log
  if [ -n "$SG_AB_MICRO" ]; then env $v python tools/bench_conv.py $SG_AB_MICRO 2>/dev/null | grep -v amdgpu.ids >> ${out}_ab.log; fi
  env $v python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); o = {k: d[k] for k in ('value', 'ms_per_step', 'host_issue_isolated_ms_per_step') if k in d}; o.update({k: v['ms_per_step'] for k, v in d.get('kernels', {}).get('top', {}).items()}); o['roofline'] = d.get('roofline', {}).get('frac'); print(json.dumps(o))" >> ${out}_ab.log 2>&1
  i=$((i+1))
done
