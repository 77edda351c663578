#!/bin/bash
# round-4 GPU session H: the driver's default bench invocation incl. the live PMC traffic leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T0=$(date +%s)
timeout 1200 python bench.py > $O/r4h_bench_default.json 2> $O/r4h_bench_default.err
echo "bench.py default run: $(( $(date +%s) - T0 )) s wall"
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r4h_bench_default.json') if l.startswith('{')][-1])
print(round(d['value'],1), round(d['ms_per_step'],3)); print(json.dumps(d['roofline'],indent=1)[:2500]); print(d.get('secondary')); print(d.get('legs')); print(d.get('cpu_baseline'))
P
tail -5 $O/r4h_bench_default.err
