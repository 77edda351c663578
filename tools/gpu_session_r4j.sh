#!/bin/bash
# round-4 GPU session J: spill-buffer gradient accumulation
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off --pmc off"
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -q -p no:cacheprovider -x -k "sinks or bit_reproducible or two_rank or step_vs_oracle or side_streams or fused_adam or fast_paths or training_loop" 2>&1 | tail -6 ) > $O/r4j_tests.log 2>&1
( timeout 600 $B ) > $O/r4j_bench.json 2> $O/r4j_bench.err
tail -4 $O/r4j_tests.log
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r4j_bench.json') if l.startswith('{')][-1])
print(round(d['value'],1), round(d['ms_per_step'],3), d['launches_per_step'], d['host_calls_per_step'], round(d['host_issue_isolated_ms_per_step'],2))
P
tail -3 $O/r4j_bench.err
