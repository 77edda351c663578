#!/bin/bash
# round-4 GPU session B: skinny-GEMM validation, micro-benchmark, full kernel table, headline A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "linear or gconv or step_vs_oracle or embedding" 2>&1 | tail -15 ) > $O/r4b_tests.log 2>&1
( timeout 300 python tools/bench_linear.py ) > $O/r4b_bench_linear.txt 2>&1
( timeout 600 python bench.py --steps 10 --warmup 4 --no_legs --no_secondary --cpu_baseline off ) > $O/r4b_bench_skinny_on.json 2> $O/r4b_bench_skinny_on.err
( SG_LINEAR_SKINNY=0 timeout 600 python bench.py --steps 10 --warmup 4 --no_legs --no_secondary --cpu_baseline off ) > $O/r4b_bench_skinny_off.json 2> $O/r4b_bench_skinny_off.err
rm -rf /tmp/prof_r4b
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r4b -o r4b -- python $OLDPWD/bench.py --steps 6 --warmup 4 --no_legs --no_secondary --no_prof --cpu_baseline off ) > $O/r4b_rocprof.log 2>&1
DB=$(find /tmp/prof_r4b -name "*.db" | head -1)
python tools/prof_db_summary.py "$DB" 2 400 > $O/r4b_kernel_stats_full.md 2>&1
tail -5 $O/r4b_tests.log; cat $O/r4b_bench_linear.txt | head -40; python - <<'P'
import json
for n in ('on','off'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4b_bench_skinny_%s.json'%n) if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d.get('launches_per_step'), d['kernels']['top'].get('linear'))
    except Exception as e: print(n,'failed',e)
P
head -5 $O/r4b_kernel_stats_full.md
