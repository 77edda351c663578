#!/bin/bash
# round-4 GPU session C: SGPR-offset A loader + interleaved-store pipeline A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "conv or linear or wino or factored or upconv or gconv" 2>&1 | tail -8 ) > $O/r4c_tests.log 2>&1
( SG_WINO_PIPE=1 timeout 300 python tools/bench_conv.py res3x3 D3_4x4 D2_4x4 down3x3s2_512 ) > $O/r4c_conv_pipe1.txt 2>&1
( SG_WINO_PIPE=2 timeout 300 python tools/bench_conv.py res3x3 D3_4x4 ) > $O/r4c_conv_pipe2.txt 2>&1
( SG_WINO_PIPE=1 timeout 600 python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off ) > $O/r4c_bench_pipe1.json 2> $O/r4c_bench_pipe1.err
( SG_WINO_PIPE=2 timeout 600 python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off ) > $O/r4c_bench_pipe2.json 2> $O/r4c_bench_pipe2.err
tail -4 $O/r4c_tests.log; cat $O/r4c_conv_pipe1.txt $O/r4c_conv_pipe2.txt | grep -v amdgpu.ids; python - <<'P'
import json
for n in ('pipe1','pipe2'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4c_bench_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['frac'],4), round(d['roofline']['avg_us'],1), 'all_gemms', round(d['kernels']['all_mfma_gemms']['frac'],4), round(d['kernels']['all_mfma_gemms']['ms_per_step'],2))
        print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernels']['top'].items() if v['tflops']})
    except Exception as e: print(n,'failed',e)
P
