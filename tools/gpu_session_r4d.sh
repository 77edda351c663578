#!/bin/bash
# round-4 GPU session D: A/B of interleaved stores in every GEMM (variant build) and XCD-pinned weight-gradient k-chunks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off"
( timeout 600 $B ) > $O/r4d_base.json 2> $O/r4d_base.err
( SG_WGRAD_XCD=1 timeout 600 $B ) > $O/r4d_xcd.json 2> $O/r4d_xcd.err
( SG_LIB_PATH=$PWD/scene_generation_amd/csrc/variants/pipe2all.so timeout 600 $B ) > $O/r4d_pipe2all.json 2> $O/r4d_pipe2all.err
( timeout 600 $B ) > $O/r4d_base2.json 2> $O/r4d_base2.err
( SG_WGRAD_XCD=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "conv2d or transpose" 2>&1 | tail -4 ) > $O/r4d_tests_xcd.log 2>&1
( SG_WGRAD_XCD=1 timeout 300 python tools/bench_conv.py down3x3s2_64 D0_4x4 D2_4x4 down3x3s2_512 first7x7 ) > $O/r4d_conv_xcd1.txt 2>&1
( SG_WGRAD_XCD=0 timeout 300 python tools/bench_conv.py down3x3s2_64 D0_4x4 D2_4x4 down3x3s2_512 first7x7 ) > $O/r4d_conv_xcd0.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|hbm|EA0_RD|EA0_WR" | head -40 > $O/r4d_counters_avail.txt
tail -3 $O/r4d_tests_xcd.log; grep -h -v amdgpu.ids $O/r4d_conv_xcd0.txt $O/r4d_conv_xcd1.txt; python - <<'P'
import json
for n in ('base','xcd','pipe2all','base2'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4d_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],3), 'all_gemms', round(d['kernels']['all_mfma_gemms']['frac'],4), round(d['kernels']['all_mfma_gemms']['ms_per_step'],2))
        print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernels']['top'].items() if v['tflops']})
    except Exception as e: print(n,'failed',e)
P
head -30 $O/r4d_counters_avail.txt
