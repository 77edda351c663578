#!/bin/bash
# round-4 GPU session K: Winograd weight gradient on the forward's V / the data gradient's Ytp (A/B) + ops package smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off --pmc off"
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "conv2d or test_full_step_vs_oracle or graphed_segments or instance_norm or batch_norm or pooling or gconv_layer or masks_to_layout_golden or crop_golden or test_losses or fused_adam or sinks or embedding" 2>&1 | tail -5 ) > $O/r4k_tests.log 2>&1
( SG_WINO_REUSE=0 timeout 300 python tools/bench_conv.py res3x3 ) > $O/r4k_conv_reuse0.txt 2>&1
( SG_WINO_REUSE=1 timeout 300 python tools/bench_conv.py res3x3 ) > $O/r4k_conv_reuse1.txt 2>&1
( SG_WINO_REUSE=0 timeout 600 $B ) > $O/r4k_reuse0.json 2> $O/r4k_reuse0.err
( SG_WINO_REUSE=1 timeout 600 $B ) > $O/r4k_reuse1.json 2> $O/r4k_reuse1.err
tail -4 $O/r4k_tests.log; grep -h res3x3 $O/r4k_conv_reuse0.txt $O/r4k_conv_reuse1.txt
python - <<'P'
import json
for n in ('reuse0','reuse1'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4k_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],3), d['kernels']['top'].get('wino_bgemm_t128'), d['kernels']['top'].get('wino_transforms'))
    except Exception as e: print(n,'failed',e); print(open('gpurun_out/r4k_%s.err'%n).read()[-1500:])
P
