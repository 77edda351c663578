#!/bin/bash
# round-4 GPU session E: k-tile depth variants of the gather / weight-gradient GEMMs, tile thresholds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off"
V=$PWD/scene_generation_amd/csrc/variants
( timeout 600 $B ) > $O/r4e_base.json 2> $O/r4e_base.err
( SG_LIB_PATH=$V/nsub2.so timeout 600 $B ) > $O/r4e_nsub2.json 2> $O/r4e_nsub2.err
( SG_LIB_PATH=$V/nsw2.so timeout 600 $B ) > $O/r4e_nsw2.json 2> $O/r4e_nsw2.err
( SG_T128_MIN=200 timeout 600 $B ) > $O/r4e_t128min200.json 2> $O/r4e_t128min200.err
( SG_TILE3_MIN=400 timeout 600 $B ) > $O/r4e_tile3min400.json 2> $O/r4e_tile3min400.err
( timeout 600 $B ) > $O/r4e_base2.json 2> $O/r4e_base2.err
python - <<'P'
import json
for n in ('base','nsub2','nsw2','t128min200','tile3min400','base2'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4e_%s.json'%n) if l.startswith('{')][-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],3), 'all_gemms', round(d['kernels']['all_mfma_gemms']['frac'],4), round(d['kernels']['all_mfma_gemms']['ms_per_step'],2))
        print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernels']['top'].items() if v['tflops']})
    except Exception as e: print(n,'failed',e)
P
