mkdir -p gpurun_out/r06
for v in "SG_X=0" "SG_TAIL_SMAX=2" "SG_TAIL_SMAX=8" "SG_TAIL_KTMIN=4" "SG_TAIL_KTMIN=16" "SG_SPLIT_TARGET=1024" "SG_SPLIT_TARGET=2048" "SG_T128_MIN=256" "SG_T128_MIN=512" "SG_TILE3_MIN=512" "SG_TILE3_MIN=1024" "SG_PAR_XCD_CHUNK=8" "SG_PAR_XCD_CHUNK=32" "SG_W24_GEMM_TILE=1" "SG_LINEAR_NSUB=1" "SG_PAR_SPLIT=2" "SG_X=1"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_options2_with_streams.txt
done
