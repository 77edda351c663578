mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_tailsplit.txt; : > $out
for rep in 1 2; do
for v in "SG_W43_TAIL_SPLIT=0 SG_W43_WGRAD_TILE=0" "SG_W43_TAIL_SPLIT=1 SG_W43_WGRAD_TILE=0" "SG_W43_TAIL_SPLIT=1 SG_W43_WGRAD_TILE=1"; do
  env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'w43', d['kernels']['top'].get('wino43_bgemm_t64'), 'frac', round(d['roofline']['frac'],3), 'gemms', round(d['kernels']['all_mfma_gemms']['frac'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" >> $out 2>&1
done; done
cat $out
