mkdir -p gpurun_out/r06
for rep in 1 2 3; do
for v in "SG_X=0" "SG_SPLIT_KMIN=2048" "SG_SPLIT_KMIN=4096" "SG_SPLIT_KMIN=2048 SG_W43_TAIL_SPLIT=2"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_options_with_streams.txt
done
done
