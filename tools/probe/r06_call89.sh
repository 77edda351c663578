mkdir -p gpurun_out/r06
for rep in 1 2; do
for v in "SG_STREAM_PAD=0" "SG_STREAM_PAD=1" "SG_STREAM_PAD=2" "SG_STREAM_PAD=3" "SG_STREAM_PAD=4"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_hw_queues.txt
done
done
