mkdir -p gpurun_out/r06
for rep in 1 2; do
for v in "SG_STREAM_GROUPS=" "SG_STREAM_GROUPS=front" "SG_STREAM_GROUPS=front,mstep" "SG_STREAM_GROUPS=front,mstep,imgD"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'host_issue', round(d['host_issue_ms_per_step'],2), 'isolated', round(d['host_issue_isolated_ms_per_step'],2), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/host_cost_streams.txt
done
done
