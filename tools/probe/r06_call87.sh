mkdir -p gpurun_out/r06
SG_STREAM_GROUPS=front,mstep,imgD,objD,adam,stage timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetcher or side_stream or training_loop" 2>&1 | tail -2
for rep in 1 2; do
for v in "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam" "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam,stage"; do
env $v python bench.py --steps 20 --warmup 5 --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'host_buffers', round(d['secondary']['host_buffers']['images_per_s'],1), round(d['secondary']['host_buffers']['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_stage_stream.txt
done
done
