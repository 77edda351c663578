mkdir -p gpurun_out/r06
for v in "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=5" "GPU_MAX_HW_QUEUES=5 SG_STREAM_GROUPS=front,mstep" "GPU_MAX_HW_QUEUES=8 SG_STREAM_GROUPS=front,mstep" "GPU_MAX_HW_QUEUES=8 SG_STREAM_GROUPS="; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_hw_queues2.txt
done
