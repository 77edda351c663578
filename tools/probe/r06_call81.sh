mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "side_stream or graphed_segments or full_step_vs_oracle or headline or exports or globalgen" 2>&1 | tail -3
for rep in 1 2 3; do
for v in "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam" "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam,wprep"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_wprep_stream.txt
done
done
