mkdir -p gpurun_out/r06
for rep in 1 2 3; do
for v in "SG_STREAM_GROUPS=front,mstep,imgD,objD" "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_adam2_stream.txt
done
done
for v in "SG_STREAM_GROUPS=front,mstep,imgD,objD" "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam" "SG_STREAM_GROUPS=front,mstep,imgD,objD" "SG_STREAM_GROUPS=front,mstep,imgD,objD,adam"; do
for leg in c5 c4; do
echo "$v $leg $(env $v python tools/run_leg.py $leg 8 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_adam2_stream.txt
done
done
