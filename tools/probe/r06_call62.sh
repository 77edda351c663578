mkdir -p gpurun_out/r06
for rep in 1 2; do
for v in "SG_PAR_SPLIT=0 SG_TAIL_CAPTURE=0" "SG_PAR_SPLIT=0 SG_TAIL_CAPTURE=1" "SG_PAR_SPLIT=1 SG_TAIL_CAPTURE=1"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_par_split.txt
done
done
for i in 1 2 3 4 5 6; do
SG_PAR_SPLIT=0 SG_TAIL_CAPTURE=0 timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "8-2-1" > gpurun_out/r06/dist_old_$i.log 2>&1
echo "old-config run $i rc=$?"
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "8-2-1" > gpurun_out/r06/dist_new_$i.log 2>&1
echo "new-config run $i rc=$?"
done
