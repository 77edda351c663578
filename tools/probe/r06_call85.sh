mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_opts.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r06/suite_opts.log
for rep in 1 2; do
for v in "SG_SPLIT_KMIN=1024 SG_W43_TAIL_SPLIT=1" "SG_X=0"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_options_with_streams.txt
done
for leg in c5 c4; do
echo "old $leg $(SG_SPLIT_KMIN=1024 SG_W43_TAIL_SPLIT=1 python tools/run_leg.py $leg 8 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_options_with_streams.txt
echo "new $leg $(python tools/run_leg.py $leg 8 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_options_with_streams.txt
done
done
