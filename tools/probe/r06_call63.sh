mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lazy_zero or parity_class_split or fused_adam or gradient_sinks" 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_lazy.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/suite_lazy.log
for rep in 1 2 3; do
for v in "SG_LAZY_ZERO=0" "SG_LAZY_ZERO=1"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" | tee -a gpurun_out/r06/ab_lazy_zero.txt
done
done
