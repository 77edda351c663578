mkdir -p gpurun_out/r06i
( time timeout 1300 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06i/gpu_tests.log 2>&1; tail -5 gpurun_out/r06i/gpu_tests.log
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), 'w43', d['kernels']['top'].get('wino43_bgemm_t64'), 'w24', d['kernels']['top'].get('wino24_bgemm_t128'), {k:round(v['images_per_s'],1) for k,v in d['legs'].items()}, 'frac', round(d['roofline']['frac'],3), round(d['kernels']['all_mfma_gemms']['frac'],3), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
