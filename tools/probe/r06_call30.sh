mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "linear or gconv or full_step_vs_oracle or modules_vs_reference or graphed or reproducible" > gpurun_out/r06/tests_call30.log 2>&1; tail -4 gpurun_out/r06/tests_call30.log
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), 'launches', d['launches_per_step'], 'linear', d['kernels']['top'].get('linear'), {k:round(v['images_per_s'],1) for k,v in d['legs'].items()}, 'sclk', d['clocks']['sclk_mhz']['median'])"
done
