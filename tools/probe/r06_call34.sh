mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv2d or conv_transpose or upconv or modules_vs_reference or full_step_vs_oracle or fast_paths" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t=d['kernels']['top']; print(round(d['value'],1), round(d['ms_per_step'],3), {k:t[k]['ms_per_step'] for k in t if 'kn1' in k or 'kn0' in k}, {k:round(v['images_per_s'],1) for k,v in d['legs'].items()}, round(d['kernels']['all_mfma_gemms']['frac'],3), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
