mkdir -p gpurun_out/r06g
( time timeout 1300 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06g/gpu_tests.log 2>&1
tail -5 gpurun_out/r06g/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06g/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'host', d.get('host'), 'sclk', d['clocks'].get('sclk_mhz'))
print({k:round(v['images_per_s'],1) for k,v in d['secondary'].items() if 'images_per_s' in v})
print({k:round(v['images_per_s'],1) for k,v in d['legs'].items()})
print(d['cpu_baseline'])
print('gemms', d['kernels']['all_mfma_gemms'], 'roofline', d['roofline']['frac'])
PY
