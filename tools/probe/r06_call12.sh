mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "batch_norm or config5_step_vs or full_step_vs_oracle or golden" > gpurun_out/r06/tests_call12.log 2>&1; tail -2 gpurun_out/r06/tests_call12.log
for i in 1 2; do python tools/run_leg.py c5 10 2>/dev/null | tail -1; done > gpurun_out/r06/c5_leg_bn.txt; cat gpurun_out/r06/c5_leg_bn.txt
python tools/probe/host_buffers_probe.py > gpurun_out/r06/host_buffers_probe.txt 2>&1; tail -5 gpurun_out/r06/host_buffers_probe.txt
python bench.py --steps 20 --warmup 5 --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no_prof', round(d['value'],1), {k:round(v['images_per_s'],1) for k,v in d['secondary'].items() if 'images_per_s' in v})" > gpurun_out/r06/hostfed_noprof.txt 2>&1
python bench.py --steps 20 --warmup 5 --no_legs --cpu_baseline off --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with_prof', round(d['value'],1), {k:round(v['images_per_s'],1) for k,v in d['secondary'].items() if 'images_per_s' in v})" >> gpurun_out/r06/hostfed_noprof.txt 2>&1
cat gpurun_out/r06/hostfed_noprof.txt
