mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "broadcast_second_source or modules_vs_reference or full_step_vs_oracle or fast_paths or graphed or bit_reproducible" 2>&1 | tail -3
rm -f gpurun_out/r06/condfold_ab2.txt
for rep in 1 2; do
for v in 0 2 1; do
SG_COND_FOLD=$v python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cond_fold=$v', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['images_per_s'],1) for k,v in d['legs'].items()}, round(d['kernels']['all_mfma_gemms']['frac'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" >> gpurun_out/r06/condfold_ab2.txt
done
done
cat gpurun_out/r06/condfold_ab2.txt
