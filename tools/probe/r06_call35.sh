mkdir -p gpurun_out/r06
python tools/probe/fixedtap_ab.py 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r06/fixedtap_ab.txt
python tools/probe/fixedtap_ab_step.py 2>&1 | grep -v amdgpu.ids | tail -8 >> gpurun_out/r06/fixedtap_ab.txt
for rep in 1 2 3; do
for v in 2 1; do
SG_FIXEDTAP=$v python bench.py --steps 20 --warmup 5 --no_secondary --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t=d['kernels']['top']; print('fixedtap=$v', round(d['value'],1), round(d['ms_per_step'],3), {k:t[k]['ms_per_step'] for k in t if 'kn1' in k or 'kn0' in k}, round(d['kernels']['all_mfma_gemms']['frac'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" >> gpurun_out/r06/fixedtap_ab.txt
done
done
cat gpurun_out/r06/fixedtap_ab.txt
