timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv_transpose or test_conv2d or full_step_vs_oracle or fast_paths or bit_reproducible" 2>&1 | tail -2
python tools/bench_gemm_classes.py --only Gup1,Gup2,Gup3,Gdn4,Gdn3,Gdn2 --pass fwd,dgrad --sweep par_split=0,1 2>/dev/null | grep -E "kn1|opt" | cut -c1-140
for rep in 1 2; do
for v in 0 1; do
SG_PAR_SPLIT=$v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t=d['kernels']['top']; print('par_split=$v', round(d['value'],1), round(d['ms_per_step'],3), {k:t[k]['ms_per_step'] for k in t if 'kn1_k3' in k}, round(d['kernels']['all_mfma_gemms']['frac'],4), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
done
