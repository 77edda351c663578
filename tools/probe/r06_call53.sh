timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "factored or sparse or full_step_vs_oracle or fast_paths or full_width" 2>&1 | tail -2
for rep in 1 2; do
for v in 3 1; do
SG_FIXEDTAP=$v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t=d['kernels']['top']; print('vecA=' + ('off' if $v==3 else 'on'), round(d['value'],1), round(d['ms_per_step'],3), {k:t[k]['ms_per_step'] for k in t if 'nk_k7' in k}, 'sclk', d['clocks']['sclk_mhz']['median'])"
done
done
