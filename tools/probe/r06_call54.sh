for rep in 1 2; do
for v in 0 1 unset; do
if [ $v == unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', round(d['value'],1), round(d['ms_per_step'],3), 'timed kernels', round(d['kernels']['timed_kernels_ms_per_step'],2), 'host isolated', round(d['host_issue_isolated_ms_per_step'],2), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
done
