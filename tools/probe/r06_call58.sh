timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "graphed or bit_reproducible or full_step_vs_oracle or fast_paths or tail_split or headline or n32" 2>&1 | tail -2
for rep in 1 2; do
for v in "SG_PAR_SPLIT=0 SG_W43_TAIL_SPLIT=0" "SG_PAR_SPLIT=0 SG_W43_TAIL_SPLIT=1" "SG_PAR_SPLIT=1 SG_W43_TAIL_SPLIT=1"; do
env $v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
done
