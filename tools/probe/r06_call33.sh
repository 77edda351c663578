mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_lastblock.txt; : > $out
for rep in 1 2; do
for v in 0 2 4 6 1 7; do
  SG_LAST_BLOCK=$v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('last_block=$v', round(d['value'],1), round(d['ms_per_step'],3), 'sclk', d['clocks']['sclk_mhz']['median'])" >> $out 2>&1
done; done
cat $out
