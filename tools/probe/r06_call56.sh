mkdir -p gpurun_out/r06
python tools/bench_gemm_classes.py --only Gup1,Gup2,Gup3,Gup4,Gdn4,Gdn3,Gdn2,Gdn1,ObjD --pass fwd,dgrad --sweep par_split=0,1 2>/dev/null | grep -E "kn1|opt" | cut -c1-140 > gpurun_out/r06/gemm_par_split.md
cat gpurun_out/r06/gemm_par_split.md
for rep in 1 2 3; do
for v in 0 1; do
SG_PAR_SPLIT=$v python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('par_split=$v', round(d['value'],1), round(d['ms_per_step'],3), d.get('repeat'), 'sclk', d['clocks']['sclk_mhz']['median'])"
done
done
