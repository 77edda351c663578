mkdir -p gpurun_out/r06
python tools/host_profile.py > gpurun_out/r06/host_profile.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
for rep in 1 2; do
  for kf in 0 256; do SG_W43_KFOLD=$kf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kfold $kf rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'], d['clocks'].get('sclk_mhz'), d['clocks'].get('power_w'))" ; done
  for ns in 1 2; do SG_W43_NSUB=$ns $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nsub $ns rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])" ; done
done > gpurun_out/r06/ab_kfold_nsub.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_gemm_classes.py --only Gdn1,Gup4,D1s1,D2s0,Gres --pmc gpurun_out/r06/pmc_classes > gpurun_out/r06/pmc_classes.log 2>&1
python tools/pmc_raw_summary.py $(find gpurun_out/r06/pmc_classes/pass0 -name "*.db" | head -1) > gpurun_out/r06/pmc_classes_mix.md 2>&1
python tools/pmc_sq_db_summary.py $(find gpurun_out/r06/pmc_classes/pass1 -name "*.db" | head -1) 2360 60 > gpurun_out/r06/pmc_classes_cycles.md 2>&1
find gpurun_out/r06/pmc_classes -name "*.db" -size +20M -delete
tail -5 gpurun_out/r06/ab_kfold_nsub.txt
