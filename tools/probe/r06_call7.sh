mkdir -p gpurun_out/r06
B="python bench.py --steps 20 --warmup 5 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
for rep in 1 2; do
  for km in 2048 1024 512; do SG_SPLIT_KMIN=$km $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split_kmin $km rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])" ; done
  for pc in 8 16 32; do SG_PAR_XCD_CHUNK=$pc $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parchunk $pc rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])" ; done
done > gpurun_out/r06/ab_splitkmin.txt 2>&1
cat gpurun_out/r06/ab_splitkmin.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "conv_instnorm or full_step_n32 or reproducible or graphed_segments or golden" > gpurun_out/r06/tests_call7.log 2>&1; tail -2 gpurun_out/r06/tests_call7.log
