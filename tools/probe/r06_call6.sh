mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "conv2d or conv_transpose or upconv or linear" > gpurun_out/r06/tests_conv_prio.log 2>&1
python tools/bench_gemm_classes.py --only Gdn1,Gdn2,Gup3,Gup4,Gdn3,Gup2 --pass fwd,dgrad --sweep par_xcd_chunk=0,16,64 --iters 10 > gpurun_out/r06/gemm_parchunk.md 2>&1
python tools/bench_gemm_classes.py --sweep wave_prio=0,1 --iters 10 > gpurun_out/r06/gemm_prio.md 2>&1
python tools/bench_gemm_classes.py --only D1s1,D2s1,D3s1,ObjD --pass fwd,dgrad --sweep split_kmin=2048,1024,512 --iters 10 > gpurun_out/r06/gemm_splitkmin.md 2>&1
python tools/probe/timeline_probe.py --layer Gup4 --pass fwd > gpurun_out/r06/timeline2_Gup4_fwd.txt 2>&1
python tools/probe/timeline_probe.py --layer D2s0 --pass dgrad > gpurun_out/r06/timeline2_D2s0_dgrad.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
for rep in 1 2; do
  for pr in 0 1; do SG_WAVE_PRIO=$pr $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio $pr rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])" ; done
  for pc in 0 16; do SG_PAR_XCD_CHUNK=$pc $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parchunk $pc rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])" ; done
done > gpurun_out/r06/ab_prio_parchunk.txt 2>&1
tail -3 gpurun_out/r06/tests_conv_prio.log; cat gpurun_out/r06/ab_prio_parchunk.txt
