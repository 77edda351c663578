mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "conv2d or factored or sparse or full_step_n32" > gpurun_out/r06/tests_call9.log 2>&1; tail -2 gpurun_out/r06/tests_call9.log
python tools/bench_gemm_classes.py --only Gdn1,Gdn2,Gdn3,Gdn4,Gup1,Gup2,Gup3,Gup4,D1s0,D2s0,D1s0x2,D2s0x2,ObjD --sweep tile=-1,0,1,3 --iters 10 > gpurun_out/r06/gemm_tile_sweep.md 2>&1
for spec in "D2s0 dgrad" "Gres fwd" "Gup4 fwd" "D1s1 fwd"; do set -- $spec; python tools/probe/timeline_probe.py --layer $1 --pass $2 > gpurun_out/r06/timeline3_$1_$2.txt 2>&1; done
B="python bench.py --steps 20 --warmup 5 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
for rep in 1 2 3; do $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head rep $rep', round(d['value'],1), round(d['ms_per_step'],3), d['repeat']['ms_per_step_blocks'])"; done > gpurun_out/r06/bench_call9.txt 2>&1
cat gpurun_out/r06/bench_call9.txt
