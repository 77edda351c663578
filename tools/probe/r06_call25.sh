mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd_f43 or conv_instnorm_fused or wino" > gpurun_out/r06/tests_call25.log 2>&1; tail -15 gpurun_out/r06/tests_call25.log
timeout 300 python tools/bench_gemm_classes.py --only Gres --sweep w43_tail_split=0,1,0,1 --iters 40 > gpurun_out/r06/gemm_tailsplit.md 2>&1; cat gpurun_out/r06/gemm_tailsplit.md | grep -v amdgpu
