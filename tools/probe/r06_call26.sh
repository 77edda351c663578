mkdir -p gpurun_out/r06
timeout 300 python tools/bench_gemm_classes.py --only Gres --pass wgrad --sweep w43_wgrad_tile=0,1,2,0,1,2 --iters 40 2>&1 | grep -v amdgpu > gpurun_out/r06/gemm_w43_wgrad_tile.md; cat gpurun_out/r06/gemm_w43_wgrad_tile.md
for t in 1 2; do SG_W43_WGRAD_TILE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd_f43 or conv_instnorm_fused" 2>&1 | tail -2; done
