mkdir -p gpurun_out/r06
python tools/bench_norm.py > gpurun_out/r06/bench_norm.md 2>&1; cat gpurun_out/r06/bench_norm.md
python -m pytest tests/test_gpu_parity.py -q -x -k "instnorm or instance_norm or norm" 2>&1 | tail -3
