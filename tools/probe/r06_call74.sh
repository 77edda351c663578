mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_imgd.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/suite_imgd.log
for i in 1 2; do
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "side_stream or bit_reproducible or graphed or multiscale" 2>&1 | tail -1
done
