mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_e.log 2>&1
echo "rc=$?"; grep -E "passed|failed|Fatal|fault|Error" gpurun_out/r06/suite_e.log | head -5
