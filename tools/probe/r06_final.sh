mkdir -p gpurun_out/r06h
( time timeout 1300 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06h/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06h/smoke.log 2>&1
bash tools/gpu_profile.sh r06h/r06
