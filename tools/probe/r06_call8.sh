mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06/gpu_tests_3.log 2>&1; echo TESTS_RC=$? >> gpurun_out/r06/gpu_tests_3.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_3.json 2> gpurun_out/r06/bench_3.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SG_GRAPHS=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/ktrace -o kt -- python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof > gpurun_out/r06/ktrace.log 2>&1
ls -R gpurun_out/r06/ktrace | head -20
tail -4 gpurun_out/r06/gpu_tests_3.log
