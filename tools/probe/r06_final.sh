# round-6 final validation on the GPU box: suite, default bench, kernel trace, PMC traffic tables
mkdir -p gpurun_out/r06f
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06f/smoke.log 2>&1
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06f/gpu_tests.log 2>&1; echo TESTS_RC=$? >> gpurun_out/r06f/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
SG_GRAPHS=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r06f/ktrace -o kt -- $B > gpurun_out/r06f/ktrace.log 2>&1
python tools/prof_db_summary.py gpurun_out/r06f/ktrace/kt_results.db 2 50 > gpurun_out/r06f/kernel_stats.md 2>&1
python tools/prof_db_summary.py gpurun_out/r06f/ktrace/kt_results.db 2 400 --by-grid > gpurun_out/r06f/kernel_stats_full.md 2>&1
B3="python bench.py --steps 3 --warmup 3 --no_legs --no_secondary --cpu_baseline off --pmc off --no_prof"
SG_GRAPHS=0 SG_LAUNCH_LOG=$GRAFT_REPO_ROOT/gpurun_out/r06f/launch.log rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r06f/pmc_fetch -o f -- $B3 > gpurun_out/r06f/pmc_fetch.log 2>&1
SG_GRAPHS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r06f/pmc_write -o w -- $B3 > gpurun_out/r06f/pmc_write.log 2>&1
python tools/pmc_db_summary.py $(find gpurun_out/r06f/pmc_fetch -name "*.db" | head -1) $(find gpurun_out/r06f/pmc_write -name "*.db" | head -1) --json gpurun_out/r06f/pmc_traffic.json --launch-log gpurun_out/r06f/launch.log > gpurun_out/r06f/pmc_traffic.md 2>&1
rm -rf gpurun_out/r06f/pmc_fetch gpurun_out/r06f/pmc_write gpurun_out/r06f/ktrace
tail -3 gpurun_out/r06f/gpu_tests.log; head -c 400 gpurun_out/r06f/bench.json; tail -2 gpurun_out/r06f/smoke.log
