mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "conv2d or embedding or factored or config5_step_vs or mask_net or upconv" > gpurun_out/r06/tests_call11.log 2>&1; tail -2 gpurun_out/r06/tests_call11.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r06/c5trace -o c5 -- python tools/run_leg.py c5 8 > gpurun_out/r06/c5_leg.txt 2>&1
python tools/prof_db_summary.py gpurun_out/r06/c5trace/c5_results.db 2 45 > gpurun_out/r06/c5_kernel_stats.md 2>&1
rm -rf gpurun_out/r06/c5trace
for i in 1 2 3; do python tools/run_leg.py c5 10 2>/dev/null | tail -1; done > gpurun_out/r06/c5_leg_plain.txt
cat gpurun_out/r06/c5_leg_plain.txt; grep -E "embedding|head_|head1x1" gpurun_out/r06/c5_kernel_stats.md
# does a hipGraph replay act as a barrier for the host?  host-fed rate with the graphs on and off
for g in 1 0; do SG_GRAPHS=$g python bench.py --steps 20 --warmup 5 --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graphs $g', round(d['value'],1), 'host_buffers', round(d['secondary']['host_buffers']['images_per_s'],1), 'issue', round(d['host_issue_ms_per_step'],1), 'iso', round(d['host_issue_isolated_ms_per_step'],1))"; done > gpurun_out/r06/graphs_on_off_hostfed.txt 2>&1
cat gpurun_out/r06/graphs_on_off_hostfed.txt
