mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ktrace_r
PROBE_MODES=resident PROBE_REPS=1 rocprofv3 --kernel-trace -f csv -d /tmp/ktrace_r -o t -- python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host" > gpurun_out/r06/kgaps_pairs.txt
python tools/probe/kernel_gaps.py /tmp/ktrace_r 300 680 >> gpurun_out/r06/kgaps_pairs.txt 2>&1
cat gpurun_out/r06/kgaps_pairs.txt | cut -c1-260
