mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in inline resident; do
  rm -rf /tmp/ktrace_$m
  SG_LEAD_STEPS=1 PROBE_MODES=$m PROBE_REPS=1 rocprofv3 --kernel-trace -f csv -d /tmp/ktrace_$m -o t -- python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host" > gpurun_out/r06/kgaps_$m.txt
  python tools/probe/kernel_gaps.py /tmp/ktrace_$m 300 680 >> gpurun_out/r06/kgaps_$m.txt 2>&1
done
cat gpurun_out/r06/kgaps_inline.txt; cat gpurun_out/r06/kgaps_resident.txt
