mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in inline resident; do
  rm -rf /tmp/hiptrace_$m
  SG_LEAD_STEPS=1 PROBE_MODES=$m PROBE_REPS=1 rocprofv3 --hip-runtime-trace -f csv -d /tmp/hiptrace_$m -o t -- python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3" > gpurun_out/r06/hip_api_$m.txt
  python tools/probe/hip_api_long_calls.py /tmp/hiptrace_$m 4 >> gpurun_out/r06/hip_api_$m.txt 2>&1
done
cat gpurun_out/r06/hip_api_inline.txt; cat gpurun_out/r06/hip_api_resident.txt | head -40
