mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_mstep.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/suite_mstep.log
for v in "SG_STREAM_GROUPS=front" "SG_STREAM_GROUPS=front,mstep" "SG_STREAM_GROUPS=front" "SG_STREAM_GROUPS=front,mstep"; do
for leg in c5 c4; do
echo "$v $leg $(env $v python tools/run_leg.py $leg 8 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_mstep_stream_legs.txt
done
done
