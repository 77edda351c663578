mkdir -p gpurun_out/r06
for lead in 0 1 2; do
  echo "######## SG_LEAD_STEPS=$lead"
  SG_LEAD_STEPS=$lead python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3"
done > gpurun_out/r06/host_stall_lead.txt 2>&1
echo "######## SG_LEAD_STEPS=0 HIP_FORCE_DEV_KERNARG=0" >> gpurun_out/r06/host_stall_lead.txt
SG_LEAD_STEPS=0 HIP_FORCE_DEV_KERNARG=0 python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3" >> gpurun_out/r06/host_stall_lead.txt
cat gpurun_out/r06/host_stall_lead.txt
