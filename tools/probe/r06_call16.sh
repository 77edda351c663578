mkdir -p gpurun_out/r06
F=gpurun_out/r06/host_stall_variants.txt
: > $F
run() { echo "######## $*" >> $F; env "$@" python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3" >> $F; }
run SG_LEAD_STEPS=1 PROBE_VARIANT=mainstream
run SG_LEAD_STEPS=1 PROBE_VARIANT=norecord
run SG_LEAD_STEPS=1 PROBE_VARIANT=x HSA_ENABLE_SDMA=0
run SG_LEAD_STEPS=1 PROBE_VARIANT=x GPU_MAX_HW_QUEUES=1
run SG_LEAD_STEPS=1 PROBE_VARIANT=x HIP_HOST_COHERENT=0
run SG_LEAD_STEPS=1 PROBE_VARIANT=x AMD_DIRECT_DISPATCH=0
cat $F
