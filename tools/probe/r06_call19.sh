mkdir -p gpurun_out/r06
F=gpurun_out/r06/host_stall_gc.txt
: > $F
run() { echo "######## $*" >> $F; env "$@" python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3|garbage|Error|error" >> $F; }
run SG_LEAD_STEPS=1 PROBE_GC=on
run SG_LEAD_STEPS=1 PROBE_GC=freeze
run SG_LEAD_STEPS=1 PROBE_GC=off
cat $F
