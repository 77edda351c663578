mkdir -p gpurun_out/r06
F=gpurun_out/r06/host_stall_memmove.txt
: > $F
run() { echo "######## $*" >> $F; env "$@" python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3|cpus:|cpu.stat|Error|error" >> $F; }
run SG_LEAD_STEPS=1
run SG_LEAD_STEPS=0
cat $F
