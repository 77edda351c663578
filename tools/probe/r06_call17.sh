mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "prefetcher" > gpurun_out/r06/tests_call17.log 2>&1; tail -3 gpurun_out/r06/tests_call17.log
F=gpurun_out/r06/host_stall_kernelcopy.txt
: > $F
run() { echo "######## $*" >> $F; env "$@" python tools/probe/host_stall_probe.py 2>&1 | grep -E "^====|per step host|calls > 0.3|Error|error" >> $F; }
run SG_LEAD_STEPS=1
run SG_LEAD_STEPS=0
cat $F
