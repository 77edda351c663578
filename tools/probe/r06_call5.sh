mkdir -p gpurun_out/r06
for spec in "Gup4 fwd" "Gdn1 dgrad" "Gdn1 fwd" "D1s1 fwd" "D2s0 dgrad"; do set -- $spec; python tools/probe/timeline_probe.py --layer $1 --pass $2 > gpurun_out/r06/timeline_$1_$2.txt 2>&1; done
python -m pytest tests/test_gpu_parity.py -q -x -k "conv_instnorm or winograd_f43 or full_step_n32 or reproducible or graphed_segments" > gpurun_out/r06/tests_biasgrad.log 2>&1
cp gpurun_out/parity_headline_n32.json gpurun_out/r06/parity_headline_n32_kfold256.json
cp gpurun_out/winograd_f43_errors.json gpurun_out/r06/f43_errors_default.json
tail -3 gpurun_out/r06/tests_biasgrad.log; cat gpurun_out/r06/timeline_Gup4_fwd.txt
