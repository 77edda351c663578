for seed in 0 100 200 300 400 500; do
echo "seed $seed"
SG_TEST_REDUCED_SEED=$seed SG_TEST_REDUCED_TOL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "full_step_vs_oracle and reduced" 2>&1 | grep -E "passed|failed|Error" | head -3
python -c "
import json; d=json.load(open('gpurun_out/grad_rel_l2_reduced.json')); print({k: '%.2e' % v for k, v in d.items()})" 2>/dev/null
rm -f gpurun_out/grad_rel_l2_reduced.json
done
