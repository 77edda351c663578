for v in 0 1; do for rep in 1 2 3; do
echo "cond_fold=$v rep $rep"
SG_COND_FOLD=$v timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "full_step_vs_oracle and reduced" 2>&1 | grep -E "passed|failed|relative L2" | head -3
python -c "
import json; d=json.load(open('gpurun_out/grad_rel_l2_reduced.json')); print({k: '%.2e' % v for k, v in d.items()})" 2>/dev/null
done; done
