for v in 1 0; do
SG_COND_FOLD=$v timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "full_step_vs_oracle and reduced" 2>&1 | grep -E "passed|failed|Error" | head -3
python -c "
import json; d=json.load(open('gpurun_out/grad_rel_l2_reduced.json')); print({k: '%.2e' % v for k, v in d.items() if 'optimizer' == k.split('/')[1] or 'best' in k})" 2>/dev/null
done
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
