python -m pytest tests/test_gpu_model.py -q -s -k "eval_vs_reference_golden or train_grads_vs_reference" 2>&1 | grep -E "max\|dlogit\||worst|passed|failed" | head -30
python -m pytest tests/test_gpu_fullsize.py -q -s -k "parity_values" 2>&1 | grep -E "passed|failed" 
cat gpurun_out/parity_values.json | python -c "
import json,sys
d=json.load(sys.stdin)
for p in ('bf16','parity'):
    b=d[p]['bert_base_L512']; print(p, 'eval', b['eval']['max_dlogit'], b['eval']['mean_dlogit'], 'train', {k:b['train_step'][k] for k in ('loss_rel_delta','gradnorm_max_rel_err','stored_grad_max_rel_err_excl_qk_bias','stored_grad_min_cosine')}, 'cfg1', d[p]['config1_bert_base']['max_dlogit'])"
