#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for g in 0 1; do
    UC2_WGRAD_GROUP=$g python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['workloads']
print('group=$g: itm %.2f ms | mlm %.2f | regime itm %.2f (%.4f) mlm %.2f (%.4f) | large bf16 %.2f fp8 %.2f | retrieval %.0f | hardneg %.2f' % (j['ms_per_step'], w['mlm']['ms_per_step'], w['reference_regime_itm']['ms_per_optimizer_step'], w['reference_regime_itm']['mfma_frac_encoder'], w['reference_regime_mlm']['ms_per_optimizer_step'], w['reference_regime_mlm']['mfma_frac_encoder'], w['uc2_large_bf16']['ms_per_step'], w['uc2_large_fp8']['ms_per_step'], w['retrieval_inference']['pairs_per_s'], w['hard_negative_finetune']['ms_per_step']))
"
  done
done
