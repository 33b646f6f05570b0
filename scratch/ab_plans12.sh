#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for pl in v8 v12; do
    UC2_GEMM_PLANS_FILE=$GRAFT_REPO_ROOT/scratch/plans_$pl.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('plans $pl: %.2f ms  ' % j['ms_per_step'] + ' '.join('%s %.0f' % (k['kernel'].split('<')[1][:-1].replace(' ','').replace('true','T').replace('false','F'), k['tflops']) for k in j['roofline']['all_gemm_kernels']['by_kernel']))
"
  done
done
