#!/bin/bash
# same box: the committed plans (ping-pong v8/v9) against the freshly tuned table (rolling v10 where the tuner picked it)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2; do
  for f in uc2_amd/gemm_plans.json scratch/plans_r3_tuned.json; do
    UC2_GEMM_PLANS_FILE=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']['all_gemm_kernels']['by_kernel']
print('$f: %.2f ms  ' % j['ms_per_step'] + '  '.join('%s %.0f/%.2f' % (k['kernel'].replace('gemm_bf16_','').replace('_kernel','').replace(' ',''), k['tflops'], k['ms_per_step']) for k in r))
"
  done
done
