#!/bin/bash
# same box: two builds of the library (scratch/lib_old.so, scratch/lib_new.so), side stream on and off
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for side in 1 0; do
  for l in old new; do
    UC2_WGRAD_SIDE=$side UC2_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_$l.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib $l side $side: %.2f ms  ' % j['ms_per_step'] + ' '.join('%s %.0f' % (k['kernel'].split('<')[1][:-1].replace(' ','').replace('true','T').replace('false','F'), k['tflops']) for k in j['roofline']['all_gemm_kernels']['by_kernel']) + ' | ' + ' '.join('%s %.0f' % (h['kernel'], h['GB_per_s']) for h in j['roofline']['hbm_kernels']))
"
  done; done
done
