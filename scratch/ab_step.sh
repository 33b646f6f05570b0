#!/bin/bash
# same box, same plans: the training step with the old epilogue route (v_permlane32_swap) and with the new one, twice each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2; do
  for f in 0x2000 0; do
    UC2_GEMM_EXTRA_FLAGS=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']['all_gemm_kernels']['by_kernel']
print('flags $f: %.2f ms  ' % j['ms_per_step'] + '  '.join('%s %.0f' % (k['kernel'].split('<')[1][:-1].replace(' ',''), k['tflops']) for k in r))
"
  done
done
