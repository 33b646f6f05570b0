#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2; do
  for f in "|0" "1:0|0" "1:0|1" "|1"; do
    s=${f%%|*}; q=${f##*|}
    UC2_WGRAD_SIDE=$s UC2_GEMM_QUEUE=$q python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('side=\"$s\" queue=$q: %.2f ms  loss %.4f' % (j['ms_per_step'], j['config']['final_loss']))
"
  done
done
