#!/bin/bash
# same box: weight-gradient GEMMs on a side stream, leaving 0 / 16 / 24 / 32 / 48 CUs to the memory-bound kernels of the main stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2; do
  for f in "" "1:0" "1:2" "1:3" "1:4" "1:6"; do
    UC2_WGRAD_SIDE=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('side=\"$f\": %.2f ms  loss %.4f' % (j['ms_per_step'], j['config']['final_loss']))
"
  done
done
