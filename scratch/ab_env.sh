#!/bin/bash
# same box, one library: bench.py with an environment switch off / on, alternating:  scratch/ab_env.sh VAR "0 1" [bench args]
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; shift 2
for i in 1 2; do
  for v in $vals; do
    env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v: %.2f ms/step  %.4f  ' % (j['ms_per_step'], j['mfma_frac_encoder']) + ' '.join('%s %.0f' % (h['kernel'], h['GB_per_s']) for h in j['roofline']['hbm_kernels']))
"
  done
done
