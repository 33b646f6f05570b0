#!/bin/bash
# same box: MLM step with two builds of the library
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for l in old new; do
    UC2_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_$l.so python bench.py --task mlm --steps 10 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib $l: %.2f ms' % j['ms_per_step'])
"
  done
done
