#!/bin/bash
# same box: scratch/regime_step.py with an environment switch, alternating:  scratch/ab_regime_env.sh VAR "0 1" [task]
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; task=${3:-itm}
for i in 1 2 3; do
  for v in $vals; do
    echo -n "$var=$v: "; env $var=$v python scratch/regime_step.py $task 20 2>/dev/null | tail -1
  done
done
