#!/bin/bash
# same box, alternating: scratch/ab_env2.sh "<env assignments A>" "<env assignments B>" [bench args]   (use "_" for no assignment)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for v in "$A" "$B"; do
    e=$v; [ "$v" = "_" ] && e=""
    echo -n "[$v] "; env $e python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['mfma_frac_encoder'])"
  done
done
