#!/bin/bash
# regime step with alternative GEMM plan files (scratch/plans/*.json), alternating with the committed table
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo -n "committed: "; python scratch/regime_step.py itm 20 2>/dev/null | tail -1
  for f in "$@"; do
    echo -n "$f: "; UC2_GEMM_PLANS_FILE=$PWD/scratch/plans/$f.json python scratch/regime_step.py itm 20 2>/dev/null | tail -1
  done
done
