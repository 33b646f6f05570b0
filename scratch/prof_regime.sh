#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3/prof_regime
python3 $R/scratch/regime_step.py itm 8
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/prof_regime -o p --output-format csv -- python3 $R/scratch/regime_step.py itm 8 > $R/gpurun_out/r3/prof_regime/out.txt 2>&1
tail -1 $R/gpurun_out/r3/prof_regime/out.txt
f=$(ls $R/gpurun_out/r3/prof_regime/*kernel_stats.csv | head -1)
python3 - $f <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=11
print("total kernel time per optimizer step %.2f ms" % (tot/n/1e6))
for r in rows[:32]: print("%6.2f ms/step %5.1f%% calls/step %5.1f avg %8.1f us  %s" % (float(r["TotalDurationNs"])/n/1e6, float(r["Percentage"]), int(r["Calls"])/n, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
rm -f $R/gpurun_out/r3/prof_regime/*kernel_trace.csv
