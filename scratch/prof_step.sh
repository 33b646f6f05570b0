#!/bin/bash
# rocprofv3 kernel trace + stats of the headline bench (ITM step), summary copied to gpurun_out/r3/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3/prof_itm
UC2_WGRAD_SIDE=${SIDE:-1} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/prof_itm -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/r3/prof_itm/bench.json 2> $R/gpurun_out/r3/prof_itm/err.txt
ls $R/gpurun_out/r3/prof_itm | head
f=$(ls $R/gpurun_out/r3/prof_itm/*kernel_stats.csv | head -1)
python3 - $f <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step %.2f ms (13 steps)" % (tot/13e6))
for r in rows[:22]: print("%6.2f ms/step %5.1f%% calls/step %4.0f avg %8.1f us  %s" % (float(r["TotalDurationNs"])/13e6, float(r["Percentage"]), int(r["Calls"])/13, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
rm -f $R/gpurun_out/r3/prof_itm/*kernel_trace.csv
