#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the N = 3072, K = 768 forward GEMM: v8 full / main loop only, v10 full / no stores / near stores
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3/pmc
for mode in "8 0" "8 8" "10 0" "10 1" "10 8"; do
  tag=$(echo $mode | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/r3/pmc/${tag}_$c -o p --output-format csv -- python3 $R/scratch/roll5.py $mode > /dev/null 2>&1
  done
done
cd $R && python3 - <<'P'
import csv, glob, os
for d in sorted(glob.glob("gpurun_out/r3/pmc/*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_bf16" in r["Kernel_Name"]]
        if rows:
            v = [float(r["Counter_Value"]) for r in rows]
            print(os.path.basename(d), rows[0]["Counter_Name"], "launches", len(v), "mean %.1f MB" % (sum(v) / len(v) / 1e6), rows[0]["Kernel_Name"][:60])
P
