"""average rocprofv3 --pmc counters per kernel (not a test): python tools/pmc_dump.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if sub in k:
        print(k[:110])
        for c, v in sorted(cs.items()):
            print("   %-34s n=%5d  avg %16.1f" % (c, len(v), sum(v) / len(v)))
