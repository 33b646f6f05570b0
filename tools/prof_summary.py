"""print the top kernels of a rocprofv3 --stats --output-format csv run (not a test)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-100s %6d %9.1f us avg %8.2f ms tot %5.1f%%" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total %.2f ms" % (tot / 1e6))
