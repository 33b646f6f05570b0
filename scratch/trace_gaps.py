"""gaps and busy time from a rocprofv3 kernel trace: python scratch/trace_gaps.py <kernel_trace.csv> [skip_fraction]
prints, for the last part of the run: wall time, union-busy time (any kernel running), per-stream busy time, the gap histogram between
consecutive kernels of the busiest stream and the kernels that follow the largest gaps"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows))
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
ev = [e for e in ev if e[0] >= t_lo]
wall = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d  wall %.2f ms  union-busy %.2f ms (%.1f %%)  sum of durations %.2f ms" % (len(ev), wall / 1e6, busy / 1e6, 100 * busy / wall, sum(e - s for s, e, _, _ in ev) / 1e6))
by = collections.defaultdict(list)
for e in ev: by[e[3]].append(e)
for q, l in sorted(by.items(), key=lambda kv: -len(kv[1])):
    print("  queue/stream %s: %d kernels, busy %.2f ms" % (q, len(l), sum(e - s for s, e, _, _ in l) / 1e6))
q = max(by, key=lambda k: len(by[k]))
l = by[q]
gaps = [(l[i + 1][0] - l[i][1], l[i][2][:50], l[i + 1][2][:50]) for i in range(len(l) - 1)]
h = collections.Counter()
for g, _, _ in gaps:
    h["<0" if g < 0 else "0-2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else ">30us"] += 1
print("gaps on the busiest queue:", dict(h), " total positive gap %.2f ms" % (sum(g for g, _, _ in gaps if g > 0) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g > 0: agg[(a, b)][0] += g; agg[(a, b)][1] += 1
for (a, b), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.1f us total, %4d x %5.1f us   %s -> %s" % (t / 1e3, n, t / n / 1e3, a, b))
