"""Timeline summary of a rocprofv3 --kernel-trace run (the *_kernel_trace.csv): what the per-kernel duration sums of --stats cannot
show when kernels of two streams overlap or the GPU waits for the host.

    python tools/timeline.py <dir with *_kernel_trace.csv> [marker kernel substring] [skip windows]

The trace is cut into windows at every launch of the marker kernel (default: adamw_kernel = one optimizer step); the first
`skip windows` (default 3: warm-up) are dropped.  Per window (averaged): wall time, time with at least one kernel running (busy),
idle time, time with two or more kernels running, and per kernel name: launches, summed duration, and EXCLUSIVE time = the part of
the window's wall time attributed to it (each instant is split evenly between the kernels running at that instant).  Also the
largest idle gaps with the kernels before and after them."""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    if not f:
        raise SystemExit("no *_kernel_trace.csv under " + d)
    rows = []
    for r in csv.DictReader(open(f[-1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    return rows


def short(n):
    n = n.replace("void ", "")
    i = n.find("(")
    return (n[:i] if i > 0 else n)[:78]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = load(d)
    cuts = [e for (s, e, n, q) in rows if marker in n]
    if len(cuts) < skip + 2:
        raise SystemExit("marker kernel %r launched %d times" % (marker, len(cuts)))
    t0, t1 = cuts[skip - 1], cuts[-1]
    nwin = len(cuts) - skip
    ks = [(s, e, short(n), q) for (s, e, n, q) in rows if s >= t0 and e <= t1 + 1]
    # sweep
    ev = []
    for i, (s, e, n, q) in enumerate(ks):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    live = set()
    last = t0
    busy = idle = multi = 0
    excl = defaultdict(float)
    gaps = []
    prev_end_name = "(window start)"
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            if not live:
                idle += dt
                gaps.append((dt, prev_end_name, None, last))
            else:
                busy += dt
                if len(live) > 1:
                    multi += dt
                for j in live:
                    excl[ks[j][2]] += dt / len(live)
        if kind == 1:
            if not live and gaps and gaps[-1][2] is None:
                g = gaps[-1]
                gaps[-1] = (g[0], g[1], ks[i][2], g[3])
            live.add(i)
        else:
            live.discard(i)
            prev_end_name = ks[i][2]
        last = t
    wall = t1 - t0
    tot = defaultdict(lambda: [0, 0])
    for s, e, n, q in ks:
        tot[n][0] += 1
        tot[n][1] += e - s
    print("windows: %d (marker %s), per window: wall %.3f ms  busy %.3f  idle %.3f  >=2 kernels %.3f" %
          (nwin, marker, wall / nwin / 1e6, busy / nwin / 1e6, idle / nwin / 1e6, multi / nwin / 1e6))
    print("%-80s %7s %10s %10s %8s" % ("kernel", "calls", "sum ms", "excl ms", "avg us"))
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -excl[kv[0]])[:40]:
        print("%-80s %7.1f %10.3f %10.3f %8.1f" % (n, c / nwin, t / nwin / 1e6, excl[n] / nwin / 1e6, t / c / 1e3))
    # idle gaps grouped by (before, after)
    gg = defaultdict(lambda: [0, 0])
    for dt, a, b, _ in gaps:
        gg[(a, b)][0] += 1
        gg[(a, b)][1] += dt
    print("idle gaps by (kernel before -> kernel after), per window:")
    for (a, b), (c, t) in sorted(gg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %8.3f ms %6.1f x %6.1f us   %s -> %s" % (t / nwin / 1e6, c / nwin, t / c / 1e3, a[:60], (b or "?")[:60]))
    qs = defaultdict(int)
    for s, e, n, q in ks:
        qs[q] += e - s
    print("per queue busy (sum of durations) ms per window:", {q: round(t / nwin / 1e6, 3) for q, t in qs.items()})


if __name__ == "__main__":
    main()
