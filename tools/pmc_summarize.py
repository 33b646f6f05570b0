"""Build profiles/rNN_pmc_summary.json from three rocprofv3 --pmc passes (not a test).

    python tools/pmc_summarize.py <round> <dir FETCH_SIZE> <dir WRITE_SIZE> <dir SQ counters> <kernel-substring> [...]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads and is doubled.
The first kernel listed is the dominant one (bench.py reads its hbm_bytes_per_launch as roofline.traffic)."""
import collections, csv, glob, json, sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(d):
    """kernel name -> average duration in ns from the kernel-trace CSV of the same pass"""
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                acc[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            except (KeyError, ValueError):
                pass
    return {k: sum(v) / len(v) for k, v in acc.items() if v}


def pick(acc, sub):
    best = None
    for k, cs in acc.items():
        if sub.replace(" ", "") in k.replace(" ", ""):
            n = max(len(v) for v in cs.values())
            if best is None or n > best[1]:
                best = (k, n, cs)
    return best


def one(sub, F, W, S, DUR=None):
    avg = lambda v: sum(v) / len(v)
    out = {}
    hit = pick(F, sub)
    if hit is None:
        return None
    k, n, cs = hit
    out["kernel"] = k.replace("void ", "").split("(")[0]
    out["launches_averaged"] = n
    fetch_kb = avg(cs["FETCH_SIZE"])
    write_kb = avg(pick(W, sub)[2]["WRITE_SIZE"])
    out["FETCH_SIZE_KB_per_launch_raw"] = round(fetch_kb, 1)
    out["WRITE_SIZE_KB_per_launch"] = round(write_kb, 1)
    out["hbm_bytes_per_launch"] = int((2.0 * fetch_kb + write_kb) * 1024)
    cs3 = pick(S, sub)[2]
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"):
        if c in cs3:
            out[c if c != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE_sum_over_8_xcd"] = round(avg(cs3[c]), 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs3 and "GRBM_GUI_ACTIVE" in cs3:
        # busy cycles are summed over 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        out["mfma_busy_frac"] = round(avg(cs3["SQ_VALU_MFMA_BUSY_CYCLES"]) / (avg(cs3["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0), 4)
    if DUR and "GRBM_GUI_ACTIVE" in cs3:
        # effective shader clock under this kernel (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE is summed over the 8 XCDs;
        # reads high on launches shorter than ~0.3 ms, and a profiled pass clocks ~2.5 % below an un-profiled one
        kn = pick(S, sub)[0]
        if kn in DUR and DUR[kn] > 0:
            out["avg_duration_us_in_pmc_pass"] = round(DUR[kn] / 1e3, 1)
            out["derived_clock_GHz"] = round(avg(cs3["GRBM_GUI_ACTIVE"]) / 8.0 / DUR[kn], 3)
            if "mfma_busy_frac" in out:
                # flop rate this kernel would have at that clock with the matrix pipes always busy: 256 CUs x 4 SIMDs x 1024 flop/cycle
                out["mfma_ceiling_at_that_clock_TFLOPs"] = round(256 * 4 * 1024 * out["derived_clock_GHz"] * 1e9 / 1e12, 1)
    if "SQ_WAIT_ANY" in cs3 and "SQ_WAVE_CYCLES" in cs3:
        out["wait_frac"] = round(avg(cs3["SQ_WAIT_ANY"]) / avg(cs3["SQ_WAVE_CYCLES"]), 4)
    return out


def main():
    rnd = int(sys.argv[1])
    F, W, S = load(sys.argv[2]), load(sys.argv[3]), load(sys.argv[4])
    DUR = durations(sys.argv[4])
    ks = [one(sub, F, W, S, DUR) for sub in sys.argv[5:]]
    ks = [k for k in ks if k]
    out = dict(ks[0])
    out["round"] = rnd
    out["command"] = ("rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 2 --warmup 2 "
                      "--no-cpu-baseline --no-extras (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                      "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES)")
    out["correction"] = ("gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled "
                         "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is; Infinity-Cache hits are included in FETCH_SIZE")
    out["other_kernels"] = ks[1:]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
