"""power limit, as a table: the ping-pong GEMM (variant 12) on 256 / 248 / 240 / 232 / 224 CUs (UC2_GEMM_SPARE leaves 8 n CUs without a
workgroup), two bench shapes.  Run under
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 scratch/cu_clock.py
and summarise with  python3 scratch/cu_clock.py summarize <dir>  (groups the dispatches by grid size: effective clock =
GRBM_GUI_ACTIVE / 8 / duration, MI355X_MICROARCH.md 'DVFS give-back')."""
import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "summarize":
    import csv, glob, collections
    d = sys.argv[2]
    dur, grid, name = {}, {}, {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Dispatch_Id"]
            dur[k] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            grid[k] = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
            name[k] = r["Kernel_Name"]
    cnt = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[r["Dispatch_Id"]] = float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for k in dur:
        if "gemm_bf16_pp16_kernel" in name[k] and k in cnt:
            acc[(name[k].split("(")[0].replace("void ", ""), grid[k])].append((dur[k], cnt[k]))
    print("| kernel | workgroups (= CUs used) | launches | avg duration us | GRBM_GUI_ACTIVE / 8 / duration = clock GHz |")
    print("|---|---|---|---|---|")
    for (kn, g), v in sorted(acc.items(), key=lambda t: (t[0][0], -t[0][1])):
        v = v[len(v) // 4:]                       # drop the first quarter (warm-up of each setting)
        du = sum(x[0] for x in v) / len(v)
        ck = sum(x[1] / 8.0 / x[0] for x in v) / len(v)
        print("| %s | %d | %d | %.1f | %.3f |" % (kn, g // 512, len(v), du / 1e3, ck))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
for (ta, tb, m, n, k, sp) in ((False, False, 98304, 2304, 768, 1), (False, False, 98304, 768, 3072, 1)):
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    bias = torch.zeros(n, device="cuda")
    for spare in (0, 1, 2, 3, 4, 0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(44):
            if i == 4:
                e0.record()
            ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, bias=bias, variant=12, flags=(spare & 7) << 28)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 40 * 1e-3
        print("%dx%dx%d  CUs %3d : %.1f us  %.0f TF/s" % (m, n, k, 256 - 8 * spare, t * 1e6, 2.0 * m * n * k / t / 1e12), flush=True)
