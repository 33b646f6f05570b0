# re-time the committed plan entries that use the ping-pong kernel (variants 8 / 9) against variant 12 (16x16x32 MFMA) with the
# epilogue the encoder uses for that shape; writes the updated table
import sys, json, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
src, dst = sys.argv[1], sys.argv[2]
plans = json.load(open(src))
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = dict(plans)
for key, (v, sp) in sorted(plans.items()):
    parts = key.split()
    if v not in (8, 9):
        continue
    ta, tb = parts[0][0] == "T", parts[0][1] == "T"
    wgrad = len(parts) > 2
    M, N, K = (int(x) for x in parts[1].split("x"))
    if (M % 256) or (N % 256) or (K % 128) or float(M) * N * K > 6e13:
        continue
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=dev, dtype=torch.bfloat16) * 0.03
    if wgrad:
        c = torch.zeros(M, N, device=dev)
        fn = lambda var: ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=c, accumulate=True, split_k=sp, variant=var)
    else:
        bias = torch.randn(N, device=dev) if not tb else None
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda var: ops.gemm(a, b, M, N, K, ta=ta, tb=tb, out=o, bias=bias, variant=var)
    t = {}
    for var in (v, 12, v, 12):
        t[var] = min(t.get(var, 1e9), timeit(lambda: fn(var)))
    pick = 12 if t[12] < 0.99 * t[v] else v
    print("%-34s v%d %.1f us   v12 %.1f us   %+.1f %%  -> v%d" % (key, v, t[v] * 1e3, t[12] * 1e3, (t[v] / t[12] - 1) * 100, pick), flush=True)
    out[key] = [pick, sp]
    del a, b
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
