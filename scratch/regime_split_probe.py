"""N = 768 GEMMs at the reference's 9 984 tokens: the planned kernel against the ping-pong kernel with the contraction split in two
(fp32 partial tiles, reduction deferred) + the existing fp32 reduction pass as a stand-in for a fused reduce-and-epilogue kernel.
python scratch/regime_split_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
dev = "cuda"
M = 9984
def timeit(fn, n=30, rounds=5):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
lib = _lib.load()
for (N, K, tb, name) in ((768, 3072, False, "FFN2 fwd"), (768, 3072, True, "FFN1 dgrad"), (768, 2304, True, "QKV dgrad"), (768, 768, False, "Wo fwd"), (768, 768, True, "Wo dgrad"),
                         (2304, 768, False, "QKV fwd"), (3072, 768, False, "FFN1 fwd (plain)")):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)).bfloat16() * 0.03
    bias = torch.randn(N, device=dev)
    plan = ops.gemm_plan(torch.bfloat16, False, tb, M, N, K, False)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_plan = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, bias=None if tb else bias, variant=plan[0], split_k=plan[1]))
    t_pp = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, bias=None if tb else bias, variant=12, split_k=1))
    res = {}
    for sp in (2, 3, 4):
        if (K // 64) % (2 * sp):
            continue
        o32 = torch.zeros(M, N, dtype=torch.float32, device=dev)
        ws = torch.empty(sp * M * N, dtype=torch.float32, device=dev)
        def part():
            _lib.call("uc2_gemm", 1, 0, int(tb), M, N, K, _lib.ptr(a), K, _lib.ptr(b), b.stride(0), _lib.ptr(o32), N, 1, None, 0, None, None, 0, 1, sp, 12,
                      _lib.ptr(ws), ws.numel() * 4, ops.GEMM_DEFER_REDUCE, _lib.stream())
        def red():
            _lib.call("uc2_gemm_splitk_reduce", M, N, _lib.ptr(o32), N, sp, 1, _lib.ptr(ws), ws.numel() * 4, _lib.stream())
        t_part = timeit(part)
        t_both = timeit(lambda: (part(), red()))
        res[sp] = (t_part, t_both)
    print("%-18s N %4d K %4d plan %s: %.1f us   variant 12: %.1f us   split partial-only / +fp32 reduce: %s" %
          (name, N, K, plan, t_plan, t_pp, "  ".join("x%d %.1f / %.1f" % (sp, v[0], v[1]) for sp, v in res.items())))
