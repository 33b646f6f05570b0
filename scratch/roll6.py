"""experiment build (UC2_ROLL_DIAG=2: output lines prefetched into L2 during the main loop): v8 vs v10, K >= 640 only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
cases = [("dgrad ffn2 plain", True, 3072, 768, False), ("dgrad out", True, 768, 768, False), ("dgrad qkv plain", True, 768, 2304, False),
         ("fwd ffn1 plain", False, 3072, 768, True), ("fwd qkv", False, 2304, 768, True)]
for name, tb, n, k, bias in cases:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = torch.randn(n, device="cuda") if bias else None
    ref = ops.gemm(a, b, M, n, k, tb=tb, bias=bv, variant=8)
    out = torch.full((M, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, variant=10)
    same = torch.equal(out, ref)
    res = {8: [], 10: [], "10 nostore": [], "8 loop": []}
    for rep in range(4):
        for key, v, dg in ((8, 8, 0), (10, 10, 0), ("10 nostore", 10, 1), ("8 loop", 8, 8)):
            t = timeit(lambda: ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, variant=v, flags=dg << 8))
            res[key].append(2.0 * M * n * k / t / 1e12)
    print("%-18s N=%5d K=%5d same %s | v8 %.0f  v8 loop %.0f  v10+prefetch %.0f  v10 nostore %.0f  (v10/v8 %+.1f %%)" % (
        name, n, k, same, max(res[8]), max(res["8 loop"]), max(res[10]), max(res["10 nostore"]), 100 * (max(res[10]) / max(res[8]) - 1)), flush=True)
