"""where the rolling kernel's time goes: v8 (full, main loop only) against v10 (full, no stores, no rolled epilogue, write-back stores)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
cases = [("fwd qkv", False, 2304, 768, True), ("fwd out", False, 768, 768, True), ("fwd ffn1 plain", False, 3072, 768, True),
         ("fwd ffn2", False, 768, 3072, True), ("dgrad out", True, 768, 768, False), ("dgrad ffn2 plain", True, 3072, 768, False)]
modes = [("v8", 8, 0, 0), ("v8 loop", 8, 8, 0), ("v10", 10, 0, 0), ("v10 nostore", 10, 1, 0), ("v10 near", 10, 8, 0), ("v10 tiled", 10, 32, 0), ("v10 wb", 10, 4, 0)]
for name, tb, n, k, bias in cases:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = torch.randn(n, device="cuda") if bias else None
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    res = {m[0]: [] for m in modes}
    for rep in range(3):
        for tag, v, dg, sk in modes:
            t = timeit(lambda: ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, variant=v, flags=(dg << 8) | (sk << 4)))
            res[tag].append(2.0 * M * n * k / t / 1e12)
    print("%-18s N=%5d K=%5d  " % (name, n, k) + "  ".join("%s %.0f" % (tag, max(res[tag])) for tag, _, _, _ in modes), flush=True)
