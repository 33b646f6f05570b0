"""is it burst synchronisation?  8x longer launch (skew start-up amortised), N = 3072, K = 768: v8 / v10 with start skew"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
for (M, n, k, tb) in [(98304 * 8, 3072, 768, False), (98304 * 4, 3072, 768, True)]:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = None if tb else torch.randn(n, device="cuda")
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    modes = [("v8", 8, 0, 0), ("v8 loop", 8, 8, 0), ("v8 sk1", 8, 0, 1), ("v10", 10, 0, 0), ("v10 nostore", 10, 1, 0), ("v10 near", 10, 8, 0),
             ("v10 sk2", 10, 0, 2), ("v10 sk5", 10, 0, 5), ("v10 sk9", 10, 0, 9), ("v10 sk15", 10, 0, 15)]
    res = {m[0]: [] for m in modes}
    for rep in range(2):
        for tag, v, dg, sk in modes:
            t = timeit(lambda: ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, variant=v, flags=(dg << 8) | (sk << 4)), n=5)
            res[tag].append(2.0 * M * n * k / t / 1e12)
    print("M=%d N=%d K=%d tb=%d  " % (M, n, k, tb) + "  ".join("%s %.0f" % (tag, max(res[tag])) for tag, _, _, _ in modes), flush=True)
