"""main-loop-only and full rates of the ping-pong kernel (for the PP_DIAG16 timing build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
for name, ta, tb, m, n, k, split in [("fwd ffn1", False, False, M, 3072, 768, 1), ("fwd ffn2", False, False, M, 768, 3072, 1),
                                     ("dgrad ffn2", False, True, M, 3072, 768, 1), ("wgrad ffn1", True, True, 3072, 768, M, 7),
                                     ("wgrad qkv", True, True, 2304, 768, M, 9)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
    wg = ta and tb
    out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device="cuda")
    row = []
    for rep in range(2):
        for diag in (0, 8):
            t = timeit(lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, accumulate=wg, split_k=split, variant=8, flags=diag << 8))
            if rep: row.append("%s %6.1f" % ("full" if diag == 0 else "loop", 2.0 * m * n * k / t / 1e12))
    print("%-11s " % name + "  ".join(row))
