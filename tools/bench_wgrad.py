"""wgrad split-K sweep (not a test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
from bench_gemm import timeit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = B * 96
lib = _lib.load()
for name, m, n in [("qkv", 2304, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072), ("out", 768, 768)]:
    a = torch.randn((M, m), device="cuda").to(torch.bfloat16)
    b = torch.randn((M, n), device="cuda").to(torch.bfloat16)
    out = torch.zeros((m, n), dtype=torch.float32, device="cuda")
    for v in (0, 1, 99):
        res = []
        for split in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            fn = lambda: ops.gemm(a, b, m, n, M, ta=True, tb=True, out=out, accumulate=True, split_k=split, variant=v)
            res.append("%d:%.0f" % (split, 2.0 * m * n * M / timeit(fn, 10) / 1e12))
        print("wgrad %-5s M'=%d N'=%d K'=%d  %s  %s" % (name, m, n, M, "generic" if v == 99 else "v%d" % v, " ".join(res)))
