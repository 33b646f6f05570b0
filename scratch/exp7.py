"""MLM decoder input gradient dz[m,768] = dlogits[m,250112] . E: direct bf16 kernels vs split over the vocabulary (fp32 partials)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
K, N = 250112, 768
b = (torch.randn((K, N), device="cuda") * 0.05).to(torch.bfloat16)
for m in (8192, 1024):
    a = (torch.randn((m, K), device="cuda") * 0.01).to(torch.bfloat16)
    o16 = torch.empty((m, N), dtype=torch.bfloat16, device="cuda")
    o32 = torch.zeros((m, N), dtype=torch.float32, device="cuda")
    row = []
    for v in (7, 6, 1, 8):
        t = timeit(lambda: ops.gemm(a, b, m, N, K, tb=True, out=o16, variant=v))
        row.append("bf16 v%d %.0f us" % (v, t * 1e6))
    for v, sp in ((8, 2), (8, 3), (8, 5), (8, 8), (8, 16), (1, 4), (1, 8), (6, 8)):
        t = timeit(lambda: ops.gemm(a, b, m, N, K, tb=True, out=o32, accumulate=True, split_k=sp, variant=v))
        row.append("f32 v%d/%d %.0f us" % (v, sp, t * 1e6))
    print("m=%d: " % m + "  ".join(row), flush=True)
