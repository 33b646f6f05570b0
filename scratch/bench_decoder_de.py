"""dE[Vp, 768] += dlogits^T z per decoder chunk (TT, M = 250 112, N = 768, K = rows): split-K factors / variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import torch
from bench_gemm import timeit
from uc2_amd import ops
from uc2_amd.ops.gemm import _plan_fits
K = int(sys.argv[1]) if len(sys.argv) > 1 else 7680
M, N = 250112, 768
bf = torch.bfloat16
a = (torch.randn(K, M, device="cuda") * 0.01).to(bf)
b = (torch.randn(K, N, device="cuda") * 0.03).to(bf)
out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
res = {}
for _ in range(3):
    for v in (12, 8):
        for s in (1, 2, 3):
            if not _plan_fits((v, s), (True, True, M, N, K, True)):
                continue
            t = timeit(lambda: ops.gemm(a, b, M, N, K, ta=True, tb=True, out=out, accumulate=True, split_k=s, variant=v), 5)
            res.setdefault((v, s), []).append(t)
for (v, s), ts in res.items():
    print("variant %d split %d: %.1f us  %.0f TF/s" % (v, s, sorted(ts)[1] * 1e6, 2.0 * M * N * K / sorted(ts)[1] / 1e12))
