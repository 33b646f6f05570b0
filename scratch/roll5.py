"""one GEMM mode per process (for rocprofv3 --pmc): python roll5.py <variant> <diag> [n k tb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
v, dg = int(sys.argv[1]), int(sys.argv[2])
n, k, tb = (int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))) if len(sys.argv) > 5 else (3072, 768, False)
M = 98304
a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
bv = None if tb else torch.randn(n, device="cuda")
out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, variant=v, flags=dg << 8)
torch.cuda.synchronize()
