"""profiling helper (not a test): run ONE GEMM shape/variant a few times (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
variant = int(sys.argv[1]); ta = int(sys.argv[2]); tb = int(sys.argv[3]); m, n, k = map(int, sys.argv[4:7])
split = int(sys.argv[7]) if len(sys.argv) > 7 else 1
diag = int(os.environ.get("UC2_DIAG", "0"))
a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
wg = split > 1
out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device="cuda")
for _ in range(5):
    ops.gemm(a, b, m, n, k, ta=bool(ta), tb=bool(tb), out=out, accumulate=wg, split_k=split, variant=variant, flags=diag << 8)
torch.cuda.synchronize()
