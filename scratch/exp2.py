"""diag: does a full-line store pattern speed up the ping-pong epilogue? (wrong results with diag 0x2000)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
cases = [("ffn1 gelu", False, 3072, 768, ops.EPI_GELU, True), ("ffn1 gelu nopre", False, 3072, 768, ops.EPI_GELU, False),
         ("ffn1 none", False, 3072, 768, ops.EPI_NONE, False), ("qkv none", False, 2304, 768, ops.EPI_NONE, False),
         ("ffn2 add", False, 768, 3072, ops.EPI_ADD, False), ("dgrad dgelu", True, 3072, 768, ops.EPI_DGELU, False)]
for name, tb, n, k, epi, wpre in cases:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bias = None if tb else torch.randn(n, device="cuda")
    aux = torch.randn((M, n), device="cuda").to(torch.bfloat16)
    pre = torch.empty((M, n), dtype=torch.bfloat16, device="cuda") if wpre else None
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    row = []
    for rep in range(2):
        for cg in (0, 6):
            fn = lambda: ops.gemm(a, b, M, n, k, tb=tb, out=out, bias=bias, epi=epi,
                                  aux_in=aux if epi in (ops.EPI_ADD, ops.EPI_DGELU) else None, aux_out=pre, variant=8, flags=(cg << 24) | ops.GEMM_AUX_DERIV)
            t = timeit(fn)
            if rep: row.append("cg%d %6.1f" % (cg, 2.0 * M * n * k / t / 1e12))
    print("%-18s N=%5d K=%5d  " % (name, n, k) + "  ".join(row))
