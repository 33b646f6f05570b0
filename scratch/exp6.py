"""column-group sweep of the ping-pong tile order (UC2_GEMM_COLGROUP) with streaming (non-temporal) output stores"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
cases = [("fwd qkv bias", False, False, M, 2304, 768, None), ("fwd ffn1 gelu", False, False, M, 3072, 768, "gelu"),
         ("dgrad ffn2 mul", False, True, M, 3072, 768, "dgelu"), ("fwd ffn2 bias", False, False, M, 768, 3072, None)]
for name, ta, tb, m, n, k, epi in cases:
    a = torch.randn((m, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda") if not tb else None
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    aux = torch.randn((m, n), device="cuda").to(torch.bfloat16)
    kw, fl = {}, 0
    if epi == "gelu": kw, fl = dict(epi=ops.EPI_GELU, aux_out=aux), ops.GEMM_AUX_DERIV
    elif epi == "dgelu": kw, fl = dict(epi=ops.EPI_DGELU, aux_in=aux), ops.GEMM_AUX_DERIV
    row = []
    for cg in (0, 1, 2, 3, 4, 6, 12):
        t = min(timeit(lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, bias=bias, out=out, variant=8, flags=fl | (cg << 24), **kw)) for _ in range(2))
        row.append("cg%d %.0f" % (cg, 2.0 * m * n * k / t / 1e12))
    print("%-15s " % name + "  ".join(row), flush=True)
