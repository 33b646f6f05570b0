"""v8 output store flavours (diag bits 1..5): nt (default), sc1, sc0 sc1, sc0, nt sc1, plain"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = 98304
names = ["nt", "sc1", "sc0 sc1", "sc0", "nt sc1", "plain"]
for name, tb, n, k, epi, fl in [("fwd ffn1 none", False, 3072, 768, ops.EPI_NONE, 0), ("fwd ffn1 gelu+gelu'", False, 3072, 768, ops.EPI_GELU, ops.GEMM_AUX_DERIV),
                                ("fwd qkv none", False, 2304, 768, ops.EPI_NONE, 0), ("dgrad out none", True, 768, 768, ops.EPI_NONE, 0)]:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = None if tb else torch.randn(n, device="cuda")
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    aux_out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda") if epi == ops.EPI_GELU else None
    ref = ops.gemm(a, b, M, n, k, tb=tb, bias=bv, epi=epi, aux_out=aux_out, variant=8, flags=fl).clone()
    res = {m: [] for m in range(6)}
    oks = []
    for m in range(6):
        o = torch.full((M, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm(a, b, M, n, k, tb=tb, bias=bv, epi=epi, aux_out=aux_out, out=o, variant=8, flags=fl | (m << 8))
        torch.cuda.synchronize()
        oks.append(torch.equal(o, ref))
    for rep in range(3):
        for m in range(6):
            t = timeit(lambda: ops.gemm(a, b, M, n, k, tb=tb, bias=bv, epi=epi, aux_out=aux_out, out=out, variant=8, flags=fl | (m << 8)))
            res[m].append(2.0 * M * n * k / t / 1e12)
    print("%-22s " % name + "  ".join("%s %.0f%s" % (names[m], max(res[m]), "" if oks[m] else "(WRONG)") for m in range(6)), flush=True)
