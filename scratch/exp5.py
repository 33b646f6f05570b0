"""the step's GEMMs with their epilogues on the ping-pong kernel (TF/s)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
M = int(os.environ.get("M", 98304))
cases = [("fwd qkv bias", False, False, M, 2304, 768, None, 1), ("fwd o bias", False, False, M, 768, 768, None, 1),
         ("fwd ffn1 gelu", False, False, M, 3072, 768, "gelu", 1), ("fwd ffn2 bias", False, False, M, 768, 3072, None, 1),
         ("dgrad ffn2 mul", False, True, M, 3072, 768, "dgelu", 1), ("dgrad ffn1 add", False, True, M, 768, 3072, "add", 1),
         ("dgrad o", False, True, M, 768, 768, None, 1), ("dgrad qkv add", False, True, M, 768, 2304, "add", 1),
         ("wgrad ffn1", True, True, 3072, 768, M, None, 7), ("wgrad qkv", True, True, 2304, 768, M, None, 9), ("wgrad o", True, True, 768, 768, M, None, 28)]
for name, ta, tb, m, n, k, epi, split in cases:
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    wg = ta and tb
    bias = torch.randn(n, device="cuda") if not tb else None
    out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device="cuda")
    aux = torch.randn((m, n), device="cuda").to(torch.bfloat16) if not wg else None
    kw = {}
    if epi == "gelu": kw = dict(epi=ops.EPI_GELU, aux_out=aux, flags=ops.GEMM_AUX_DERIV)
    elif epi == "dgelu": kw = dict(epi=ops.EPI_DGELU, aux_in=aux, flags=ops.GEMM_AUX_DERIV)
    elif epi == "add": kw = dict(epi=ops.EPI_ADD, aux_in=aux)
    f = lambda v: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, bias=bias, out=out, accumulate=wg, split_k=split, variant=v, **kw)
    ref = f(99).float().clone() if not wg else None
    got = f(8).float().clone()
    err = (got - ref).abs().max().item() / ref.abs().max().item() if ref is not None else float("nan")
    t = min(timeit(lambda: f(8)) for _ in range(3))
    print("%-15s %7.1f TF/s  %6.0f us   rel err vs generic %.2g" % (name, 2.0 * m * n * k / t / 1e12, t * 1e6, err), flush=True)
