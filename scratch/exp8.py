"""FFN1 forward GEMM (98304 x 3072 x 768): what the GELU epilogue and its second output stream cost"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
m, n, k = 98304, 3072, 768
a = torch.randn((m, k), device="cuda").to(torch.bfloat16)
b = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(n, device="cuda")
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
aux = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
cases = [("plain + bias", {}), ("gelu, one stream", dict(epi=ops.EPI_GELU)), ("gelu + pre-activation stream", dict(epi=ops.EPI_GELU, aux_out=aux)),
         ("gelu + gelu' stream", dict(epi=ops.EPI_GELU, aux_out=aux, flags=ops.GEMM_AUX_DERIV)),
         ("tanh, one stream", dict(epi=ops.EPI_TANH)), ("residual add (aux_in)", dict(epi=ops.EPI_ADD, aux_in=aux))]
for rep in range(2):
    for name, kw in cases:
        t = min(timeit(lambda: ops.gemm(a, b, m, n, k, bias=bias, out=out, variant=8, **kw)) for _ in range(2))
        if rep: print("%-30s %7.1f TF/s %6.0f us" % (name, 2.0 * m * n * k / t / 1e12, t * 1e6), flush=True)
