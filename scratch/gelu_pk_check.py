"""the FFN1 GEMM with the GELU + gelu' epilogue (and the forward-only GELU) from two library builds: bit-identical?
python scratch/gelu_pk_check.py <lib_a.so> <lib_b.so>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from uc2_amd import ops, _lib
from ab_gemm_libs import load_lib
libs = [load_lib(os.path.abspath(p)) for p in sys.argv[1:3]]
M, N, K = 8192, 3072, 768
torch.manual_seed(0)
a = (torch.randn(M, K, device="cuda") * 1.5).to(torch.bfloat16)
b = (torch.randn(N, K, device="cuda") * 0.08).to(torch.bfloat16)
bias = torch.randn(N, device="cuda") * 2
res = []
for lib in libs:
    _lib._lib = lib
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); aux = torch.empty_like(out)
    ops.gemm(a, b, M, N, K, out=out, bias=bias, epi=ops.EPI_GELU, aux_out=aux, variant=12, flags=ops.GEMM_AUX_DERIV)
    out2 = torch.empty_like(out)
    ops.gemm(a, b, M, N, K, out=out2, bias=bias, epi=ops.EPI_GELU, aux_out=None, variant=12)
    torch.cuda.synchronize()
    res.append((out, aux, out2))
pre = (a.float() @ b.float().t() + bias)
print("pre range %.1f .. %.1f" % (pre.min().item(), pre.max().item()))
print("gelu equal:", torch.equal(res[0][0], res[1][0]), " gelu' equal:", torch.equal(res[0][1], res[1][1]), " forward-only gelu equal:", torch.equal(res[0][2], res[1][2]))
ref = torch.nn.functional.gelu(pre)
print("max |gelu - erf gelu| (bf16 out): %.4f" % (res[1][0].float() - ref).abs().max().item())
