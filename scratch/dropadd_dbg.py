import torch, sys
from uc2_amd import ops, _lib
lib = _lib.load()
M, N, K = 512, 768, 768
x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16() * 0.03
res = torch.randn(M, N, device="cuda").bfloat16(); out = torch.empty_like(res)
seed = torch.tensor([4321], dtype=torch.int64, device="cuda")
for p in (0.0, 0.1):
    rc = lib.uc2_gemm_drop_residual(M, N, K, x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, None, res.data_ptr(), N, p, seed.data_ptr(), 35, 0, None, None)
    torch.cuda.synchronize()
    print("p", p, "rc", rc, lib.uc2_last_error(), flush=True)
