# transposed weight copies: the batched transpose, the NN forms of the input-gradient epilogues, and the layer backward with / without
import os, sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops, _lib
dev = "cuda"
torch.manual_seed(0)
# 1. batched transpose through the store
import uc2_amd
from uc2_amd.store import store_of, set_compute_dtype
lin = torch.nn.Sequential(torch.nn.Linear(768, 3072), torch.nn.Linear(3072, 768), torch.nn.Linear(768, 768)).to(dev)
set_compute_dtype(lin, torch.bfloat16)
st = store_of(lin); st.sync_shadow()
for m in lin:
    wt = st.compute_t(m.weight)
    ref = st.compute(m.weight, torch.bfloat16).t().contiguous()
    assert torch.equal(wt, ref), "transpose"
with torch.no_grad(): lin[0].weight.mul_(2.0)
st.mark_dirty(); 
wt = st.compute_t(lin[0].weight); assert torch.equal(wt, st.compute(lin[0].weight, torch.bfloat16).t().contiguous())
wt2 = st.compute_t(lin[1].weight); assert torch.equal(wt2, st.compute(lin[1].weight, torch.bfloat16).t().contiguous())
print("transposes ok")
# 2. NN forms of the epilogues
M, N, K = 4096, 3072, 768            # dX[M,K] = dY[M,N] W[N,K]
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
wt = w.t().contiguous(); aux = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
for epi, fl, nm in ((ops.EPI_NONE, 0, "none"), (ops.EPI_ADD, 0, "add"), (ops.EPI_DGELU, 0, "dgelu"), (ops.EPI_DGELU, 2, "mul")):
    c1 = torch.zeros(K, device=dev); c2 = torch.zeros(K, device=dev)
    kw1 = dict(aux_out=c1) if epi == ops.EPI_DGELU else {}
    kw2 = dict(aux_out=c2) if epi == ops.EPI_DGELU else {}
    a_in = aux if epi != ops.EPI_NONE else None
    r1 = ops.gemm(dy, w, M, K, N, tb=True, epi=epi, aux_in=a_in, variant=12, flags=fl, **kw1)
    r2 = ops.gemm(dy, wt, M, K, N, tb=False, epi=epi, aux_in=a_in, variant=12, flags=fl, **kw2)
    e = ((r1.float() - r2.float()).norm() / r1.float().norm()).item()
    ec = ((c1 - c2).norm() / (c1.norm() + 1e-30)).item() if epi == ops.EPI_DGELU else 0.0
    print("NN vs NT %-5s rel %.2e  colsum rel %.2e" % (nm, e, ec)); assert e < 2e-3 and ec < 2e-3
print("ALL OK")
