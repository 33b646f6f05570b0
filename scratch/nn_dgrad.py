# input-gradient GEMMs dX = dY W: as they run now (W read k-strided, "NT") against the same product with a transposed copy of W
# (both operands k-contiguous, "NN"), variant 12, with the residual-add epilogue
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 98304
for (nout, kin, epi) in [(2304, 768, "add"), (768, 768, "none"), (3072, 768, "add"), (768, 3072, "none")]:
    # forward weight W [nout, kin]; dY [M, nout]; dX [M, kin]
    dy = torch.randn(M, nout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(nout, kin, device=dev, dtype=torch.bfloat16) * 0.03
    wt = w.t().contiguous()
    aux = torch.randn(M, kin, device=dev, dtype=torch.bfloat16)
    code = ops.EPI_ADD if epi == "add" else ops.EPI_NONE
    kw = dict(epi=code, aux_in=aux if epi == "add" else None)
    r_nt = ops.gemm(dy, w, M, kin, nout, tb=True, variant=12, **kw)
    r_nn = ops.gemm(dy, wt, M, kin, nout, tb=False, variant=12, **kw)
    assert torch.equal(r_nt, r_nn) or ((r_nt.float() - r_nn.float()).abs().max() < 0.1)
    t = {}
    for nm, fn in (("NT", lambda: ops.gemm(dy, w, M, kin, nout, tb=True, variant=12, **kw)), ("NN", lambda: ops.gemm(dy, wt, M, kin, nout, tb=False, variant=12, **kw))) * 2:
        t[nm] = min(t.get(nm, 1e9), timeit(fn))
    fl = 2.0 * M * kin * nout
    print("dX[%d,%d] = dY[.,%d] W  epi %-4s: NT %.1f us (%.0f TF/s)   NN %.1f us (%.0f TF/s)   %+.1f %%" % (M, kin, nout, epi, t["NT"], fl / t["NT"] / 1e6, t["NN"], fl / t["NN"] / 1e6, (t["NT"] / t["NN"] - 1) * 100))
