# regime shapes (9984 rows): input gradients as planned now (W k-strided, "NT") against the k-contiguous form with W^T ("NN"), each
# with the kernel its own plan / tuner picks, residual-add and gelu'-multiply epilogues
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (9984, 3072, 38400):
    for (nout, kin, epi) in [(2304, 768, "add"), (768, 768, "none"), (3072, 768, "add"), (768, 3072, "mul")]:
        dy = torch.randn(M, nout, device=dev, dtype=torch.bfloat16)
        w = torch.randn(nout, kin, device=dev, dtype=torch.bfloat16) * 0.03
        wt = w.t().contiguous()
        aux = torch.randn(M, kin, device=dev, dtype=torch.bfloat16)
        code = {"add": ops.EPI_ADD, "none": ops.EPI_NONE, "mul": ops.EPI_DGELU}[epi]
        fl = ops.GEMM_AUX_DERIV if epi == "mul" else 0
        cs = torch.zeros(kin, device=dev) if epi == "mul" else None
        kw = dict(epi=code, aux_in=None if epi == "none" else aux, aux_out=cs, flags=fl)
        p_nt = ops.gemm_plan(torch.bfloat16, False, True, M, kin, nout)
        p_nn = ops.gemm_plan(torch.bfloat16, False, False, M, kin, nout)
        f_nt = lambda: ops._gemm_planned(dy, w, M, kin, nout, False, True, **kw)
        f_nn = lambda: ops._gemm_planned(dy, wt, M, kin, nout, False, False, **kw)
        t = {}
        for nm, fn in (("NT", f_nt), ("NN", f_nn)) * 2:
            t[nm] = min(t.get(nm, 1e9), timeit(fn))
        print("M=%5d dX[.,%4d] = dY[.,%4d] W %-4s: NT plan %s %.1f us   NN plan %s %.1f us   %+.1f %%" % (M, kin, nout, epi, p_nt, t["NT"], p_nn, t["NN"], (t["NT"] / t["NN"] - 1) * 100), flush=True)
