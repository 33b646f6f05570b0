"""time the headline-size forward GEMMs (no aux tile) of the loaded library build: python scratch/epi_ab.py  (UC2_LIB_PATH selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
dev = "cuda"
M = 589824
def timeit(fn, n=6, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
x = torch.randn(M, 768, device=dev).bfloat16()
out = []
for (N, epi, name) in ((2304, ops.EPI_NONE, "QKV fwd"), (3072, ops.EPI_GELU, "FFN1 GELU+gelu'"), (768, ops.EPI_NONE, "Wo fwd")):
    w = (torch.randn(N, 768, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if epi == ops.EPI_GELU else None
    t = timeit(lambda: ops.gemm(x, w, M, N, 768, out=o, bias=bias, epi=epi, aux_out=pre, variant=12, flags=ops.GEMM_AUX_DERIV if pre is not None else 0))
    out.append("%s %.1f us" % (name, t))
    del w, o, pre
print(os.environ.get("UC2_LIB_PATH", "current")[-20:], " | ".join(out))
