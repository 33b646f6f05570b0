"""LayerNorm forward / backward at a uc2-large layer's size, with and without the e4m3 copy (fp8 mode), with and without dropout + residual"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import torch
from bench_gemm import timeit
from uc2_amd import ops
M, H = int(os.environ.get("M", "133120")), int(os.environ.get("H", "1024"))
bf = torch.bfloat16
x = torch.randn(M, H, device="cuda").to(bf)
res = torch.randn(M, H, device="cuda").to(bf)
dy = (torch.randn(M, H, device="cuda") * 0.1).to(bf)
g, b = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
seed = torch.tensor([5], dtype=torch.int64, device="cuda")
key = ("bench-ln-q", 0)
kb = ("bench-ln-q", 1)
ops.fp8_quantize_act(x, key)          # start the roles' histories
ops.fp8_quantize_act(dy, kb)
dg, db, dbi = [torch.zeros(H, device="cuda") for _ in range(3)]
for name, r, p, q in (("plain", None, 0.0, None), ("res+drop", res, 0.1, None), ("plain  +e4m3", None, 0.0, key), ("res+drop +e4m3", res, 0.1, key)):
    t = timeit(lambda: ops.ln_fwd(x, r, g, b, 1e-12, p, seed if p else None, 7, q_key=q), 20)
    out = ops.ln_fwd(x, r, g, b, 1e-12, p, seed if p else None, 7, q_key=q)
    nbytes = M * H * (2 + 2 + (2 if r is not None else 0) + (1 if q else 0))
    print("ln_fwd %-16s %7.1f us  %5.2f TB/s" % (name, t * 1e6, nbytes / t / 1e12), "(e4m3 written: %s)" % (out[3] is not None if q else "-"))
_, mean, rstd = ops.ln_fwd(x, None, g, b, 1e-12)
for name, r, p, da, q in (("fused form (drop_after 2)", None, 0.1, 2, None), ("res+drop", res, 0.1, False, None), ("fused form +e4m3", None, 0.1, 2, kb), ("res+drop +e4m3", res, 0.1, False, kb)):
    t = timeit(lambda: ops.ln_bwd(dy, x, r, g, mean, rstd, dg, db, p, seed, 7, dbias=dbi, drop_after=da, q_key=q), 20)
    nbytes = M * H * (2 + 2 + (2 if r is not None else 0) + 2 + 2 + (1 if q else 0))
    print("ln_bwd %-26s %7.1f us  %5.2f TB/s" % (name, t * 1e6, nbytes / t / 1e12))
ops.join_side_streams()
