"""What fusing dropout + residual into the Wo / FFN2 GEMM epilogue would buy, measured with kernels that exist:
the GEMMs with and without the residual-add epilogue, LayerNorm forward with 3 / 2 streams, backward with 5 / 4.
usage: python scratch/ln_fusion_estimate.py [pairs]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
dev = "cuda"


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def med(fn):
    return statistics.median([timeit(fn) for _ in range(5)]) * 1e6


pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M, H, I = pairs * 96, 768, 3072
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
res = torch.randn(M, H, device=dev).to(torch.bfloat16)
u = torch.randn(M, I, device=dev).to(torch.bfloat16)
wo = (torch.randn(H, H, device=dev) * 0.03).to(torch.bfloat16)
w2 = (torch.randn(H, I, device=dev) * 0.03).to(torch.bfloat16)
bias = torch.randn(H, device=dev)
out = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
g_plain_o = med(lambda: ops.gemm(x, wo, M, H, H, out=out, bias=bias, variant=12))
g_add_o = med(lambda: ops.gemm(x, wo, M, H, H, out=out, bias=bias, epi=ops.EPI_ADD, aux_in=res, variant=12))
g_plain_2 = med(lambda: ops.gemm(u, w2, M, H, I, out=out, bias=bias, variant=12))
g_add_2 = med(lambda: ops.gemm(u, w2, M, H, I, out=out, bias=bias, epi=ops.EPI_ADD, aux_in=res, variant=12))
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
seed = torch.tensor([1], dtype=torch.int64, device=dev)
f3 = med(lambda: ops.ln_fwd(x, res, gamma, beta, 1e-12, 0.1, seed, 3))
f2 = med(lambda: ops.ln_fwd(x, None, gamma, beta, 1e-12, 0.0, None, 0))
y, mean, rstd = ops.ln_fwd(x, res, gamma, beta, 1e-12, 0.1, seed, 3)
dy = torch.randn(M, H, device=dev).to(torch.bfloat16)
dg, db, dbias = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)


def b5():
    ops.ln_bwd(dy, x, res, gamma, mean, rstd, dg, db, 0.1, seed, 3, dbias=dbias)


def b4():      # one input stream less (x is the pre-LayerNorm sum), still two outputs (masked dx + dres): dropout on, no residual read
    ops.ln_bwd(dy, x, None, gamma, mean, rstd, dg, db, 0.1, seed, 3, dbias=dbias)


t5, t4 = med(b5), med(b4)
ops.flush_ln_reductions(); ops.join_side_streams(); torch.cuda.synchronize()
print("%d pairs (%d tokens), us per launch" % (pairs, M))
print("Wo GEMM   %dx768x768   bias only %7.1f   + residual tile in the epilogue %7.1f   (+%.1f)" % (M, g_plain_o, g_add_o, g_add_o - g_plain_o))
print("FFN2 GEMM %dx768x3072  bias only %7.1f   + residual tile in the epilogue %7.1f   (+%.1f)" % (M, g_plain_2, g_add_2, g_add_2 - g_plain_2))
print("LayerNorm forward   3 streams (x, residual -> y, dropout) %7.1f   2 streams (sum -> y) %7.1f   (-%.1f)" % (f3, f2, f3 - f2))
print("LayerNorm backward  5 streams %7.1f   4 streams (no residual read) %7.1f   (-%.1f)" % (t5, t4, t5 - t4))
per_layer = (g_add_o - g_plain_o) + (g_add_2 - g_plain_2) - 2 * (f3 - f2) - 2 * (t5 - t4)
print("per layer: %+.1f us BEFORE the dropout hash the epilogue would also have to evaluate; x 12 layers = %+.2f ms per step of %d pairs" % (per_layer, per_layer * 12e-3, pairs))
