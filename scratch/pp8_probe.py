"""fp8 ping-pong kernel (gemm_pp8.hip) against the fp8 ring kernel and the bf16 ping-pong kernel (variant 12) on the uc2-large
shapes (256 pairs x 130 = 33 280 tokens; H 1024, I 4096).  usage: python scratch/pp8_probe.py [tokens]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
dev = "cuda"
E = ops


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


T = int(sys.argv[1]) if len(sys.argv) > 1 else 33280
H, I = 1024, 4096
for name, m, n, k, epi, fl in [("fwd qkv", T, 3 * H, H, E.EPI_NONE, 0), ("fwd out", T, H, H, E.EPI_NONE, 0), ("fwd ffn1 gelu'", T, I, H, E.EPI_GELU, E.GEMM_AUX_DERIV),
                               ("fwd ffn2", T, H, I, E.EPI_NONE, 0), ("dgrad ffn2 mul", T, I, H, E.EPI_DGELU, E.GEMM_AUX_DERIV), ("dgrad ffn1 add", T, H, I, E.EPI_ADD, 0),
                               ("dgrad qkv add", T, H, 3 * H, E.EPI_ADD, 0)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = (torch.randn(n, k, device=dev) * 0.03).to(torch.bfloat16)
    a8, sa = ops.fp8_quantize(a)
    b8, sb = ops.fp8_quantize(b)
    out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(n, device=dev) if epi in (E.EPI_NONE, E.EPI_GELU) else None
    aux_in = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi in (E.EPI_DGELU, E.EPI_ADD) else None
    aux_out = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == E.EPI_GELU else (torch.zeros(n, device=dev) if epi == E.EPI_DGELU else None)
    fns = {"bf16 v12": lambda: ops.gemm(a, b, m, n, k, out=out, bias=bias, epi=epi, aux_in=aux_in, aux_out=aux_out, variant=12, flags=fl),
           "fp8 ring": lambda: ops.gemm_fp8(a8, sa, b8, sb, bias=bias, epi=epi, aux_in=aux_in, aux_out=aux_out, flags=fl | 4),
           "fp8 ping-pong": lambda: ops.gemm_fp8(a8, sa, b8, sb, bias=bias, epi=epi, aux_in=aux_in, aux_out=aux_out, flags=fl)}
    ts = {k_: [] for k_ in fns}
    for r in range(5):
        for k_, f in fns.items():
            ts[k_].append(timeit(f))
    fl_ = 2.0 * m * n * k
    print("%-16s %6dx%5dx%5d " % (name, m, n, k) + " | ".join("%s %7.1f us %5.0f TF" % (k_, statistics.median(t) * 1e6, fl_ / statistics.median(t) / 1e12) for k_, t in ts.items()), flush=True)
