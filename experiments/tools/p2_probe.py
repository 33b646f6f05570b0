"""Variant 14 (gemm_p2.hip: one wave per SIMD, epilogue in the next item's MFMA gaps) against variant 12 and an fp32 product,
then interleaved timing of the K = 768 forward GEMMs.  usage: python scratch/p2_probe.py [pairs ...]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops

dev = "cuda"
torch.manual_seed(0)
E = ops
V = int(os.environ.get("P2_VARIANT", "14"))


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def ulps(a, b):
    """elements that differ, and the largest difference in units of 2^-8 of the larger magnitude (one bf16 ulp is 1..2 of them); values
    below 1e-2 are compared absolutely (the two kernels add the bias at different ends of the fp32 sum: near zero the SUMS differ by
    fp32 rounding, which is a large relative difference of a tiny number)"""
    af, bf = a.float(), b.float()
    d = (af - bf).abs()
    mag = torch.maximum(af.abs(), bf.abs()).clamp_min(1e-2)
    return int((d != 0).sum()), float((d / mag).max() * 256.0)


ok = True
K = 768
for (M, N) in [(256, 256), (512, 256), (256, 768), (4096, 3072), (9984 // 256 * 256, 3072), (66 * 256, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    for with_bias in (True, False):
        bias = torch.randn(N, device=dev) if with_bias else None
        for epi, fl, nm in ((E.EPI_NONE, 0, "plain"), (E.EPI_GELU, E.GEMM_AUX_DERIV, "gelu+gelu'")):
            res = []
            for v in (12, V, V):
                o = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
                x = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) if epi == E.EPI_GELU else None
                ops.gemm(a, b, M, N, K, out=o, bias=bias, epi=epi, aux_out=x, variant=v, flags=fl)
                res.append((o, x))
            torch.cuda.synchronize()
            h = a.float() @ b.float().t()
            if with_bias:
                h = h + bias
            if epi == E.EPI_GELU:
                ref = torch.nn.functional.gelu(h)
                hh = h.double()
                refd = (0.5 * (1 + torch.erf(hh / 2 ** 0.5)) + hh * torch.exp(-0.5 * hh * hh) / (2 * 3.141592653589793) ** 0.5).float()
            else:
                ref, refd = h, None
            rep = torch.equal(res[1][0].view(torch.int16), res[2][0].view(torch.int16))
            nd, mu = ulps(res[1][0], res[0][0])
            err = ((res[1][0].float() - ref).norm() / ref.norm()).item()
            e12 = ((res[0][0].float() - ref).norm() / ref.norm()).item()
            finite = bool(torch.isfinite(res[1][0].float()).all())
            line = "%-11s bias=%d %6dx%5d  repeatable %s  vs v12: %d of %d differ, max %.2f x 2^-8  rel err vs fp32 %.3e (v12 %.3e)" % (
                nm, with_bias, M, N, rep, nd, M * N, mu, err, e12)
            good = rep and finite and err < 4e-3 and mu <= 2.01 and nd < 0.02 * M * N
            if refd is not None:
                nd2, mu2 = ulps(res[1][1], res[0][1])
                errd = ((res[1][1].float() - refd).norm() / refd.norm()).item()
                rep2 = torch.equal(res[1][1].view(torch.int16), res[2][1].view(torch.int16))
                line += " | gelu': %d differ, max %.2f x 2^-8, rel err %.3e, repeatable %s" % (nd2, mu2, errd, rep2)
                good = good and rep2 and errd < 4e-3 and bool(torch.isfinite(res[1][1].float()).all())
            ok &= good
            print(line + ("" if good else "   <-- FAIL"), flush=True)
print("PARITY", "OK" if ok else "FAILED", flush=True)

H, I = 768, 3072
for pairs in [int(x) for x in sys.argv[1:]] or [1024]:
    M = pairs * 96
    for name, m, n, k, epi, fl in [("fwd ffn1 gelu'", M, I, H, E.EPI_GELU, E.GEMM_AUX_DERIV), ("plain 3072x768", M, I, H, E.EPI_NONE, 0),
                                   ("fwd qkv", M, 3 * H, H, E.EPI_NONE, 0), ("fwd out", M, H, H, E.EPI_NONE, 0)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = (torch.randn(n, k, device=dev) * 0.03).to(torch.bfloat16)
        out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
        bias = torch.randn(n, device=dev)
        aux_out = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == E.EPI_GELU else None
        res = {}
        for diag, tag in ((0, "kernel"), (0x8, "main loop")):
            ts = {12: [], V: []}
            for r in range(5):
                for v in (12, V):
                    ts[v].append(timeit(lambda: ops.gemm(a, b, m, n, k, out=out, bias=bias, epi=epi, aux_out=aux_out, variant=v, flags=fl | (diag << 8))))
            res[tag] = {v: statistics.median(t) for v, t in ts.items()}
        f = 2.0 * m * n * k
        print("%-16s %7dx%5dx%5d | kernel v12 %8.1f us %5.0f TF  v%d %8.1f us %5.0f TF | main loop v12 %8.1f us %5.0f TF  v%d %8.1f us %5.0f TF" % (
            name, m, n, k, res["kernel"][12] * 1e6, f / res["kernel"][12] / 1e12, V, res["kernel"][V] * 1e6, f / res["kernel"][V] / 1e12,
            res["main loop"][12] * 1e6, f / res["main loop"][12] / 1e12, V, res["main loop"][V] * 1e6, f / res["main loop"][V] / 1e12), flush=True)
        if V == 14:         # what the one-wave stream is made of (no stores; stale LDS contents): - LDS-DMA, - fragment reads, - both
            line = "   v14 without stores"
            for diag, tag in ((0x8, "all"), (0x9, "no LDS-DMA"), (0xA, "no fragment reads"), (0xB, "MFMAs + epilogue steps + barriers only")):
                tt = statistics.median([timeit(lambda: ops.gemm(a, b, m, n, k, out=out, bias=bias, epi=epi, aux_out=aux_out, variant=V, flags=fl | (diag << 8))) for _ in range(3)])
                line += " | %s %.1f us (%.0f TF)" % (tag, tt * 1e6, f / tt / 1e12)
            print(line, flush=True)
        del a, b, out, aux_out
