"""Variant 13 (one wave per SIMD, 128 x 128 per wave, gemm_p1.hip) against variant 12 (ping-pong, 16x16x32): bit comparison on
small and bench shapes, then interleaved timing of the k-contiguous GEMMs of the step (whole kernel and main loop only).
usage: python scratch/p1_probe.py [pairs ...]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops

dev = "cuda"
torch.manual_seed(0)
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


def run(v, a, b, m, n, k, bias, epi, aux_in, aux_out, out, fl):
    return ops.gemm(a, b, m, n, k, out=out, bias=bias, epi=epi, aux_in=aux_in, aux_out=aux_out, variant=v, flags=fl)


ok = True
for (M, N, K) in [(256, 256, 128), (512, 768, 768), (1024, 256, 256), (4096, 2304, 768), (2048, 768, 3072), (9984 // 256 * 256, 3072, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for epi, fl, nm in ((E.EPI_NONE, 0, "plain"), (E.EPI_ADD, 0, "add"), (E.EPI_DGELU, E.GEMM_AUX_DERIV, "mul")):
        outs = []
        for v in (12, 13, 13):
            o = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            cs = torch.zeros(N, device=dev) if epi == E.EPI_DGELU else None
            run(v, a, b, M, N, K, bias if epi == E.EPI_NONE else None, epi, aux if epi != E.EPI_NONE else None, cs, o, fl)
            outs.append((o, cs))
        same = torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[1][0].view(torch.int16), outs[2][0].view(torch.int16))
        csd = 0.0 if outs[0][1] is None else ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()).item()
        ref = a.float() @ b.float().t()
        if epi == E.EPI_NONE: ref = ref + bias
        elif epi == E.EPI_ADD: ref = ref + aux.float()
        else: ref = ref * aux.float()
        err = ((outs[1][0].float() - ref).norm() / ref.norm()).item()
        good = same and err < 4e-3 and csd < 1e-3
        ok &= good
        print("%-6s %6dx%5dx%5d  bit-identical to v12: %s  rel err vs fp32 %.2e  colsum diff %.1e%s" % (nm, M, N, K, same, err, csd, "" if good else "   <-- FAIL"), flush=True)
print("PARITY", "OK" if ok else "FAILED", flush=True)

H, I = 768, 3072
for pairs in [int(x) for x in sys.argv[1:]] or [1024]:
    M = pairs * 96
    shapes = [("fwd qkv", M, 3 * H, H, E.EPI_NONE, 0), ("fwd out", M, H, H, E.EPI_NONE, 0), ("fwd ffn2", M, H, I, E.EPI_NONE, 0),
              ("plain 3072x768", M, I, H, E.EPI_NONE, 0),
              ("dgrad ffn2 mul", M, I, H, E.EPI_DGELU, E.GEMM_AUX_DERIV), ("dgrad ffn1 add", M, H, I, E.EPI_ADD, 0), ("dgrad qkv add", M, H, 3 * H, E.EPI_ADD, 0)]
    for name, m, n, k, epi, fl in shapes:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = (torch.randn(n, k, device=dev) * 0.03).to(torch.bfloat16)
        out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
        bias = torch.randn(n, device=dev) if epi == E.EPI_NONE else None
        aux_in = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi != E.EPI_NONE else None
        aux_out = torch.zeros(n, device=dev) if epi == E.EPI_DGELU else None
        res = {}
        for diag, tag in ((0, "kernel"), (0x8, "main loop")):
            ts = {12: [], 13: []}
            for r in range(5):
                for v in (12, 13):
                    ts[v].append(timeit(lambda: run(v, a, b, m, n, k, bias, epi, aux_in, aux_out, out, fl | (diag << 8))))
            res[tag] = {v: statistics.median(t) for v, t in ts.items()}
        f = 2.0 * m * n * k
        print("%-16s %7dx%5dx%5d | kernel v12 %8.1f us %5.0f TF  v13 %8.1f us %5.0f TF | main loop v12 %8.1f us %5.0f TF  v13 %8.1f us %5.0f TF" % (
            name, m, n, k, res["kernel"][12] * 1e6, f / res["kernel"][12] / 1e12, res["kernel"][13] * 1e6, f / res["kernel"][13] / 1e12,
            res["main loop"][12] * 1e6, f / res["main loop"][12] / 1e12, res["main loop"][13] * 1e6, f / res["main loop"][13] / 1e12), flush=True)
        del a, b, out, aux_in
