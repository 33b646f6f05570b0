"""PROBE (not a test): do the HBM-bound kernels of the step run beside a persistent GEMM that leaves some CUs free?

The step is 81 % GEMM (power-limited: the same PF/s on 232 as on 256 CUs, profiles/HISTORY.md) and 17 % HBM-bound kernels
(LayerNorm, attention) that run one after the other with it.  If a GEMM launched on 256 - 8 n CUs (UC2_GEMM_SPARE(n)) keeps its
speed while a LayerNorm / attention kernel of ANOTHER half of the batch streams on the free CUs, a two-pipeline step could hide
part of the 52 ms.  This probe measures exactly that, nothing else: stream A runs GEMMs back to back, stream B the HBM-bound
kernel back to back, alone and together.

    python tools/coresident_probe.py [tokens]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                        # noqa: E402

from uc2_amd import ops                                             # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
    H, I_ = 768, 3072
    dev = "cuda"
    bf = torch.bfloat16
    x = torch.randn(M, H, device=dev).to(bf)
    u = torch.randn(M, I_, device=dev).to(bf)
    w1 = (torch.randn(I_, H, device=dev) * 0.03).to(bf)
    w2 = (torch.randn(H, I_, device=dev) * 0.03).to(bf)
    b1 = torch.zeros(I_, device=dev)
    b2 = torch.zeros(H, device=dev)
    out1 = torch.empty(M, I_, dtype=bf, device=dev)
    aux1 = torch.empty(M, I_, dtype=bf, device=dev)
    out2 = torch.empty(M, H, dtype=bf, device=dev)
    g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    xs = [torch.randn(M, H, device=dev).to(bf) for _ in range(4)]          # LayerNorm inputs (rotating: no cache residency)
    B, L, nh, D = M // 96, 96, 12, 64
    qkv = torch.randn(M, 3 * H, device=dev).to(bf)
    mask = torch.zeros(B, L, device=dev)

    def gemm_ffn1(spare):
        ops.gemm(x, w1, M, I_, H, bias=b1, epi=ops.EPI_GELU, aux_out=aux1, out=out1, variant=12, flags=ops.GEMM_AUX_DERIV | ((spare & 7) << 28))

    def gemm_ffn2(spare):
        ops.gemm(u, w2, M, H, I_, bias=b2, out=out2, variant=12, flags=(spare & 7) << 28)

    cnt = [0]

    def ln():
        cnt[0] += 1
        ops.ln_fwd(xs[cnt[0] & 3], None, g, b, 1e-12, want_stats=True)

    def attn():
        ops.attn_fwd(qkv, mask, B, L, nh, D)

    A, Bs = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(stream, fn, n):
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
        return e0, e1

    def alone(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = timed(A, fn, n)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3                      # us per launch

    for gname, gfn in (("FFN1+GELU+gelu' %dx3072x768" % M, gemm_ffn1), ("FFN2 %dx768x3072" % M, gemm_ffn2)):
        for hname, hfn in (("ln_fwd [%d,768]" % M, ln), ("attn_fwd %d heads" % (B * nh), attn)):
            t_h = alone(hfn)
            print("%s | %s alone %.1f us" % (gname, hname, t_h), flush=True)
            for spare in (0, 2, 4, 6):
                t_g = alone(lambda: gfn(spare))
                # together: n_g GEMMs on A; on B enough launches of the HBM kernel to last about as long (it runs on ~8*spare CUs)
                n_g = 24
                slow = 256.0 / max(8 * spare, 8)
                n_h = max(2, int(n_g * t_g / (t_h * slow)))
                torch.cuda.synchronize()
                cur = torch.cuda.current_stream()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                A.wait_stream(cur)
                Bs.wait_stream(cur)
                with torch.cuda.stream(A):
                    gfn(spare)                                     # resident first
                ea = timed(A, lambda: gfn(spare), n_g)
                eb = timed(Bs, hfn, n_h)
                cur.wait_stream(A)
                cur.wait_stream(Bs)
                ev1.record()
                torch.cuda.synchronize()
                ta, tb = ea[0].elapsed_time(ea[1]) / n_g * 1e3, eb[0].elapsed_time(eb[1]) / n_h * 1e3
                wall = ev0.elapsed_time(ev1) * 1e3
                serial = (n_g + 1) * t_g + n_h * t_h
                print("   spare %2d CUs: GEMM alone %.1f us, beside %.1f us (x%.3f) | HBM kernel beside the GEMM %.1f us (x%.2f of alone) | "
                      "%d GEMMs + %d HBM launches: together %.0f us, one after the other %.0f us (x%.3f)"
                      % (8 * spare, t_g, ta, ta / t_g, tb, tb / t_h, n_g + 1, n_h, wall, serial, wall / serial), flush=True)

if __name__ == "__main__":
    main()
