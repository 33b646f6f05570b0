"""A/B (not a test): the two epilogue-heavy K = 768 GEMMs of the step -- FFN1 + GELU + gelu' stream, and the gelu'-multiply input
gradient -- on the persistent ping-pong kernel (variant 12: 256 x 256 tile, 8 waves = 2 per SIMD of 128 x 64, ONE workgroup per CU)
against the LDS-DMA ring kernels that put TWO co-resident workgroups of 256 x 128 on a CU (variant 2: 8 waves of 64 x 64 per
workgroup = 4 waves per SIMD, 72 KiB of LDS each; variant 0: 128 x 128), i.e. the structure in which one workgroup's epilogue runs
beside another's main loop (VERDICT r5 #1a).  Same process, same box, interleaved.   python tools/epi_gemm_ab.py [tokens]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                        # noqa: E402

from bench_gemm import timeit                                       # noqa: E402
from uc2_amd import ops                                             # noqa: E402


def main():
    for M in ([int(a) for a in sys.argv[1:]] or [98304, 589824]):
        H, I_ = 768, 3072
        bf = torch.bfloat16
        x = torch.randn(M, H, device="cuda").to(bf)
        dy = torch.randn(M, H, device="cuda").to(bf)
        w1 = (torch.randn(I_, H, device="cuda") * 0.03).to(bf)
        w2t = (torch.randn(I_, H, device="cuda") * 0.03).to(bf)          # W2^T [3072, 768]: d_pre = dy W2 as an X W^T product
        b1 = torch.zeros(I_, device="cuda")
        out = torch.empty(M, I_, dtype=bf, device="cuda")
        aux = torch.empty(M, I_, dtype=bf, device="cuda")
        gp = torch.rand(M, I_, device="cuda").to(bf)
        cs = torch.zeros(I_, device="cuda")
        fl = 2.0 * M * I_ * H
        rows = []
        for rnd in range(3):
            for v in (12, 2, 0, 1):
                t1 = timeit(lambda: ops.gemm(x, w1, M, I_, H, bias=b1, epi=ops.EPI_GELU, aux_out=aux, out=out, variant=v, flags=ops.GEMM_AUX_DERIV), 10)
                t2 = timeit(lambda: ops.gemm(dy, w2t, M, I_, H, epi=ops.EPI_DGELU, aux_in=gp, aux_out=cs, out=out, variant=v, flags=ops.GEMM_AUX_DERIV), 10)
                t3 = timeit(lambda: ops.gemm(x, w1, M, I_, H, bias=b1, out=out, variant=v), 10)
                rows.append((v, t1, t2, t3))
        for v in (12, 2, 0, 1):
            r = [q for q in rows if q[0] == v]
            med = lambda i: sorted(q[i] for q in r)[len(r) // 2]
            print("tokens %7d variant %2d: FFN1+GELU+gelu' %7.1f us (%4.0f TF/s) | x gelu' dgrad %7.1f us (%4.0f TF/s) | bias only %7.1f us (%4.0f TF/s)"
                  % (M, v, med(1) * 1e6, fl / med(1) / 1e12, med(2) * 1e6, fl / med(2) / 1e12, med(3) * 1e6, fl / med(3) / 1e12), flush=True)
        del x, dy, out, aux, gp


if __name__ == "__main__":
    main()
