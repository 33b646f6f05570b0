"""Micro-benchmark (not a test): the N = 768 GEMMs of the reference's 104-pair micro-batch (9 984 tokens) on every kernel that takes
them -- generic, the LDS-DMA rings, the ping-pong kernel with 256- / 192- / 128-row tiles (variants 8 / 9 / 5) and on the 16x16x32
MFMA (12).  One process, interleaved, median of 3 x 20 launches.   python tools/bench_regime_gemms.py [tokens]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                        # noqa: E402

from bench_gemm import timeit                                       # noqa: E402
from uc2_amd import ops                                             # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 9984
    bf = torch.bfloat16
    shapes = [("Wo forward (bias)", 768, 768, False, "none"), ("FFN2 forward (bias)", 768, 3072, False, "none"),
              ("Wo input gradient", 768, 768, True, "none"), ("QKV input gradient + residual", 768, 2304, True, "add"),
              ("FFN1 input gradient + residual", 768, 3072, True, "add")]
    for name, N, K, tb, epi in shapes:
        a = torch.randn(M, K, device="cuda").to(bf)
        b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.03).to(bf)
        bias = None if tb else torch.zeros(N, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(bf) if epi == "add" else None
        out = torch.empty(M, N, dtype=bf, device="cuda")
        code = ops.EPI_ADD if epi == "add" else ops.EPI_NONE
        res = {}
        for _ in range(3):
            for v in (99, 1, 2, 6, 7, 8, 9, 12, 5):
                if v == 9 and M % 192:
                    continue
                t = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, bias=bias, epi=code, aux_in=aux, out=out, variant=v), 20)
                res.setdefault(v, []).append(t)
        fl = 2.0 * M * N * K
        line = "  ".join("v%d %.1f us (%d TF/s)" % (v, sorted(ts)[1] * 1e6, fl / sorted(ts)[1] / 1e12) for v, ts in res.items())
        print("%-32s M=%d N=%d K=%d: %s" % (name, M, N, K, line), flush=True)


if __name__ == "__main__":
    main()
