"""Micro-benchmark (not a test): the MLM head's input-gradient GEMM dz = dlogits E (M = rows of a decoder chunk, N = 768, K = 250 112,
fp32 partial tiles + one reduction pass) over split-K factors.  The tuner's candidates aim at 0.5 / 1 / 2 items per CU; with 90 tiles
per 7 680-row chunk that is 1, 3 or 6 splits = 90 / 270 / 540 items on 256 CUs = 35 % / 53 % / 70 % of whole rounds, while the
contraction (3 908 k-tiles) would still be 230 k-tiles per item at 17 splits = 1 530 items = 99.6 %.
python tools/bench_decoder_dz.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                        # noqa: E402

from bench_gemm import timeit                                       # noqa: E402
from uc2_amd import ops                                             # noqa: E402
from uc2_amd.ops.gemm import _plan_fits                            # noqa: E402


def main():
    for M in ([int(x) for x in sys.argv[1:]] or [7680]):
        one(M)


def one(M):
    N, K = 768, 250112
    bf = torch.bfloat16
    a = (torch.randn(M, K, device="cuda") * 0.01).to(bf)
    res = {}
    for tb, name in ((False, "E^T [768, Vp] (NN)"), (True, "E [Vp, 768] (NT)")):
        b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.03).to(bf)
        out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        key = (False, tb, M, N, K, True)
        tiles = (M // 256) * (N // 256)
        splits = [s for s in range(1, 40) if _plan_fits((12, s), key)]
        for _ in range(3):
            for s in splits:
                if s not in (1, 3, 5, 6, 8, 11, 14, 17, 20, 23, 26, 31, 34) and (tiles * s) % 256 > 40 and (tiles * s) % 256 < 216:
                    continue
                t = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, accumulate=True, split_k=s, variant=12), 5)
                res.setdefault((name, s), []).append(t)
        fl = 2.0 * M * N * K
        for (nm, s), ts in res.items():
            if nm == name:
                print("%-20s M=%d split %2d: %4d items = %5.2f rounds  %8.1f us  %6.0f TF/s" % (nm, M, s, tiles * s, tiles * s / 256.0, sorted(ts)[1] * 1e6, fl / sorted(ts)[1] / 1e12), flush=True)
        best = min(((sorted(ts)[1], s) for (nm, s), ts in res.items() if nm == name))
        print("BEST %s M=%d: split %d (%.1f us)" % ("NT" if tb else "NN", M, best[1], best[0] * 1e6), flush=True)


if __name__ == "__main__":
    main()
