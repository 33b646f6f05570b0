"""A/B of several builds of libuc2_hip.so in ONE process: LayerNorm backward (the fused-tail form the bench step runs: drop_after = 2,
dy + the pre-LayerNorm sum in, dx + dres out, partial column sums) and forward.   python scratch/ab_ln_libs.py <tokens> <lib1.so> <lib2.so> ..."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_gemm_libs import load_lib, timeit


def main():
    M = int(sys.argv[1])
    libs = [load_lib(os.path.abspath(p)) for p in sys.argv[2:]]
    H = 768
    bf = torch.bfloat16
    xs = [torch.randn(M, H, device="cuda").to(bf) for _ in range(3)]
    dys = [torch.randn(M, H, device="cuda").to(bf) for _ in range(3)]
    g, b = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
    seed = torch.tensor([7], dtype=torch.int64, device="cuda")
    dg, dbt, dbias = (torch.zeros(H, device="cuda") for _ in range(3))
    _lib._lib = libs[0]
    _, mean, rstd = ops.ln_fwd(xs[0], None, g, b, 1e-12)
    cnt = [0]

    def bwd():
        cnt[0] += 1
        i = cnt[0] % 3
        ops.ln_bwd(dys[i], xs[i], None, g, mean, rstd, dg, dbt, 0.1, seed, 3, dbias=dbias, drop_after=2)

    def fwd():
        cnt[0] += 1
        ops.ln_fwd(xs[cnt[0] % 3], None, g, b, 1e-12)
    outs = []
    for lib in libs:                                     # same results from every build
        _lib._lib = lib
        dx, dres = ops.ln_bwd(dys[0], xs[0], None, g, mean, rstd, dg, dbt, 0.1, seed, 3, dbias=dbias, drop_after=2)
        torch.cuda.synchronize()
        outs.append((dx.clone(), dres.clone()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    for name, fn, nbytes in (("ln_bwd (drop_after 2)", bwd, 4 * M * H * 2), ("ln_fwd", fwd, 2 * M * H * 2)):
        ts = [[] for _ in libs]
        for r in range(5):
            for i, lib in enumerate(libs):
                _lib._lib = lib
                ts[i].append(timeit(fn, 20))
        print("%-22s %7d tokens: " % (name, M) + "   ".join("%s %.1f us (%.2f TB/s, min %.1f)" % (os.path.basename(sys.argv[2 + i]), statistics.median(t) * 1e6, nbytes / statistics.median(t) / 1e12, min(t) * 1e6) for i, t in enumerate(ts)), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
