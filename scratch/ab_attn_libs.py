"""A/B of several builds of libuc2_hip.so in ONE process: attention forward / backward at the bench size, interleaved rounds.
usage: python scratch/ab_attn_libs.py <lib1.so> <lib2.so> ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from uc2_amd import ops, _lib
from ab_gemm_libs import load_lib, timeit


def main():
    paths = sys.argv[1:]
    libs = [load_lib(os.path.abspath(p)) for p in paths]
    B, L, nh, D = int(os.environ.get("AB_PAIRS", "1024")), 96, 12, 64
    H = nh * D
    qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
    mask = torch.zeros(B, L, device="cuda")
    dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    db = torch.zeros(3 * H, device="cuda")
    p = 0.1
    _lib._lib = libs[0]
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2)
    ref = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2)
    tf = [[] for _ in libs]; tb = [[] for _ in libs]
    for r in range(5):
        for i, lib in enumerate(libs):
            _lib._lib = lib
            c2, _ = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2)
            g2 = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2)
            if os.environ.get("AB_NOCHECK") != "1":        # (builds that read the buffers in another layout: timing only)
                assert torch.equal(c2, ctx) and torch.equal(g2, ref), paths[i]
            tf[i].append(timeit(lambda: ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2)))
            tb[i].append(timeit(lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2, dbias=db)))
    fb = B * L * H * 2 * 4 + B * nh * L * 4
    bb = B * L * H * 2 * 8 + B * nh * L * 4
    for i, pth in enumerate(paths):
        f, b = statistics.median(tf[i]), statistics.median(tb[i])
        print("%-28s fwd %6.1f us (%.2f TB/s, min %.1f)   bwd+dbias %6.1f us (%.2f TB/s, min %.1f)"
              % (os.path.basename(pth), f * 1e6, fb / f / 1e12, min(tf[i]) * 1e6, b * 1e6, bb / b / 1e12, min(tb[i]) * 1e6))


if __name__ == "__main__":
    main()
