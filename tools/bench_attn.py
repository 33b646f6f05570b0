"""attention kernel timings (not a test): forward/backward, dropout on/off"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
from bench_gemm import timeit

def main():
    B, L, nh, D = 1024, 96, 12, 64
    H = nh * D
    qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
    mask = torch.zeros(B, L, device="cuda")
    dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    for p in (0.0, 0.1):
        ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2)
        tf = timeit(lambda: ops.attn_fwd(qkv, mask, B, L, nh, D, p, seed, 3, impl=2))
        tb = timeit(lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2))
        db = torch.zeros(3 * H, device="cuda")
        tbb = timeit(lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2, dbias=db))
        dq = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, p, seed, 3, impl=2)
        tc = timeit(lambda: ops.colsum_accum(dq, db))
        print("dropout %.1f: fwd %.1f us, bwd %.1f us, bwd with the q|k|v bias gradient %.1f us (separate column-sum pass: %.1f us)"
              % (p, tf * 1e6, tb * 1e6, tbb * 1e6, tc * 1e6))

if __name__ == "__main__":
    main()
