"""does the 128-byte-per-row access pattern limit the attention kernels?  12 heads in [B, L, 3H] rows of 4608 B (the model's layout)
against the same 12 288 heads as one-head "models" ([B * 12, L, 192]: a head's q|k|v rows are 384 B apart)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit
B, L, nh, D = 1024, 96, 12, 64
seed = torch.tensor([7], dtype=torch.int64, device="cuda")
for (b, h) in ((B, nh), (B * nh, 1)):
    H = h * D
    qkv = torch.randn(b * L, 3 * H, device="cuda").to(torch.bfloat16)
    mask = torch.zeros(b, L, device="cuda")
    dctx = torch.randn(b * L, H, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.1):
        ctx, lse = ops.attn_fwd(qkv, mask, b, L, h, D, p, seed, 3)
        tf = timeit(lambda: ops.attn_fwd(qkv, mask, b, L, h, D, p, seed, 3))
        tb = timeit(lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, b, L, h, D, p, seed, 3))
        nb = b * h * L * D * 2
        print("B=%d heads=%d p=%.1f: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (b, h, p, tf * 1e6, 4 * nb / tf / 1e12, tb * 1e6, 8 * nb / tb / 1e12), flush=True)
