"""attention forward / backward only, for rocprofv3 --pmc: python scratch/attn_only.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
B, L, nh, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 96, 12, 64
H = nh * D
qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
mask = torch.zeros(B, L, device="cuda")
dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
db = torch.zeros(3 * H, device="cuda")
for _ in range(6):
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 3, impl=2)
    ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, 0.1, seed, 3, impl=2, dbias=db)
torch.cuda.synchronize()
