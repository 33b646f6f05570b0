"""cycles the waves of one workgroup spend in the counted vmcnt waits / L-section barriers of the rolling kernel: far vs near stores"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
M = 98304
for (n, k, tb) in [(3072, 768, False), (3072, 768, True), (768, 768, False), (2304, 768, False)]:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = None if tb else torch.randn(n, device="cuda")
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    items = (M // 256) * (n // 256) // 256
    for tag, dg in (("far", 16), ("near", 16 | 8), ("nostore", 16 | 1), ("noepi", 16 | 2)):
        dbg = torch.zeros((8, 8), dtype=torch.int32, device="cuda")
        for _ in range(3):
            ops.gemm(a, b, M, n, k, tb=tb, bias=bv, out=out, aux_out=dbg, variant=10, flags=dg << 8)
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().astype("int64") & 0xffffffff
        print("N=%d K=%d tb=%d %-8s items/WG %d: per item (wave 0 / wave 4): plain waits %6d / %6d | store-window waits %6d / %6d | L barriers %6d / %6d"
              % (n, k, tb, tag, items, t[0, 0] // items, t[4, 0] // items, t[0, 1] // items, t[4, 1] // items, t[0, 2] // items, t[4, 2] // items), flush=True)
