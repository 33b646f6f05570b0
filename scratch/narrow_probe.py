"""TIMING PROBE (diagnostic build, UC2_LIB_PATH=uc2_amd/libuc2_hip_diag.so): the ping-pong kernel with the A x B1 half of every k-tile
skipped (flag 0x400: wrong results) = what a 256 x 128 tile would cost per workgroup with the 4-phase schedule kept.
python scratch/narrow_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
dev = "cuda"
M = 9984
def timeit(fn, n=30, rounds=5):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
for (N, K, tb, name) in ((768, 3072, False, "FFN2 fwd"), (768, 3072, True, "FFN1 dgrad"), (768, 2304, True, "QKV dgrad"), (768, 768, False, "Wo fwd")):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)).bfloat16() * 0.03
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    plan = ops.gemm_plan(torch.bfloat16, False, tb, M, N, K, False)
    t_plan = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, variant=plan[0], split_k=plan[1]))
    t_pp = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, variant=12))
    t_half = timeit(lambda: ops.gemm(a, b, M, N, K, tb=tb, out=out, variant=12, flags=0x400))
    print("%-12s N %4d K %4d: plan %s %.1f us | ping-pong 256x256 (117 tiles) %.1f us | A x B1 half skipped (= 234 tiles of 256x128) %.1f us" % (name, N, K, plan, t_plan, t_pp, t_half))
