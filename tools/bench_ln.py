"""LayerNorm kernel timings (not a test): dropout on/off, with/without the fused bias-gradient sums"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
from bench_gemm import timeit

def main():
    M, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 98304), 768
    x = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    dy = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
    seed = torch.tensor([7], dtype=torch.int64, device="cuda")
    dg = torch.zeros(H, device="cuda"); dbt = torch.zeros(H, device="cuda"); dbias = torch.zeros(H, device="cuda")
    for p in (0.0, 0.1):
        y, mean, rstd = ops.ln_fwd(x, r, g, b, 1e-12, p, seed, 3)
        tf = timeit(lambda: ops.ln_fwd(x, r, g, b, 1e-12, p, seed, 3))
        tb = timeit(lambda: ops.ln_bwd(dy, x, r, g, mean, rstd, dg, dbt, p, seed, 3, dbias=dbias))
        tb2 = timeit(lambda: ops.ln_bwd(dy, x, r, g, mean, rstd, dg, dbt, p, seed, 3))
        nb = 5 if p > 0 else 4          # without dropout dres == dx is one stream
        print("p=%.1f: ln_fwd %.1f us (%.2f TB/s)  ln_bwd %.1f us (%.2f TB/s)  ln_bwd no dbias %.1f us" %
              (p, tf * 1e6, 3 * M * H * 2 / tf / 1e12, tb * 1e6, nb * M * H * 2 / tb / 1e12, tb2 * 1e6))

if __name__ == "__main__":
    main()
