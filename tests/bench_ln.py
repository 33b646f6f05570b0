"""LayerNorm kernel timings (not a test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
from bench_gemm import timeit

def main():
    M, H = 49152, 768
    x = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    dy = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
    seed = torch.tensor([7], dtype=torch.int64, device="cuda")
    y, mean, rstd = ops.ln_fwd(x, r, g, b, 1e-12, 0.1, seed, 3)
    dg = torch.zeros(H, device="cuda"); dbt = torch.zeros(H, device="cuda"); dbias = torch.zeros(H, device="cuda")
    tf = timeit(lambda: ops.ln_fwd(x, r, g, b, 1e-12, 0.1, seed, 3))
    tb = timeit(lambda: ops.ln_bwd(dy, x, r, g, mean, rstd, dg, dbt, 0.1, seed, 3, dbias=dbias))
    print("ln_fwd %.1f us (%.2f TB/s)  ln_bwd %.1f us (%.2f TB/s)" % (tf * 1e6, 3 * M * H * 2 / tf / 1e12, tb * 1e6, 5 * M * H * 2 / tb / 1e12))

if __name__ == "__main__":
    main()
