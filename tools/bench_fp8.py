"""fp8 vs bf16 GEMM rates at the large-geometry shapes (not a test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops
from bench_gemm import timeit
for name, M, N, K in [("large qkv", 66560, 3072, 1024), ("large ffn1", 66560, 4096, 1024), ("large ffn2", 66560, 1024, 4096),
                      ("base ffn1", 98304, 3072, 768), ("base ffn2", 98304, 768, 3072)]:
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    x8, sx = ops.fp8_quantize(x)
    w8, sw = ops.fp8_quantize(w)
    t8 = timeit(lambda: ops.gemm_fp8(x8, sx, w8, sw, bias=bias))
    tq = timeit(lambda: ops.fp8_quantize(x))
    row = ["fp8 %7.1f TF/s (%.0f us; quantising the activation %.0f us)" % (2.0 * M * N * K / t8 / 1e12, t8 * 1e6, tq * 1e6)]
    for v in (8, 1):
        if v == 8 and (M % 256 or N % 256 or K % 128):
            continue
        t = timeit(lambda: ops.gemm(x, w, M, N, K, bias=bias, variant=v))
        row.append("bf16 v%d %7.1f" % (v, 2.0 * M * N * K / t / 1e12))
    print("%-11s M=%6d N=%5d K=%5d  " % (name, M, N, K) + "  ".join(row))
