"""ping-pong GEMM kernel with the fused epilogues of the encoder layer (not a test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
from bench_gemm import timeit

def main():
    lib = _lib.load()
    M = 49152
    cases = [("ffn1 gelu", False, 3072, 768, ops.EPI_GELU), ("ffn1 gelu nopre", False, 3072, 768, ops.EPI_GELU), ("ffn1 tanh", False, 3072, 768, ops.EPI_TANH), ("ffn1 none", False, 3072, 768, ops.EPI_NONE),
             ("out add", False, 768, 768, ops.EPI_ADD), ("ffn2 add", False, 768, 3072, ops.EPI_ADD),
             ("dgrad ffn2 dgelu", True, 3072, 768, ops.EPI_DGELU), ("dgrad qkv add", True, 768, 2304, ops.EPI_ADD)]
    for name, tb, n, k, epi in cases:
        a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
        b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
        bias = None if tb else torch.randn(n, device="cuda")
        aux = torch.randn((M, n), device="cuda").to(torch.bfloat16)
        pre = torch.empty((M, n), dtype=torch.bfloat16, device="cuda") if (epi == ops.EPI_GELU and "nopre" not in name) else None
        out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: ops.gemm(a, b, M, n, k, tb=tb, out=out, bias=bias, epi=epi,
                              aux_in=aux if epi in (ops.EPI_ADD, ops.EPI_DGELU) else None, aux_out=pre)
        row = []
        for v in (8, 9, 7, 2):
            with ops.force_variant(v):
                row.append("v%d %6.1f" % (v, 2.0 * M * n * k / timeit(fn) / 1e12))
        for sk in (1, 2, 3):
            with ops.force_variant(8, flags=sk << 4):
                row.append("v8 skew%d %6.1f" % (sk, 2.0 * M * n * k / timeit(fn) / 1e12))
        print("%-18s N=%5d K=%5d  " % (name, n, k) + "  ".join(row))

if __name__ == "__main__":
    main()
