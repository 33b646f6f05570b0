"""ping-pong GEMM kernel diagnostics (not a test): full kernel vs main loop only, a few shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib
from bench_gemm import timeit

def main():
    lib = _lib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = B * 96
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 7]
    shapes = [("fwd qkv", False, False, M, 2304, 768), ("fwd ffn2", False, False, M, 768, 3072),
              ("dgrad ffn1", False, True, M, 768, 3072), ("dgrad ffn2", False, True, M, 3072, 768),
              ("wgrad ffn1", True, True, 3072, 768, M)]
    x = torch.empty(49152 * 2304, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    t = timeit(lambda: x.fill_(1.0)); print("fill_ 226 MB: %.2f TB/s" % (x.numel() * 2 / t / 1e12))
    t = timeit(lambda: y.copy_(x)); print("copy_ 226 MB: %.2f TB/s written" % (x.numel() * 2 / t / 1e12))
    for name, ta, tb, m, n, k in shapes:
        a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
        b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
        wg = ta and tb
        out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device="cuda")
        split = 9 if wg else 1
        row = []
        for v in variants:
            for mode in (0, 8, 64, 8 + 1, 8 + 2, 8 + 1 + 2):
                fn = lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, accumulate=wg, split_k=split, variant=v, flags=mode << 8)
                t = timeit(fn)
                row.append("v%d/%s %6.1f" % (v, {0: "full", 8: "loop", 64: "epi", 9: "loop-nodma", 10: "loop-nord", 11: "loop-mfma"}[mode], 2.0 * m * n * k / t / 1e12))
        print("%-11s M=%6d N=%5d K=%6d  " % (name, m, n, k) + "  ".join(row))

if __name__ == "__main__":
    main()
