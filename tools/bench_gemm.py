"""GEMM micro-benchmark (not a test): TFLOP/s of the hot-path shapes, pipelined vs generic kernel,
interleaved in one process (cdna_hip_programming.md rule 24), random bf16 data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uc2_amd import ops, _lib

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

VARIANTS = [1, 2, 6, 7, 8]   # see gf_launch2 in uc2_amd/csrc/gemm_fast.hip


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    M = B * 96
    lib = _lib.load()
    dev = "cuda"
    only = sys.argv[2] if len(sys.argv) > 2 else None
    shapes = [("fwd qkv   ", False, False, M, 2304, 768), ("fwd out   ", False, False, M, 768, 768),
              ("fwd ffn1  ", False, False, M, 3072, 768), ("fwd ffn2  ", False, False, M, 768, 3072),
              ("dgrad qkv ", False, True, M, 768, 2304), ("dgrad ffn1", False, True, M, 768, 3072),
              ("dgrad ffn2", False, True, M, 3072, 768), ("dgrad out ", False, True, M, 768, 768),
              ("wgrad qkv ", True, True, 2304, 768, M), ("wgrad ffn1", True, True, 3072, 768, M),
              ("wgrad ffn2", True, True, 768, 3072, M), ("wgrad out ", True, True, 768, 768, M)]
    for name, ta, tb, m, n, k in shapes:
        if only and only not in name:
            continue
        a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
        b = torch.randn((k, n) if tb else (n, k), device=dev).to(torch.bfloat16)
        wg = ta and tb
        out = torch.zeros((m, n), dtype=torch.float32 if wg else torch.bfloat16, device=dev)
        split = ops._wgrad_split(torch.bfloat16, m, n, k) if wg else 1
        res = []
        fn = lambda: ops.gemm(a, b, m, n, k, ta=ta, tb=tb, out=out, accumulate=wg, split_k=split)
        for rep in range(2):
            for v in VARIANTS:
                with ops.force_variant(v):
                    t = timeit(fn)
                if rep == 1:
                    res.append(2.0 * m * n * k / t / 1e12)
        diag = {}
        for mode, label in ((1, "fetch only"), (17, "fetch + epilogue"), (32, "compute only (ws)")):
            diag[label] = []
            for v in VARIANTS:
                with ops.force_variant(v, flags=mode << 8):
                    diag[label].append(2.0 * m * n * k / timeit(fn) / 1e12)
        with ops.force_variant(ops.GEMM_GENERIC):
            tg = 2.0 * m * n * k / timeit(fn) / 1e12
        print("%s M=%6d N=%5d K=%6d split=%2d  " % (name, m, n, k, split) + " ".join("v%d %6.1f" % (v, r) for v, r in zip(VARIANTS, res)) + "  generic %6.1f" % tg)
        for label, fo in diag.items():
            print("      %-16s (equivalent TF/s):           " % label + " ".join("v%d %6.1f" % (v, r) for v, r in zip(VARIANTS, fo)))

if __name__ == "__main__":
    main()
