"""rolling-epilogue ping-pong GEMM (variant 10) against variant 8: bit-identical results, then interleaved timing"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uc2_amd import ops
from bench_gemm import timeit

torch.manual_seed(0)
def check(name, M, N, K, tb, epi=ops.EPI_NONE, bias=True, reps=3):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.05).to(torch.bfloat16)
    bv = torch.randn(N, device="cuda") if bias else None
    ref = ops.gemm(a, b, M, N, K, tb=tb, bias=bv, epi=epi, variant=8)
    gen = ops.gemm(a, b, M, N, K, tb=tb, bias=bv, epi=epi, variant=99)
    ok = True
    for r in range(reps):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm(a, b, M, N, K, tb=tb, bias=bv, epi=epi, out=out, variant=10)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        if not same:
            d = (out.float() - ref.float())
            bad = (~(d == 0)) | torch.isnan(out.float())
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("  MISMATCH %s rep %d: %d bad elems, rows %s..%s (n=%d) cols %s..%s (n=%d), maxerr %.3g, nan %d" % (
                name, r, int(bad.sum()), rows[:1].tolist(), rows[-1:].tolist(), rows.numel(), cols[:1].tolist(), cols[-1:].tolist(), cols.numel(),
                float(d[~torch.isnan(d)].abs().max()) if (~torch.isnan(d)).any() else -1, int(torch.isnan(out.float()).sum())))
            # row-block pattern of the first bad tile
            r0 = int(rows[0]) // 256 * 256; c0 = int(cols[0]) // 256 * 256
            blk = bad[r0:r0 + 256, c0:c0 + 256].view(8, 32, 4, 64).any(3).any(1)
            print("  first bad tile (%d,%d): 32-row x 64-col blocks bad:\n%s" % (r0, c0, blk.int().cpu().numpy()))
            ok = False
            break
    print("%-26s M=%6d N=%5d K=%5d tb=%d epi=%d : %s (vs generic rel %.2e)" % (name, M, N, K, tb, epi, "bit-identical to v8" if ok else "FAIL",
          float((ref.float() - gen.float()).norm() / gen.float().norm())), flush=True)
    return ok

ok = True
for (name, M, N, K, tb, epi, bias) in [
        ("one tile", 256, 256, 256, False, 0, True), ("one tile nt", 256, 256, 256, True, 0, False),
        ("4 tiles K=768", 512, 512, 768, False, 0, True), ("few items", 4096, 768, 768, False, 0, True),
        ("many items", 32768, 2304, 768, False, 0, True), ("many items nt", 32768, 768, 2304, True, 0, False),
        ("many items K=256", 98304, 768, 256, False, 0, True), ("gelu noaux", 16384, 3072, 768, False, ops.EPI_GELU, True),
        ("tanh", 16384, 768, 768, False, ops.EPI_TANH, True), ("odd item count", 256 * 37, 768 * 3, 512, False, 0, True)]:
    ok &= check(name, M, N, K, tb, epi, bias)
print("ALL OK" if ok else "SOME FAILED", flush=True)

M = 98304
cases = [("fwd qkv", False, 2304, 768, 0, True), ("fwd out", False, 768, 768, 0, True), ("fwd ffn1 plain", False, 3072, 768, 0, True),
         ("fwd ffn1 gelu-noaux", False, 3072, 768, ops.EPI_GELU, True), ("fwd ffn2", False, 768, 3072, 0, True),
         ("dgrad out", True, 768, 768, 0, False), ("dgrad qkv plain", True, 768, 2304, 0, False), ("dgrad ffn2 plain", True, 3072, 768, 0, False)]
for name, tb, n, k, epi, bias in cases:
    a = torch.randn((M, k), device="cuda").to(torch.bfloat16)
    b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(torch.bfloat16)
    bv = torch.randn(n, device="cuda") if bias else None
    out = torch.empty((M, n), dtype=torch.bfloat16, device="cuda")
    res = {8: [], 10: []}
    for rep in range(4):
        for v in (8, 10):
            t = timeit(lambda: ops.gemm(a, b, M, n, k, tb=tb, bias=bv, epi=epi, out=out, variant=v))
            res[v].append(2.0 * M * n * k / t / 1e12)
    print("%-22s N=%5d K=%5d  v8 %s  v10 %s   best %.0f -> %.0f TF/s (%+.1f %%)" % (
        name, n, k, " ".join("%.0f" % x for x in res[8]), " ".join("%.0f" % x for x in res[10]), max(res[8]), max(res[10]),
        100 * (max(res[10]) / max(res[8]) - 1)), flush=True)
