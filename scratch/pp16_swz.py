import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
DIAG = lambda m: (m & 0xFFFF) << 8
out = []
for (ta, tb, M, N, K, sp) in [(0, 1, 98304, 768, 3072, 1), (1, 0, 3072, 768, 98304, 7), (1, 1, 3072, 768, 98304, 7), (1, 1, 3072, 768, 9984, 4)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=dev, dtype=torch.bfloat16)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if sp > 1 else torch.bfloat16)
    ref = ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=c.clone(), accumulate=sp > 1, split_k=sp, variant=8)
    got = ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=torch.zeros_like(c), accumulate=sp > 1, split_k=sp, variant=12)
    err = ((got.double() - ref.double()).norm() / ref.double().norm()).item()
    r = []
    for mode in (0, 0x8):
        t = min(timeit(lambda: ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=c, accumulate=sp > 1, split_k=sp, variant=12, flags=DIAG(mode))) for _ in range(2))
        r.append(t)
    out.append("ta=%d tb=%d K=%d: whole %.1f main %.1f (err %.1e)" % (ta, tb, K, r[0], r[1], err))
print(" | ".join(out))
