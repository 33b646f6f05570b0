# main loop only (diag 0x800) and epilogue only (0x4000): variant 8 against 12 per operand layout
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
for (ta, tb, M, N, K, sp) in [(0, 0, 98304, 768, 3072, 1), (0, 1, 98304, 768, 3072, 1), (1, 0, 3072, 768, 98304, 7), (1, 1, 3072, 768, 98304, 7), (1, 1, 3072, 768, 9984, 4)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=dev, dtype=torch.bfloat16)
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if sp > 1 else torch.bfloat16)
    for mode, nm in ((0, "whole"), (0x8, "main loop only"), (0x40, "epilogue only")):
        r = {}
        for v in (8, 12, 8, 12):
            t = timeit(lambda: ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=c, accumulate=sp > 1, split_k=sp, variant=v, flags=DIAG(mode)))
            r[v] = min(r.get(v, 1e9), t)
        print("ta=%d tb=%d %dx%dx%d split %d  %-15s v8 %7.1f us  v12 %7.1f us  %+.1f %%" % (ta, tb, M, N, K, sp, nm, r[8], r[12], (r[8] / r[12] - 1) * 100), flush=True)
