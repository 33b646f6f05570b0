# how fast would ONE launch over the 108 weight-gradient tiles of a layer be at 9984 tokens?  (one GEMM with the same tile count and K)
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9984
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def wg(nout, nin, split, variant=8):
    dy = torch.randn(rows, nout, device=dev, dtype=torch.bfloat16)
    x = torch.randn(rows, nin, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(nout, nin, device=dev, dtype=torch.float32)
    return timeit(lambda: ops.gemm(dy, x, nout, nin, rows, ta=True, tb=True, out=dw, accumulate=True, split_k=split, variant=variant))
tot = 0
for (nout, nin) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    v, sp = ops.gemm_plan(torch.bfloat16, True, True, nout, nin, rows, True)
    t = wg(nout, nin, sp, v if v >= 0 else 8)
    best = min((wg(nout, nin, s), s) for s in (2, 3, 4, 5, 6, 7, 9, 13, 26) if (156 // s) >= 2)
    print("dW %4d x %4d: plan (v%d, split %d) %.1f us; best split %d %.1f us" % (nout, nin, v, sp, t, best[1], best[0]))
    tot += t
print("sum of the four launches: %.1f us" % tot)
for split in (2, 3, 4, 5, 6, 7, 9, 13):
    t = wg(768, 9216, split)
    print("one launch, 108 tiles (768 x 9216), split %2d: %.1f us" % (split, t))
