# dE[V,768] += dlogits^T z  (the tied decoder's weight gradient): ping-pong kernel with the accumulate epilogue vs the generic kernel
import sys, time, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
torch.manual_seed(0)
dev = "cuda"
def run(V, H, m, variant, C, a, b, lda):
    return ops.gemm(a, b, V, H, m, ta=True, tb=True, out=C, accumulate=True, lda=lda, variant=variant)
# correctness on a small shape (lda > M like the padded vocabulary)
for (V, H, m) in [(1024, 768, 512), (2560, 768, 4608)]:
    a = torch.randn(m, V + 256, device=dev, dtype=torch.bfloat16)[:, :V]
    b = torch.randn(m, H, device=dev, dtype=torch.bfloat16)
    C0 = torch.randn(V, H, device=dev, dtype=torch.float32)
    c8, c0 = C0.clone(), C0.clone()
    run(V, H, m, 8, c8, a, b, a.stride(0)); run(V, H, m, 0, c0, a, b, a.stride(0))
    ref = C0.double() + a.double().t() @ b.double()
    print("V=%d m=%d  v8 vs fp64 max %.3e   generic vs fp64 max %.3e   v8 vs generic max %.3e" % (
        V, m, (c8.double() - ref).abs().max().item(), (c0.double() - ref).abs().max().item(), (c8 - c0).abs().max().item()))
    c8b = C0.clone(); run(V, H, m, 8, c8b, a, b, a.stride(0))
    assert torch.equal(c8, c8b)
V, H = 250112, 768
for m in (4608, 4352):
    a = torch.randn(m, V, device=dev, dtype=torch.bfloat16) * 0.01
    b = torch.randn(m, H, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(V, H, device=dev, dtype=torch.float32)
    for variant in (8, 8, 8):
        for _ in range(2): run(V, H, m, variant, C, a, b, V)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(V, H, m, variant, C, a, b, V)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        print("m=%d variant %d: %.3f ms  %.0f TF/s" % (m, variant, t, 2.0 * V * H * m / t / 1e9))
