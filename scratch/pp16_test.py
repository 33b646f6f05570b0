# variant 12 (ping-pong on the 16x16x32 MFMA) against variant 8 and an fp32 product, every epilogue kind; then the step's shapes
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
torch.manual_seed(0)
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
ok = True
def check(name, got, ref, tol):
    global ok
    e = rel(got, ref)
    flag = "" if e < tol else "   <-- FAIL"
    if e >= tol: ok = False
    print("%-34s rel %.2e%s" % (name, e, flag))
for (ta, tb, M, N, K) in [(0, 0, 256, 256, 128), (0, 0, 512, 768, 768), (0, 1, 512, 768, 1024), (1, 0, 512, 512, 256), (1, 1, 768, 768, 512), (0, 0, 4096, 2304, 768)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=dev, dtype=torch.bfloat16) * 0.05
    bias = torch.randn(N, device=dev)
    A = (a.t() if ta else a).float(); B = (b.t() if tb else b).float()
    ref = A @ B.t() + bias
    o8 = ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), bias=bias, variant=8)
    for _ in range(2):
        o12 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), bias=bias, out=o12, variant=12)
    check("plain ta=%d tb=%d %dx%dx%d vs fp32" % (ta, tb, M, N, K), o12.float(), ref, 4e-3)
    check("   ... vs v8", o12.float(), o8.float(), 3e-3)
    # split-K accumulate
    c0 = torch.randn(M, N, device=dev)
    sk = 2 if K >= 256 else 1
    r8 = ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=c0.clone(), accumulate=True, split_k=sk, variant=8)
    r12 = ops.gemm(a, b, M, N, K, ta=bool(ta), tb=bool(tb), out=c0.clone(), accumulate=True, split_k=sk, variant=12)
    check("   split %d accumulate vs v8" % sk, r12, r8, 1e-5)
# epilogues
M, N, K = 1024, 768, 512
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
bias = torch.randn(N, device=dev)
for deriv in (0, 2):
    p8 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); p12 = torch.zeros_like(p8)
    u8 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=p8, variant=8, flags=deriv)
    u12 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_GELU, aux_out=p12, variant=12, flags=deriv)
    check("gelu (deriv flag %d) out" % deriv, u12.float(), u8.float(), 3e-3); check("   second stream", p12.float(), p8.float(), 3e-3)
u8 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_GELU, variant=8); u12 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_GELU, variant=12)
check("gelu no aux", u12.float(), u8.float(), 3e-3)
t8 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_TANH, variant=8); t12 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_TANH, variant=12)
check("tanh", t12.float(), t8.float(), 3e-3)
aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
r8 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_ADD, aux_in=aux, variant=8); r12 = ops.gemm(x, w, M, N, K, bias=bias, epi=ops.EPI_ADD, aux_in=aux, variant=12)
check("add (NN)", r12.float(), r8.float(), 3e-3)
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); auxk = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
for (epi, fl, nm) in ((ops.EPI_ADD, 0, "add (NT)"), (ops.EPI_DGELU, 0, "dgelu"), (ops.EPI_DGELU, 2, "mul")):
    cs8 = torch.zeros(K, device=dev); cs12 = torch.zeros(K, device=dev)
    kw8 = dict(aux_out=cs8) if epi == ops.EPI_DGELU else {}
    kw12 = dict(aux_out=cs12) if epi == ops.EPI_DGELU else {}
    d8 = ops.gemm(dy, w, M, K, N, tb=True, epi=epi, aux_in=auxk, variant=8, flags=fl, **kw8)
    d12 = ops.gemm(dy, w, M, K, N, tb=True, epi=epi, aux_in=auxk, variant=12, flags=fl, **kw12)
    check(nm, d12.float(), d8.float(), 3e-3)
    if epi == ops.EPI_DGELU: check("   column sums", cs12, cs8, 2e-3)
# unsplit accumulate (EPI_ACC)
a = torch.randn(1024, 768, device=dev, dtype=torch.bfloat16); b = torch.randn(1024, 512, device=dev, dtype=torch.bfloat16); c0 = torch.randn(768, 512, device=dev)
r8 = ops.gemm(a, b, 768, 512, 1024, ta=True, tb=True, out=c0.clone(), accumulate=True, variant=8)
r12 = ops.gemm(a, b, 768, 512, 1024, ta=True, tb=True, out=c0.clone(), accumulate=True, variant=12)
check("unsplit accumulate", r12, r8, 1e-5)
print("ALL OK" if ok else "FAILURES")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 98304
xs = {k: torch.randn(M, k, device=dev, dtype=torch.bfloat16) for k in (768, 2304, 3072)}
ws = {(n, k): torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.03 for (n, k) in ((2304, 768), (768, 768), (3072, 768), (768, 3072))}
bs = {n: torch.randn(n, device=dev) for n in (768, 2304, 3072)}
pre = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
cases = [("fwd qkv", lambda v: ops.gemm(xs[768], ws[(2304, 768)], M, 2304, 768, bias=bs[2304], variant=v), 2304, 768),
         ("fwd out", lambda v: ops.gemm(xs[768], ws[(768, 768)], M, 768, 768, bias=bs[768], variant=v), 768, 768),
         ("fwd ffn1 gelu'", lambda v: ops.gemm(xs[768], ws[(3072, 768)], M, 3072, 768, bias=bs[3072], epi=ops.EPI_GELU, aux_out=pre, flags=2, variant=v), 3072, 768),
         ("fwd ffn2", lambda v: ops.gemm(xs[3072], ws[(768, 3072)], M, 768, 3072, bias=bs[768], variant=v), 768, 3072),
         ("dgrad ffn2 mul", lambda v: ops.gemm(xs[768], ws[(768, 3072)], M, 3072, 768, tb=True, epi=ops.EPI_DGELU, aux_in=pre, flags=2, variant=v), 3072, 768),
         ("dgrad ffn1 add", lambda v: ops.gemm(xs[3072], ws[(3072, 768)], M, 768, 3072, tb=True, epi=ops.EPI_ADD, aux_in=xs[768], variant=v), 768, 3072),
         ("dgrad qkv add", lambda v: ops.gemm(xs[2304], ws[(2304, 768)], M, 768, 2304, tb=True, epi=ops.EPI_ADD, aux_in=xs[768], variant=v), 768, 2304)]
dw = {(n, k): torch.zeros(n, k, device=dev) for (n, k) in ws}
cases += [("wgrad %dx%d" % (n, k), (lambda n, k: (lambda v: ops.gemm(xs[n] if n in xs else xs[768], xs[k], n, k, M, ta=True, tb=True, out=dw[(n, k)], accumulate=True, split_k=sp, variant=v)))(n, k), n, k)
          for (n, k, sp) in ((2304, 768, 9), (768, 768, 28), (3072, 768, 7), (768, 3072, 7)) for sp in (sp,)]
for name, fn, n, k in cases:
    r = []
    for v in (8, 12, 8, 12):
        t = timeit(lambda: fn(v))
        r.append(2.0 * M * n * k / t / 1e9)
    print("%-18s v8 %5.0f %5.0f   v12 %5.0f %5.0f TF/s   %+.1f %%" % (name, r[0], r[2], r[1], r[3], (max(r[1], r[3]) / max(r[0], r[2]) - 1) * 100))
