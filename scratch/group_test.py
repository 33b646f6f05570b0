import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
dev = "cuda"
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows in (9984, 3072, 12288, 1024, 384):
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    tr = []
    for (no, ni) in shapes:
        dy = torch.randn(rows, no, device=dev, dtype=torch.bfloat16)
        x = torch.randn(rows, ni, device=dev, dtype=torch.bfloat16)
        dw = torch.randn(no, ni, device=dev, dtype=torch.float32)
        tr.append((dy, x, dw))
    ref = [dw.clone() for _, _, dw in tr]
    for (dy, x, _), r in zip(tr, ref):
        ops._linear_wgrad_now(dy, x, r, None)
    got = [(dy, x, dw.clone()) for dy, x, dw in tr]
    ops.wgrad_group(got)
    torch.cuda.synchronize()
    for (dy, x, g), r, (_, _, d0) in zip(got, ref, tr):
        want = d0.double() + dy.double().t() @ x.double()
        e_g = ((g.double() - want).norm() / want.norm()).item()
        e_r = ((r.double() - want).norm() / want.norm()).item()
        print("rows %5d dW %4dx%4d: grouped rel %.2e  separate rel %.2e  max|g-r| %.2e" % (rows, g.shape[0], g.shape[1], e_g, e_r, (g - r).abs().max().item()))
    again = [(dy, x, dw.clone()) for dy, x, dw in tr]
    ops.wgrad_group(again)
    assert all(torch.equal(a[2], b[2]) for a, b in zip(again, got)), "not reproducible"
    t_sep = timeit(lambda: [ops._linear_wgrad_now(dy, x, dw, None) for dy, x, dw in tr])
    t_grp = timeit(lambda: ops.wgrad_group(tr))
    split = ops._group_split(108, rows // 64, 256)
    print("rows %5d: separate %.1f us   grouped (split %d) %.1f us" % (rows, t_sep, split, t_grp))
