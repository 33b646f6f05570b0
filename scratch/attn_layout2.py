# does the QKV access pattern bound the attention kernels?  same heads, two layouts: [B*L, 3*12*64] (a head's rows are 128-byte
# segments at a 4608-byte stride) against nh = 1 with B*12 "batches" ([L, 3*64] per head: compact 36 KB per head)
import sys, torch
sys.path.insert(0, "/root/repo")
from uc2_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
L, D = 96, 64
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for (B, nh) in ((1024, 12), (12288, 1)):
    H = nh * D
    qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
    mask = torch.zeros(B, L, device="cuda")
    dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 3, impl=2)
    tf = min(timeit(lambda: ops.attn_fwd(qkv, mask, B, L, nh, D, 0.1, seed, 3, impl=2)) for _ in range(2))
    db = torch.zeros(3 * H, device="cuda")
    tb = min(timeit(lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, nh, D, 0.1, seed, 3, impl=2, dbias=db)) for _ in range(2))
    print("B=%d nh=%d: fwd %.1f us  bwd %.1f us" % (B, nh, tf, tb))
