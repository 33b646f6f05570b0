"""which GEMM shapes an MLM step launches outside the encoder layers (variant / split per call), one step at PAIRS pairs"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import ops, _lib
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model); st.sync_shadow(); st.auto_sync = False
P = int(os.environ.get("PAIRS", "6144"))
b = bench.synth_batch(P, "mlm", 3, dev)
lib = _lib.load()
f0 = lib.uc2_gemm
log = collections.Counter()
def w(*a):
    # dtype, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, c_f32, bias, epi, aux_in, aux_out, ldaux, accumulate, split_k, variant, ...
    log[(a[1], a[2], a[3], a[4], a[5], a[12], a[14], a[18], a[19], a[20])] += 1
    return f0(*a)
class Proxy:
    def __getattr__(self, n):
        return w if n == "uc2_gemm" else getattr(lib, n)
_lib.load = lambda: Proxy()
for i in range(2):
    log.clear()
    l = model(b, "mlm", compute_loss=True)
    (l[0] if isinstance(l, tuple) else l).mean().backward()
    ops.join_side_streams()
    model.zero_grad()
torch.cuda.synchronize()
for k, n in sorted(log.items(), key=lambda kv: -kv[1]):
    ta, tb, M, N, K, cf32, epi, acc, sp, v = k
    if n <= 8:
        print("%2d x  %s%s M=%d N=%d K=%d fp32out=%d epi=%d acc=%d split=%d variant=%d" % (n, "T" if ta else "N", "T" if tb else "N", M, N, K, cf32, epi, acc, sp, v))
