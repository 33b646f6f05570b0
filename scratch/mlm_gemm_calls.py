"""which GEMM calls of an MLM step run outside the ping-pong kernels: python scratch/mlm_gemm_calls.py [pairs]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import ops
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.store import set_compute_dtype, store_of
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model)
st.sync_shadow(); st.auto_sync = False
b = bench.synth_batch(pairs, "mlm", 9000, dev)
def step():
    loss = model(b, "mlm", compute_loss=True)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.mean().backward()
for _ in range(2): step()
torch.cuda.synchronize()
calls = collections.Counter()
orig = ops.gemm
def spy(a, b_, M, N, K, **kw):
    import traceback
    v = kw.get("variant")
    if v not in (8, 12):
        fr = [f for f in traceback.extract_stack(limit=6)[:-1]]
        where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:])
        calls[(kw.get("ta", False), kw.get("tb", False), M, N, K, v, kw.get("split_k", 1), where)] += 1
    return orig(a, b_, M, N, K, **kw)
ops.gemm = spy
step()
ops.join_side_streams()
torch.cuda.synchronize()
for k, c in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(c, k)
