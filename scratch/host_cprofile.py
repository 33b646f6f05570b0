"""cProfile of the host side of a regime micro-batch: the forward on the main thread, BertLayerFn.backward inside autograd's device
thread (the profiler is enabled inside the wrapped backward).  python scratch/host_cprofile.py [pairs]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd.ops import layer as L
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model)
opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
st.sync_shadow(); st.auto_sync = False
PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 104
rb = [bench.synth_batch(PAIRS, "itm", 9000 + i, dev) for i in range(3)]
pf, pb, po = cProfile.Profile(), cProfile.Profile(), cProfile.Profile()
ON = [False]
f0 = L.BertLayerFn.backward
def bw(*a):
    if not ON[0]:
        return f0(*a)
    pb.enable()
    try:
        return f0(*a)
    finally:
        pb.disable()
L.BertLayerFn.backward = staticmethod(bw)
def step():
    for b in rb:
        if ON[0]: pf.enable()
        loss = model(b, "itm", compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        if ON[0]: pf.disable()
        loss.mean().backward()
    if ON[0]: po.enable()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
    if ON[0]: po.disable()
for _ in range(3): step()
torch.cuda.synchronize()
ON[0] = True
K = 10
for _ in range(K): step()
torch.cuda.synchronize()
for name, p, n in (("FORWARD (30 micro-batches)", pf, 45), ("BertLayerFn.backward (360 calls)", pb, 45), ("clip + AdamW (10 steps)", po, 25)):
    s = io.StringIO()
    pstats.Stats(p, stream=s).sort_stats("tottime").print_stats(n)
    print("=" * 30, name); print(s.getvalue())
