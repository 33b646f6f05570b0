"""hipGraph capture of the forward+backward micro-step in the reference's regime (104 pairs x 3 accumulation steps): does it
capture, and what does replay give against eager enqueue (host-bound check)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import ops
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
ops.rng.manual_seed(1, dev)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
st = store_of(model); st.sync_shadow(); st.auto_sync = False
task = "itm"
rb = [bench.synth_batch(104, task, 9000 + i, dev) for i in range(3)]
static = {k: v.clone() for k, v in rb[0].items()}
def micro(b):
    loss = model(b, task, compute_loss=True)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.mean().backward()
    return loss
def opt_step():
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
def eager():
    for b in rb: micro(b)
    opt_step()
for _ in range(3): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for _ in range(N): eager()
torch.cuda.synchronize()
print("eager : %.2f ms per optimizer step" % ((time.perf_counter() - t0) / N * 1e3), flush=True)
# capture one micro-step on static inputs
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): micro(static)
torch.cuda.current_stream().wait_stream(s)
opt_step()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    sl = micro(static)
torch.cuda.synchronize()
active = [p for p in model.parameters() if p.grad is not None]     # the gradient views the captured backward installed
opt_step()
def graphed():
    for b in rb:
        for k in static: static[k].copy_(b[k])
        g.replay()
    for p in active:                                   # replay fills the arena; the Python-side p.grad views are re-attached
        if p.grad is None:
            p.grad = st.view(st.grad, p)
            p._uc2_gepoch = -1
    opt_step()
for _ in range(3): graphed()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): graphed()
torch.cuda.synchronize()
print("graph : %.2f ms per optimizer step (loss %.4f)" % ((time.perf_counter() - t0) / N * 1e3, float(sl.mean())), flush=True)
