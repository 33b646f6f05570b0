"""cProfile of the host side of the reference-regime optimizer step (scratch/regime_step.py): python scratch/regime_hostprof.py [native 0/1]"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UC2_NATIVE_LAYER"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import torch
import bench
from uc2_amd import ops
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model)
opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
st.sync_shadow(); st.auto_sync = False
rb = [bench.synth_batch(bench.REF_MICRO, "itm", 9000 + i, dev) for i in range(bench.REF_ACCUM)]
def step():
    for b in rb:
        loss = model(b, "itm", compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(6): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
