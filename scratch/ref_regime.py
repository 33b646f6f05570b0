"""the reference's regime (104-pair micro-batches x 3 accumulation steps, config/uc2_pretrain.json:17-19), ITM, for rocprofv3 --stats"""
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
task = sys.argv[1] if len(sys.argv) > 1 else "itm"
rb = [bench.synth_batch(104, task, 9000 + i, dev) for i in range(3)]
def step():
    for b in rb:
        loss = model(b, task, compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for _ in range(N): step()
tc = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print("%s: %.2f ms per optimizer step (host enqueue %.2f ms), %.0f pairs/s" % (task, t / N * 1e3, tc / N * 1e3, 312 * N / t))
