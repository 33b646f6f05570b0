"""the reference's regime (104-pair micro-batches x 3 accumulation, config/uc2_pretrain.json:17-19) as a stand-alone loop for rocprofv3:
python scratch/regime_step.py [task] [opt steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import ops
from uc2_amd.model.model import VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
task = sys.argv[1] if len(sys.argv) > 1 else "itm"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = VLXLMRForPretraining(bench.make_cfg(12), img_dim=2048, img_label_dim=1601).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
st = store_of(model)
opt = AdamW(param_groups(model, 0.01), lr=4e-5, betas=(0.9, 0.98))
st.sync_shadow(); st.auto_sync = False
PAIRS = int(os.environ.get("PAIRS", str(bench.REF_MICRO)))            # micro-batch size (default: the reference's 104)
ACCUM = int(os.environ.get("ACCUM", str(bench.REF_ACCUM)))          # micro-batches per optimizer step (default: the reference's 3)
rb = [bench.synth_batch(PAIRS, task, 9000 + i, dev) for i in range(ACCUM)]
def step():
    for b in rb:
        loss = model(b, task, compute_loss=True)
        loss = loss[0] if isinstance(loss, tuple) else loss
        loss.mean().backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(nsteps): step()
th = (time.perf_counter() - t0) / nsteps          # host enqueue time (the loop does not wait for the GPU)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nsteps
print("host enqueue %.2f ms per optimizer step" % (th * 1e3))
n = PAIRS * ACCUM
from uc2_amd.config import cfg as knobs
print("%s regime, %d-pair micro-batches x %d, overlap %s (max rows %d): %.2f ms per optimizer step, %.0f pairs/s, mfma %.4f, overlap passes %d"
      % (task, PAIRS, ACCUM, knobs.accum_overlap, knobs.accum_overlap_max_rows, dt * 1e3, n / dt, n / dt * 49.94e9 / 2.5e15, sum(s.passes for s in ops._accum.values())))
