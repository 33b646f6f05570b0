"""the itm.py finetune window (bench.py itm_rank_finetune: 120 sequences x 8 accumulation steps, num_bb in [10,100]) as a stand-alone
loop: python scratch/rank_step.py [opt steps]   (env knobs: UC2_ACCUM_OVERLAP_MAX_ROWS, UC2_PAD_ROWS, ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uc2_amd import ops
from uc2_amd.config import cfg as knobs
from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype, store_of
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = VLXLMRForImageTextRetrieval(bench.make_cfg(12), img_dim=2048, margin=0.2).to(dev).train()
set_compute_dtype(model, torch.bfloat16)
opt = AdamW(param_groups(model, 0.0), lr=5e-5, betas=(0.9, 0.98))
st = store_of(model); st.sync_shadow(); st.auto_sync = False
rk = [bench.synth_batch_varlen(120, "itm", 500 + j, dev, reg_range=(10, 100), sample_size=3)[0] for j in range(16)]
def step(i):
    for j in range(8):
        loss = model(rk[(i % 2) * 8 + j], compute_loss=True)
        loss.mean().backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 2.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(nsteps): step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nsteps
print("overlap_max_rows %d pad_rows %s: %.2f ms per optimizer step, %.0f triplets/s, overlap passes %d, fallbacks %d"
      % (knobs.accum_overlap_max_rows, knobs.pad_rows, dt * 1e3, 320 / dt, sum(s.passes for s in ops._accum.values()), ops.gemm_fallbacks()), flush=True)
