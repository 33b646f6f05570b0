"""EXPERIMENT: the reference's regime (104-pair micro-batches x 3 accumulation) with the micro-batches software-pipelined over two HIP
streams -- forward of micro-batch i+1 beside the backward of micro-batch i (the backward passes themselves stay in order: they
accumulate into one gradient arena).  Same micro-batches, same sums; only the enqueue order and the streams change.
python scratch/regime_pipelined.py [task] [opt steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
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
rb = [bench.synth_batch(bench.REF_MICRO, task, 9000 + i, dev) for i in range(bench.REF_ACCUM)]
main = torch.cuda.current_stream(dev)
S = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

def fwd(b):
    loss = model(b, task, compute_loss=True)
    loss = loss[0] if isinstance(loss, tuple) else loss
    return loss.mean()

def step_seq():
    for b in rb:
        fwd(b).backward()
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)

def step_pipe():
    n = len(rb)
    for s in S:
        s.wait_stream(main)                      # the optimizer step before us
    losses = [None] * n
    done = [None] * n
    with torch.cuda.stream(S[0]):
        losses[0] = fwd(rb[0])
    for i in range(n):
        if i + 1 < n:
            with torch.cuda.stream(S[(i + 1) & 1]):
                losses[i + 1] = fwd(rb[i + 1])   # enqueued before backward i: runs beside it
        with torch.cuda.stream(S[i & 1]):
            if i > 0:
                S[i & 1].wait_event(done[i - 1])  # gradient accumulation stays in order
            losses[i].backward()
            done[i] = torch.cuda.Event()
            done[i].record()
    main.wait_stream(S[0]); main.wait_stream(S[1])
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)

S3 = [torch.cuda.Stream(dev) for _ in range(3)]


def step_fwd_first():
    """all forwards at once on three streams, then the backward passes in order"""
    n = len(rb)
    for s in S3:
        s.wait_stream(main)
    losses, done = [None] * n, [None] * n
    for i in range(n):
        with torch.cuda.stream(S3[i]):
            losses[i] = fwd(rb[i])
    for i in range(n):
        with torch.cuda.stream(S3[i]):
            if i > 0:
                S3[i].wait_event(done[i - 1])
            losses[i].backward()
            done[i] = torch.cuda.Event()
            done[i].record()
    for s in S3:
        main.wait_stream(s)
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)


def step_unordered():
    """TIMING ONLY (the gradient sums race): the three micro-batches on three streams with no order between their backward passes --
    what a second gradient arena would make legal"""
    n = len(rb)
    for s in S3:
        s.wait_stream(main)
    for i in range(n):
        with torch.cuda.stream(S3[i]):
            fwd(rb[i]).backward()
    for s in S3:
        main.wait_stream(s)
    _, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0, fused=True)
    opt.step(grad_scale=coef, zero_grad=True)


for name, step in (("sequential", step_seq), ("pipelined", step_pipe), ("unordered(timing only)", step_unordered), ("sequential", step_seq), ("pipelined", step_pipe), ("unordered(timing only)", step_unordered)):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nsteps
    print("%s %s regime: %.2f ms per optimizer step, %.0f pairs/s, mfma %.4f" % (name, task, dt * 1e3, 312 / dt, 312 / dt * 49.94e9 / 2.5e15), flush=True)
