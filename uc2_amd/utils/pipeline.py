"""Gradient accumulation over micro-batches -- compatibility wrapper.

Round 5 offered `accumulate([...])` as an opt-in API that software-pipelined the micro-batches of an accumulation window over two
HIP streams (forward of micro-batch i+1 beside backward i).  Since round 6 the overlap happens INSIDE the top-level models
(`uc2_amd.ops.accum_pass`: a training forward of a small micro-batch runs on one of two library-owned streams, autograd runs its
backward there, and the loop's next forward runs beside it), so the reference's loop gets it as it is written (pretrain.py:514-566)
and this function is the plain loop.  What the round-5 version got wrong is handled where the overlap now lives: the
per-optimizer-step derived state (bf16 copies, W^T copies) is refreshed on the caller's stream or inside the ordered backward
passes, fp8 stores (delayed-scaling histories assume one in-order stream) and stores that re-cast their bf16 copies at every
forward run sequentially.
"""


def accumulate(forward_fns, device=None, seeds_per_forward=4, before_backward=None):
    """`loss_i = forward_fns[i](); loss_i.backward()` for every i, in order; returns the detached losses.
    before_backward(i): called on the host right before backward i is enqueued (e.g. GradSync.arm() for the last one).
    `device` and `seeds_per_forward` are accepted for round-5 callers and ignored."""
    out = []
    for i, f in enumerate(forward_fns):
        loss = f()
        if before_backward is not None:
            before_backward(i)
        loss.backward()
        out.append(loss.detach())
    return out
