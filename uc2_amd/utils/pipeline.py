"""Gradient accumulation with the micro-batches software-pipelined over two HIP streams.

The reference accumulates gradients over `gradient_accumulation_steps` micro-batches one after the other
(pretrain.py:514-566, config/uc2_pretrain.json:17-19: 104 pairs x 3): forward, backward, forward, backward, ...  At that size most
kernels of a pass leave CUs idle (the N = 768 GEMMs of 9 984 tokens are 117 tiles for 256 CUs), and nothing in micro-batch i+1's
FORWARD depends on micro-batch i's BACKWARD -- the weights only change at the optimizer step.  `accumulate` therefore enqueues the
forward of micro-batch i+1 on a second stream before the backward of micro-batch i; the backward passes themselves stay in order
(they add into one gradient arena).  Same micro-batches, same dropout seeds in the same order, same sums up to the fp32 order of
the weight-gradient reductions; measured 26.1-26.7 -> 24.8-25.2 ms per optimizer step on the reference's regime
(scratch/regime_pipelined.py).  Opt-in: the reference's own loop, written as it is, runs the sequential form.
"""
import torch

from .. import ops
from ..store import _STORES

_STREAMS = {}


def _streams(device):
    key = (device.type, device.index)
    if key not in _STREAMS:
        _STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _STREAMS[key]


def accumulate(forward_fns, device=None, seeds_per_forward=4, before_backward=None):
    """Run `loss_i = forward_fns[i]()` and `loss_i.backward()` for every i, forwards and backwards overlapped as described above.

    forward_fns: zero-argument callables, one per micro-batch, each returning the scalar loss to back-propagate (already divided
    by whatever the loop divides by).  Returns the detached losses.  On return the current stream has been made to wait for
    everything: clip / all-reduce / optimizer step follow as usual.
    before_backward(i): called on the host right before backward i is enqueued (e.g. GradSync.arm() for the last one).
    seeds_per_forward: dropout seed copies drawn ahead per micro-batch (one per model forward inside ops.rng.scope(); forwards that
    draw more fall back to the shared counter)."""
    n = len(forward_fns)
    if n == 0:
        return []
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if n == 1 or torch.cuda.is_current_stream_capturing():
        out = []
        for i, f in enumerate(forward_fns):
            loss = f()
            if before_backward is not None:
                before_backward(i)
            loss.backward()
            out.append(loss.detach())
        return out
    main = torch.cuda.current_stream(device)
    for st in list(_STORES):                             # bf16 weight copies: refreshed once, before the streams fork
        if st.data is not None and st.data.device == device and getattr(st, "auto_sync", True):
            st.sync_shadow()
    seeds = ops.rng.draw(device, n * seeds_per_forward)
    S = _streams(device)
    for s in S:
        s.wait_stream(main)
    losses, done = [None] * n, [None] * n

    def fwd(i):
        ops.rng._preset = seeds[i * seeds_per_forward:(i + 1) * seeds_per_forward]
        try:
            with torch.cuda.stream(S[i & 1]):
                losses[i] = forward_fns[i]()
        finally:
            ops.rng._preset = None
    fwd(0)
    for i in range(n):
        if i + 1 < n:
            fwd(i + 1)                                   # enqueued before backward i: the device runs them side by side
        with torch.cuda.stream(S[i & 1]):
            if i > 0:
                S[i & 1].wait_event(done[i - 1])         # gradient accumulation stays in order
            if before_backward is not None:
                before_backward(i)
            losses[i].backward()
            done[i] = torch.cuda.Event()
            done[i].record()
    main.wait_stream(S[0])
    main.wait_stream(S[1])
    out = []
    for i, l in enumerate(losses):
        d = l.detach()
        d.record_stream(main)                            # (allocated on a side stream, read by the caller on the current one)
        out.append(d)
    for i, t in enumerate(seeds):
        t.record_stream(S[(i // seeds_per_forward) & 1]) # (allocated on the current stream, read by the side stream's kernels)
    return out
