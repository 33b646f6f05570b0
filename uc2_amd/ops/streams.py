"""Streams beside the caller's: the weight-gradient side stream and the two accumulation-overlap streams, with the joins every
consumer of gradients performs.  Part of uc2_amd.ops."""

import torch

from .. import _lib
from .. import store as _store
from ..config import cfg as knobs

_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


# Weight-gradient GEMMs are off the critical path of backward (nothing downstream in the same backward pass reads
# dW), so they are enqueued on a side HIP stream and overlap the dgrad / LayerNorm / attention chain on the main
# stream.  Every consumer of gradients (optimizer, clipping, all-reduce, end of autograd's backward) joins first.
_side_streams = {}
_side_dirty = set()
_join_queued = [False]


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        # (same priority as the main stream: torch on ROCm offers (0, -1) only; a high-priority side stream measured 111.8-116.7 ms
        #  per step against 111.3-111.5)
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


_side_keep = []         # tensors the side stream reads: kept alive until the join (see _on_side_stream)


def join_side_streams(compute=True):
    """make the current stream wait for every weight-gradient kernel enqueued on a side stream and (compute=True: what every
    consumer of gradients asks for -- optimizer, clipping, all-reduce, zero_grad) for the forward / backward passes that the
    accumulation overlap put on its own two streams (accum_overlap below)"""
    for key in list(_side_dirty):
        torch.cuda.current_stream(torch.device(*key)).wait_stream(_side_streams[key])
    _side_dirty.clear()
    _side_keep.clear()                 # from here on the main stream is ordered behind their last reader: the blocks may be reused
    _join_queued[0] = False            # (a backward that raised never ran its callback: the next one must queue a new join)
    if compute:
        join_accum_streams()


def _queue_pass_callback(fn):
    """autograd end-of-pass callback that runs `fn` on the stream that is current NOW (the stream of the backward node that
    registers it).  The engine runs final callbacks in the thread and stream context of whoever called backward(); a pass whose
    forward ran on another stream (utils/pipeline.py, accum_overlap) must flush its deferred launches there."""
    s = torch.cuda.current_stream()

    def run():
        with torch.cuda.stream(s):
            fn()
    torch.autograd.Variable._execution_engine.queue_callback(run)


def _end_of_backward_join():
    _join_queued[0] = False
    join_side_streams(compute=False)


def pending_side_stream(device):
    """the weight-gradient side stream of `device` if kernels have been enqueued on it since the last join, else None.
    GradSync orders a layer's all-reduce behind it (uc2_comm_allreduce_bucket_after) instead of joining it into the main stream."""
    key = (device.type, device.index)
    return _side_streams[key] if key in _side_dirty else None


# --------------------------------------------------------------------------------------
# Gradient accumulation, overlapped without an API change.  The reference's loop runs micro-batch after micro-batch
# (pretrain.py:514-566: forward, backward, forward, backward, ..., all-reduce, clip, step; config/uc2_pretrain.json:17-19: 104
# pairs x 3).  At that size most kernels of a pass leave CUs idle (9 984 tokens = 117 tiles of 256 x 256 for 256 CUs), and nothing
# in micro-batch i+1's FORWARD depends on micro-batch i's BACKWARD: the weights only change at the optimizer step.  The top-level
# models (VLXLMRForPretraining / VLXLMRForImageTextRetrieval) therefore run a TRAINING forward of fewer than ACCUM_OVERLAP_MAX_ROWS
# tokens on one of two library-owned streams, alternating per call; autograd runs each node's backward on its forward's stream, so
# backward i is on stream i & 1 and the loop's next forward, enqueued right after it on the other stream, runs beside it:
#   * entry: the pass's stream waits for the caller's current stream (inputs, the optimizer's weights); exit: the caller's stream
#     waits for the pass's stream (the returned losses / scores are safe to use there) -- NOT for any backward;
#   * gradient accumulation stays in order: a pass's first backward node (_AccumMarker) makes its stream wait for the other one
#     (the arena's += are not atomic);
#   * every consumer of gradients (AdamW.step, clip_grad_norm_, all_reduce_and_rescale_tensors, zero_grad -- they all call
#     join_side_streams()) makes its stream wait for both; the per-layer all-reduce hooks of GradSync run inside the pass.
# Same micro-batches, same dropout seeds in the same order, same kernels, same accumulation order: gradients equal to the order of
# the float atomics (what two runs on one stream differ by).
# Off for: eval / no-grad forwards, fp8 stores (delayed-scaling histories assume one in-order stream), stores whose bf16 copies
# are re-cast at every forward (store.auto_sync: the re-cast would race with the backward beside it; AdamW.step turns auto_sync
# off), stream capture, UC2_ACCUM_OVERLAP=0, and for the first forward of a window when the previous window had only one (no
# accumulation: nothing to run beside).  Measured: profiles/r06_experiments.md.
_accum = {}


class _AccumState:
    def __init__(self, device):
        self.device = device
        self.streams = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        self.k = 0                         # eligible forwards since the last join
        self.used = [False, False]
        self.raw = tuple(s_.cuda_stream for s_ in self.streams)
        self.unjoined = False              # a backward pass was enqueued here that the caller's stream has not been ordered behind yet
        self.passes = 0                    # (statistics: passes that ran on the overlap streams)
        self.last_window = 0               # eligible forwards of the previous window (between two gradient consumers); 0 = not known yet


def _accum_state(device):
    key = (device.type, device.index)
    st = _accum.get(key)
    if st is None:
        st = _accum[key] = _AccumState(device)
    return st


def forget_accum_history():
    """the next window's first forward is overlapped again whatever the previous window looked like (tests; a caller that switches
    from single-batch steps to accumulation and does not want to give away one window)"""
    for st in _accum.values():
        st.last_window = 0


def join_accum_streams():
    """the current stream waits for every pass enqueued on the accumulation-overlap streams; the next pass starts on stream 0"""
    for st in _accum.values():
        if st.used[0] or st.used[1]:
            cur = torch.cuda.current_stream(st.device)
            for i in (0, 1):
                if st.used[i]:
                    cur.wait_stream(st.streams[i])
            st.used = [False, False]
        if st.k:
            st.last_window = st.k          # (a window = the eligible forwards between two joins that saw any)
        st.k = 0
        st.unjoined = False
    _store._GRAD_ACCESS[0] = None


def _on_grad_access():
    """ArenaParameter.grad was read or written from Python while a backward pass enqueued on an overlap stream may still be running:
    the accessing stream waits for the passes in flight -- unless it IS one of the overlap streams (the library's own code inside a
    pass: stream order covers it).  One wait per backward pass, whatever the number of parameters read."""
    raw = _lib.stream()
    pending = False
    for st in _accum.values():
        if not st.unjoined:
            continue
        if raw in st.raw:
            pending = True                   # inside a pass: leave the hook armed for the caller's stream
            continue
        cur = torch.cuda.current_stream(st.device)
        for i in (0, 1):
            if st.used[i]:
                cur.wait_stream(st.streams[i])
        st.unjoined = False
    if not pending:
        _store._GRAD_ACCESS[0] = None


class _AccumMarker(torch.autograd.Function):
    """identity on a pass's output: its backward is the first node of the pass's backward (it runs on the pass's stream) and
    orders that stream behind the other one, i.e. behind the previous micro-batch's backward"""

    @staticmethod
    def forward(ctx, x, state, idx, main):
        ctx.state, ctx.idx, ctx.main = state, idx, main
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        st, i = ctx.state, ctx.idx
        S = st.streams[i]
        # behind everything the caller's stream holds NOW: `loss = model(b); optimizer.zero_grad(); loss.backward()` zeroes the arena
        # on the caller's stream between this pass's forward and its backward, a gradient consumer may have joined (and reset `used`)
        # in between, the root gradient may come from the caller -- all of it is ordered before the first += of this pass.  (The
        # caller's stream never holds a backward pass: waiting for it costs the overlap nothing.)
        S.wait_stream(ctx.main)
        if st.used[1 - i]:
            S.wait_stream(st.streams[1 - i])
        st.unjoined = True                  # from here on .grad views are being written on this stream: store.ArenaParameter.grad
        _store._GRAD_ACCESS[0] = _on_grad_access
        return g, None, None, None


class accum_pass:
    """`with accum_pass(model_store, rows, tensors) as ap: out = ap.mark(forward(...))` -- see the block comment above.
    Inactive (a plain pass on the caller's stream) whenever one of the conditions does not hold."""

    def __init__(self, store, rows, tensors, fp8=False, bf16=True):
        self.state = None
        t0 = next((t for t in tensors if torch.is_tensor(t) and t.is_cuda), None)
        if not (knobs.accum_overlap and t0 is not None and torch.is_grad_enabled() and 0 < rows < knobs.accum_overlap_max_rows and not fp8
                and not (bf16 and store.auto_sync) and not torch.cuda.is_current_stream_capturing()):
            return
        st = _accum_state(t0.device)
        if st.last_window == 1 and st.k == 0:
            # No accumulation in the previous window (one forward per optimizer step: the retrieval loops, plain fine-tuning): the
            # stream hops then buy nothing and cost 1.4 % at 104 pairs, 3 % at 32 (profiles/r06_experiments.md section 1).  The pass
            # runs on the caller's stream but is counted, so a second forward in this window switches the overlap back on.
            st.k = 1
            return
        self.state = st
        self.store = store
        self.tensors = [t for t in tensors if torch.is_tensor(t) and t.is_cuda]

    def __enter__(self):
        st = self.state
        if st is None:
            return self
        self.store.pin_grad_accumulators()          # (on the caller's stream, before the switch: see store.py)
        self.idx = st.k & 1
        st.k += 1
        st.passes += 1
        S = st.streams[self.idx]
        self.main = torch.cuda.current_stream(st.device)
        S.wait_stream(self.main)
        for t in self.tensors:                      # allocated on the caller's stream, read by this pass (and its backward) on S
            t.record_stream(S)
        st.used[self.idx] = True
        self._ctx = torch.cuda.stream(S)
        self._ctx.__enter__()
        return self

    def mark(self, out):
        """tag the tensors of `out` that carry a graph (losses / scores) and make them usable on the caller's stream"""
        st = self.state
        if st is None:
            return out

        def one(t):
            if not torch.is_tensor(t):
                return t
            if t.requires_grad:
                t = _AccumMarker.apply(t, st, self.idx, self.main)
            if t.is_cuda:
                t.record_stream(self.main)
            return t
        if isinstance(out, (tuple, list)):
            return type(out)(one(t) for t in out)
        return one(out)

    def __exit__(self, *exc):
        st = self.state
        if st is None:
            return False
        self._ctx.__exit__(*exc)
        self.main.wait_stream(st.streams[self.idx])
        return False


def _side_route(rows):
    return knobs.wgrad_side_stream and rows >= knobs.wgrad_side_min_rows and not torch.cuda.is_current_stream_capturing()


def _on_side_stream(dev, fn, inputs):
    """run fn() on the side stream of `dev`, ordered after everything enqueued so far on the current stream; `inputs` are the
    tensors it reads.  They are kept ALIVE (a reference, not Tensor.record_stream) until the join: a block marked with
    record_stream cannot be reused before the side stream's event has completed on the device, and with the host several steps
    ahead of the GPU the caching allocator then answers every new request with a fresh hipMalloc -- round 4 measured 32-187 device
    allocations and 21-110 GB of pool growth INSIDE a 10-step timed region, and one run in eight at 146-165 ms per step instead
    of 110.  Released after the join, the blocks return to the main stream's pool in stream order: no event, no growth.
    The side stream is joined at the end of the backward pass (autograd callback), or right away outside one."""
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)                       # inputs were produced on the main stream
    if dev.index == _cur_device():
        # (torch.cuda.stream(side) as a context manager costs ~12 us of Python per use; this runs once per layer and backward pass)
        torch.cuda.set_stream(side)
        try:
            fn()
        finally:
            torch.cuda.set_stream(main)
    else:
        with torch.cuda.stream(side):
            fn()
    _side_keep.extend(t for t in inputs if t is not None)
    was_clean = not _side_dirty
    _side_dirty.add((dev.type, dev.index))
    if was_clean or not _join_queued[0]:
        # (keyed on the dirty set going non-empty, not only on the flag: a backward that raised after queueing never runs its
        #  callback, and a flag left set would keep every later backward from registering the join)
        try:
            _queue_pass_callback(_end_of_backward_join)
            _join_queued[0] = True
        except RuntimeError:                     # not inside a backward pass: join right away
            join_side_streams()
