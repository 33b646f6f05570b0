"""Data-parallel communication with the reference's function names (utils/distributed.py), on
torch.distributed: backend "nccl" IS RCCL on ROCm (xGMI inside a node); "gloo" for CPU tests.
One process per GPU, launched by torch.distributed.run (the reference uses horovodrun -np N).

What changes relative to the reference (utils/distributed.py:15-42):
  * no flatten/unflatten: gradients already live in one contiguous fp32 arena (uc2_amd/store.py),
    so the collective runs in place on views of it;
  * bucketed and overlapped: GradSync launches one async all-reduce per encoder layer as soon as that
    layer's backward has finished (layers finish last-to-first), so only the embedding/head tail is
    exposed; RCCL picks the multi-link algorithm (7 xGMI links per GPU), nothing forces a ring;
  * the mean over ranks (Horovod's default) and `/ rescale_denom` are one scaling pass.
"""
import os

import torch
import torch.distributed as dist

from .. import _lib
from ..config import cfg, state

# The gradient tail (embeddings + heads: 0.77 GB of fp32, final only when backward ends, so its all-reduce is exposed) is reduced
# in fp32 by default: the library never rounds gradients on its own.  UC2_ALLREDUCE_TAIL=bf16 (or TAIL_BF16 = True) opts a store
# that COMPUTES in bf16 into casting its tail to bf16 for the collective -- half the bytes of the one collective nothing overlaps;
# the sum over ranks is then taken by RCCL in bf16 (8 significant bits; the reference reduces fp16 gradients, 11 bits, under a loss
# scaler: apex O2 + Horovod, pretrain.py:557-566) and cast back; the fp32 master gradients of the encoder layers are never rounded.
# `auto` = bf16 for bf16 stores (round 5's default; no run with more than one rank has compared loss curves or gradient norms of
# the two yet, one GPU per lease, so it is an opt-in again -- ADVICE r5).  The decision is taken PER STORE: a store in fp32 parity
# mode always reduces fp32.  The mode in use is logged once on rank 0.
_TAIL_MODE = cfg.allreduce_tail
TAIL_BF16 = _TAIL_MODE == "bf16"                 # (kept as a module attribute: tests and tools flip it)
_tail_logged = [False]


def _tail_bf16(st):
    """True when store `st`'s exposed tail travels as bf16: opted in (UC2_ALLREDUCE_TAIL=bf16 / auto, or TAIL_BF16) AND the store
    keeps bf16 compute copies.  A store in fp32 parity mode is never rounded."""
    if not (TAIL_BF16 or _TAIL_MODE in ("bf16", "auto")):
        return False
    from ..store import compute_dtype_of_store
    return compute_dtype_of_store(st) == torch.bfloat16


_TAIL_MIN = 1 << 20              # elements; smaller spans are not worth two cast passes


class NativeComm:
    """The library's own RCCL communicator (include/uc2_hip.h uc2_comm_*): collectives on a library-owned side stream,
    ordered against the compute stream with events.  The 128-byte unique id travels over the torch.distributed control
    group that the launcher (torch.distributed.run) set up -- any backend -- exactly once."""
    active = False
    _atexit = False

    @classmethod
    def init(cls, device):
        if cls.active:
            return True
        lib = _lib.load()
        world, rank = _world(), _rank()
        nb = lib.uc2_comm_unique_id_bytes()
        obj = [None]
        if rank == 0:
            import ctypes
            raw = (ctypes.c_char * nb)()
            try:                                             # a failure here must still reach the broadcast below, or the
                _lib.check(lib.uc2_comm_unique_id(raw, nb))   # other ranks would wait in it for ever
                obj = [list(raw.raw)]
            except Exception as e:                           # noqa: BLE001
                obj = [str(e)]
        if world > 1:
            dist.broadcast_object_list(obj, src=0)
        if not isinstance(obj[0], list):
            raise _lib.Uc2Error("rank 0 could not create the RCCL unique id: %s" % obj[0])
        idb = bytes(obj[0])
        torch.cuda.set_device(device)
        _lib.check(lib.uc2_comm_init(rank, world, idb, nb))
        cls.active = True
        if not cls._atexit:
            import atexit
            import sys
            prev_hook = sys.excepthook

            def _failed(tp, val, tb):                        # an uncaught exception ends this rank while its peers may sit in a
                cls._failing = True                          # collective: ncclCommDestroy could then block for ever at exit
                prev_hook(tp, val, tb)
            sys.excepthook = _failed
            atexit.register(cls._at_exit)                    # communicator + stream / events, once, at interpreter exit
            cls._atexit = True
        return True

    _failing = False

    @classmethod
    def mark_failed(cls):
        """call on ANY abnormal way out of the training loop (non-zero sys.exit, a watchdog, SIGTERM handler, an exception caught
        and re-raised in a worker thread): the exit handler then aborts the communicator (ncclCommAbort) instead of destroying it --
        ncclCommDestroy can block for ever while peers sit in a collective.  sys.excepthook alone only sees uncaught exceptions of
        the main thread.  Nothing is re-executed; the process still ends with the caller's exit code."""
        cls._failing = True

    @classmethod
    def _at_exit(cls):
        if cls._failing:
            cls.abort()
        else:
            cls.destroy()

    @classmethod
    def abort(cls):
        """error-path teardown: ncclCommAbort (does not wait for outstanding collectives), then the library's handles"""
        if cls.active:
            _lib.load().uc2_comm_abort()
            cls.active = False

    @staticmethod
    def world():
        """ranks the RCCL communicator itself reports (ncclCommCount); 0 when it is not up"""
        return int(_lib.load().uc2_comm_world()) if NativeComm.active else 0

    @staticmethod
    def version():
        """RCCL version string of the loaded librccl ("" when the library cannot be loaded)"""
        import ctypes
        buf = ctypes.create_string_buffer(64)
        try:
            _lib.load().uc2_comm_version(buf, 64)
        except Exception:                                    # noqa: BLE001
            return ""
        return buf.value.decode()

    @classmethod
    def destroy(cls):
        if cls.active:
            _lib.load().uc2_comm_destroy()
            cls.active = False

    @staticmethod
    def allreduce_avg(t, after=None):
        """in-place mean over ranks of a contiguous fp32 / bf16 CUDA tensor, asynchronous (the library's side stream), ordered
        after the current stream and, if given, after the torch stream `after` (the weight-gradient side stream)"""
        _lib.call("uc2_comm_allreduce_bucket_after", t.data_ptr(), t.numel(), _lib.dt(t.dtype), 1, _lib.stream(),
                  None if after is None else after.cuda_stream)

    @staticmethod
    def broadcast(t, root):
        _lib.call("uc2_comm_broadcast", t.data_ptr(), t.numel(), _lib.dt(t.dtype), int(root), _lib.stream())

    @staticmethod
    def wait():
        _lib.call("uc2_comm_wait", _lib.stream())


def _native_ok(t):
    return NativeComm.active and t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _Reducer:
    """one in-place mean-over-ranks of a flat tensor, on whichever data path is up: the library's RCCL communicator
    (mean computed by the collective) or torch.distributed (sum, scaled afterwards)"""

    def __init__(self):
        self.works, self.sum_views, self.native_used = [], [], False
        self.staged = []               # (device view, host copy) pairs of the gloo route

    def start(self, v, after=None):
        """after: a second torch stream whose work so far must be complete before the reduction reads v (the weight-gradient
        side stream of uc2_amd.ops); the current stream is always waited for"""
        if _world() == 1 and not (NativeComm.active and _native_ok(v)):
            return                                           # one rank: the mean is the value (and there may be no process group)
        if _native_ok(v):
            NativeComm.allreduce_avg(v, after)
            self.native_used = True
        elif v.is_cuda and dist.get_backend() == "gloo" and not cfg.gloo_direct:
            # gloo is the functional-test data plane (several ranks sharing the one GPU of a test box; the product path is RCCL).
            # Its own handling of device tensors (internal streams + events per collective) stopped making progress with 4
            # processes on one MI355X (every rank parked in work.wait(), round 4; 2 ranks and a bare 4-rank all-reduce of the same
            # sizes were fine): the bytes are staged through host memory here instead -- copy down (ordered after both streams),
            # reduce the host copy, copy back in finish().  What the test checks (hook order, spans, mean, replicas) is unchanged.
            if after is not None:
                torch.cuda.current_stream(v.device).wait_stream(after)
            h = v.detach().to("cpu")                               # synchronises with the current stream
            self.works.append(dist.all_reduce(h, async_op=True))
            self.staged.append((v, h))
            self.sum_views.append(v)
        elif after is not None and v.is_cuda:
            # torch.distributed orders a collective behind the stream that is current when it is issued: issue it from the side
            # stream, after that stream has been told to wait for the main one -- the main stream itself waits for nothing
            after.wait_stream(torch.cuda.current_stream(v.device))
            with torch.cuda.stream(after):
                self.works.append(dist.all_reduce(v, async_op=True))
            self.sum_views.append(v)
        else:
            self.works.append(dist.all_reduce(v, async_op=True))
            self.sum_views.append(v)

    def finish(self, extra_scale):
        """wait for everything started; apply 1/world to the summed views and `extra_scale` to all of `views`"""
        for w in self.works:
            w.wait()
        for v, h in self.staged:
            v.copy_(h)
        self.staged = []
        if self.native_used:
            NativeComm.wait()
        W = _world()
        for v in self.sum_views:
            _scale_(v, 1.0 / W)
        self.works, self.sum_views, self.native_used = [], [], False


_TAIL_STAGE = {}


def _tail_stage(device, n):
    buf = _TAIL_STAGE.get(device)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=torch.bfloat16, device=device)
        _TAIL_STAGE[device] = buf
    return buf[:n]


def _scale_(t, s):
    """t *= s in place (HIP kernel on the GPU; CPU tensors only occur in the gloo tests)"""
    if s == 1.0:
        return
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        _lib.call("uc2_scale", t.numel(), _lib.ptr(t), None, float(s), _lib.stream())
    else:
        t.mul_(s)


def _merge(ranges, gap=64):
    ranges = sorted(ranges)
    merged = []
    for o, n in ranges:
        if merged and o - (merged[-1][0] + merged[-1][1]) <= gap:
            merged[-1][1] = max(merged[-1][1], o + n - merged[-1][0])
        else:
            merged.append([o, n])
    return merged


def all_reduce_and_rescale_tensors(tensors, rescale_denom):
    """All-reduce (mean over ranks) and divide by rescale_denom, in place (utils/distributed.py:15-42).

    `tensors` is what the reference passes: [p.grad.data for p in model.parameters() if p.grad is not None].
    Views of a gradient arena are reduced in place as a few large contiguous spans; anything else is
    flattened into a temporary buffer exactly like the reference does."""
    tensors = list(tensors)
    if not tensors:
        return
    if tensors[0].is_cuda:
        from .. import ops
        ops.join_side_streams()
    W = _world()
    # find arena-backed spans by address
    from ..store import _STORES
    spans, loose = [], []
    stores = [st for st in list(_STORES) if st.grad is not None]
    for t in tensors:
        hit = None
        for st in stores:
            lo = st.grad.data_ptr()
            if t.device == st.grad.device and lo <= t.data_ptr() < lo + 4 * st.total and t.dtype == torch.float32 \
                    and t.is_contiguous():
                hit = (st, (t.data_ptr() - lo) // 4, t.numel())
                break
        if hit is None:
            loose.append(t)
        else:
            spans.append(hit)
    red = _Reducer()
    by_store = {}
    for st, o, n in spans:
        by_store.setdefault(id(st), (st, []))[1].append((o, n))
    views, staged, todo = [], [], []
    for st, ranges in by_store.values():
        sync = getattr(st, "_grad_sync", None)
        done = sync.take_done() if sync is not None else []
        for o, n in _merge(ranges):
            for (a, b) in _subtract((o, o + n), done):
                v = st.grad[a:b]
                views.append(v)
                todo.append(v)
        if sync is not None:
            sync.merge_into(red)
            views.extend(sync.take_views())
    timer = state.comm_timer          # bench.py: receives (start, end) HIP events around the EXPOSED part of the all-reduce
    if timer is not None and tensors[0].is_cuda:
        # from here on nothing of this step's compute is left to overlap with: what the stream waits for below is exposed
        e_exp0, e_exp1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e_exp0.record()
    else:
        e_exp0 = None
    if W > 1 or NativeComm.active:
        # dtype of the exposed tail, decided per owning store (a store in fp32 parity mode is never rounded because another
        # store of the process computes in bf16)
        half_ids = set()
        for st, _ in by_store.values():
            if _tail_bf16(st):
                half_ids.add(st.grad.data_ptr())
        if not _tail_logged[0] and _rank() == 0 and by_store:
            _tail_logged[0] = True
            import logging
            logging.getLogger("uc2_amd").info("gradient all-reduce: per-layer buckets fp32, embedding / head tail %s (UC2_ALLREDUCE_TAIL=%s)",
                                              "bf16" if half_ids else "fp32", _TAIL_MODE)

        def _owner_half(v):
            p = v.data_ptr()
            for st, _ in by_store.values():
                lo = st.grad.data_ptr()
                if lo <= p < lo + 4 * st.total:
                    return lo in half_ids
            return False

        def half(v):
            return bool(half_ids) and v.is_cuda and v.numel() >= _TAIL_MIN and _owner_half(v) and \
                (NativeComm.active or dist.get_backend() == "nccl")
        n_half = sum((v.numel() + 63) // 64 * 64 for v in todo if half(v))
        stage = _tail_stage(todo[0].device, n_half) if n_half else None
        off = 0
        for v in todo:
            if half(v):
                # exposed tail: fp32 -> bf16 staging (one slice per span), reduce half the bytes, back to fp32 below
                h = stage[off:off + v.numel()]
                off += (v.numel() + 63) // 64 * 64
                _lib.call("uc2_cast", 0, 1, v.numel(), _lib.ptr(v), _lib.ptr(h), _lib.stream())
                red.start(h)
                staged.append((v, h))
            else:
                red.start(v)
    buf = None
    if loose:
        sz = sum(t.numel() for t in loose)
        buf = loose[0].new_zeros(sz)
        off = 0
        for t in loose:
            buf[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
        red.start(buf)
    red.finish(1.0)
    for v, h in staged:
        _lib.call("uc2_cast", 1, 0, v.numel(), _lib.ptr(h), _lib.ptr(v), _lib.stream())
    if e_exp0 is not None:
        e_exp1.record()
        timer.append((e_exp0, e_exp1))
    scale = 1.0 / rescale_denom
    for v in views:
        _scale_(v, scale)
    if buf is not None:
        _scale_(buf, scale)
        off = 0
        for t in loose:
            t.reshape(-1).copy_(buf[off:off + t.numel()])
            off += t.numel()


def _subtract(span, done):
    """parts of [a,b) not covered by the (sorted, disjoint) intervals in `done`"""
    a, b = span
    out = []
    for (x, y) in sorted(done):
        if y <= a or x >= b:
            continue
        if x > a:
            out.append((a, x))
        a = max(a, y)
    if a < b:
        out.append((a, b))
    return out


class GradSync:
    """Overlap the gradient all-reduce with backward.  Usage (mirrors pretrain.py:557-566):

        sync = GradSync(model)                     # once
        ...
        sync.arm()                                 # before the LAST micro-step's backward of a window
        loss.backward()                            # each BertLayer's all-reduce starts as it finishes
        all_reduce_and_rescale_tensors(grads, 1)   # reduces the rest, waits, averages

    Not arming it simply gives the non-overlapped behaviour."""

    def __init__(self, model):
        from ..model.layer import BertLayer
        from ..store import store_of
        self.st = store_of(model)
        self.st._grad_sync = self
        self.layers = [m for m in model.modules() if isinstance(m, BertLayer)]
        self.armed = False
        self._red, self._views, self._done = _Reducer(), [], []
        for l in self.layers:
            l.grad_ready_hook = self._on_layer_done
        if _world() > 1:
            # collectives will run beside the backward GEMMs and hold CUs: persistent GEMMs take their items from the queue
            if cfg.gemm_queue_allowed:
                cfg.gemm_queue = True

    def arm(self):
        self.armed = True
        self._red, self._views, self._done = _Reducer(), [], []

    def will_reduce(self):
        """True when the next layer-done hook starts an all-reduce (armed, and there is somebody to reduce with)"""
        return self.armed and (_world() > 1 or NativeComm.active)

    def _on_layer_done(self, layer):
        if not self.will_reduce():
            return
        st = self.st
        ps = list(layer.parameters())
        lo = min(st.offsets[id(p)] for p in ps)
        hi = max(st.offsets[id(p)] + p.numel() for p in ps)
        v = st.grad[lo:hi]
        after = None
        if v.is_cuda:
            from .. import ops
            after = ops.pending_side_stream(v.device)          # this layer's dW GEMMs, if they went to the side stream
        self._red.start(v, after)          # 28 MB of fp32 per layer, in flight while the layers below run their backward
        self._views.append(v)
        self._done.append((lo, hi))

    def take_done(self):
        d, self._done = self._done, []
        return d

    def merge_into(self, red):
        """hand the in-flight per-layer reductions to the caller's reducer (it waits for them and finishes the mean)"""
        red.works.extend(self._red.works)
        red.sum_views.extend(self._red.sum_views)
        red.staged.extend(self._red.staged)
        red.native_used = red.native_used or self._red.native_used
        self._red = _Reducer()
        self.armed = False

    def take_views(self):
        v, self._views = self._views, []
        return v


def broadcast_tensors(tensors, root_rank, buffer_size=10485760):
    """rank `root_rank`'s values everywhere (utils/distributed.py:99-147).  Arena-backed tensors are
    broadcast in place in buffer_size-byte chunks of the arena; others one by one."""
    from ..store import mark_all_dirty
    tensors = list(tensors)
    if NativeComm.active and tensors and all(_native_ok(t) for t in tensors):
        for t in tensors:
            NativeComm.broadcast(t, root_rank)
        NativeComm.wait()
    elif _world() > 1:
        for t in tensors:
            if t.is_contiguous():
                flat = t.reshape(-1)
                step = max(1, buffer_size // t.element_size())
                for o in range(0, flat.numel(), step):
                    dist.broadcast(flat[o:o + step], root_rank)
            else:
                c = t.contiguous()
                dist.broadcast(c, root_rank)
                t.copy_(c)
    mark_all_dirty()


def all_gather_list(data):
    """gather arbitrary picklable data from every rank into a list (utils/distributed.py:175-204)"""
    if _world() == 1:
        return [data]
    out = [None] * _world()
    dist.all_gather_object(out, data)
    return out


def any_broadcast(data, root_rank):
    """broadcast arbitrary picklable data from root_rank (utils/distributed.py:207-230)"""
    if _world() == 1:
        return data
    obj = [data]
    dist.broadcast_object_list(obj, src=root_rank)
    return obj[0]
