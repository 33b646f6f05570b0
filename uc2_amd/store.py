"""Flat parameter / gradient / compute-copy arenas.

One ParamStore re-homes every parameter of a module tree into ONE contiguous fp32
buffer (the master weights the reference keeps under apex amp O2, pretrain.py:463-465),
with a same-layout fp32 gradient arena and, for bf16 compute, a bf16 shadow arena.
nn.Parameters stay ordinary tensors (views), so `state_dict()` keys and shapes are exactly
the reference's; what the arena buys on MI355X:

  * query/key/value weights (and biases) of a layer are adjacent -> the fused-QKV GEMM and
    its weight-gradient GEMM use a zero-copy [3H, H] view (model/layer.py:76-78 runs three GEMMs);
  * weight-gradient kernels accumulate straight into the arena (no per-tensor .grad allocs, no
    flatten/unflatten copies in the all-reduce: utils/distributed.py:23-42 does both every step);
  * AdamW is one launch over the arena and writes the bf16 compute copy in the same pass.
"""
import weakref

import torch

from . import _lib

_ALIGN = 64                      # elements; keeps every view 16-byte aligned in fp32 and bf16
_ROW_TILE = 256                  # large tables are followed by zero rows up to a multiple of the GEMM tile (see padded())
_PAD_MIN = 65536                 # ... when they have at least this many rows / elements (the 250 002-row vocabulary tables)
_STORES = weakref.WeakSet()
_GRAD_ACCESS = [None]            # hook run by ArenaParameter.grad (set by ops/streams.py: orders the caller's stream behind the
                                 # accumulation-overlap passes still in flight); None = nothing pending, the attribute costs one check
_tensor_grad = torch.Tensor.grad


def raw_grad(p):
    """p.grad without the ArenaParameter hook (library code that runs inside a pass or behind a join)"""
    return _tensor_grad.__get__(p, type(p))


class ArenaParameter(torch.nn.Parameter):
    """nn.Parameter re-homed into a ParamStore.  Its gradient is a view of the gradient arena that the kernels of a backward pass
    write directly -- and with the accumulation overlap (ops/streams.py::accum_pass) that pass may still be running on one of the
    two overlap streams when `backward()` returns.  Every Python-level access to `.grad` from another stream therefore first makes
    that stream wait for the passes in flight (torch.nn.utils.clip_grad_norm_, an optimizer of the caller's, logging code: no
    caller has to know).  Same storage, same class hierarchy (`isinstance(p, nn.Parameter)`), same state_dict."""

    @property
    def grad(self):
        h = _GRAD_ACCESS[0]
        if h is not None:
            h()
        return _tensor_grad.__get__(self, type(self))

    @grad.setter
    def grad(self, value):
        h = _GRAD_ACCESS[0]
        if h is not None:
            h()
        _tensor_grad.__set__(self, value)

    @grad.deleter
    def grad(self):
        _tensor_grad.__delete__(self)


def _slot_numel(p):
    """elements reserved for p in the arenas: the vocabulary-sized tables (word embeddings = tied MLM decoder weight
    [250002, H], decoder bias [250002]) get zero rows up to a multiple of 256, so that the decoder GEMMs run on whole
    256-row tiles of the arena itself (the reference disabled its own pad_vocab, model/model.py:1051-1054)"""
    if p.dim() >= 1 and p.shape[0] >= _PAD_MIN and p.shape[0] % _ROW_TILE:
        rows = (p.shape[0] + _ROW_TILE - 1) // _ROW_TILE * _ROW_TILE
        return rows * (p.numel() // p.shape[0])
    return p.numel()


def _unique_named_params(module):
    seen, out = set(), []
    for n, p in module.named_parameters():
        if id(p) not in seen:
            seen.add(id(p))
            out.append((n, p))
    return out


def _pack_groups(named):
    """reorder so that q/k/v weights, and q/k/v biases, of each attention block are adjacent"""
    by_name = dict(named)
    used, order = set(), []
    for n, p in named:
        if n in used:
            continue
        if n == "query.weight" or n.endswith(".query.weight"):
            base = n[:-len("query.weight")]
            grp_w = [base + k + ".weight" for k in ("query", "key", "value")]
            grp_b = [base + k + ".bias" for k in ("query", "key", "value")]
            if all(g in by_name for g in grp_w + grp_b):
                order.append([(g, by_name[g]) for g in grp_w])
                order.append([(g, by_name[g]) for g in grp_b])
                used.update(grp_w + grp_b)
                continue
        order.append([(n, p)])
        used.add(n)
    return order


class ParamStore:
    def __init__(self, module):
        named = _unique_named_params(module)
        assert named, "module has no parameters"
        dev = named[0][1].device
        for n, p in named:
            if p.dtype != torch.float32:
                raise _lib.Uc2Error("uc2_amd keeps fp32 master weights; parameter %s is %s "
                                    "(set compute dtype with uc2_amd.set_compute_dtype, do not .half()/.bfloat16() "
                                    "the module)" % (n, p.dtype))
            if p.device != dev:
                raise _lib.Uc2Error("all parameters of one module tree must live on one device")
        groups = _pack_groups(named)
        self.offsets, self.names, self.params = {}, [], []
        off = 0
        for grp in groups:
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            for n, p in grp:
                self.offsets[id(p)] = off
                self.names.append(n)
                self.params.append(p)
                off += _slot_numel(p)
        self.total = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        self.pos = {id(p): i for i, p in enumerate(self.params)}
        self.device = dev
        self.data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = None
        self.grad_epoch = 0
        with torch.no_grad():
            for p in self.params:
                v = self.view(self.data, p)
                v.copy_(p.data)
                p.data = v
                if type(p) is torch.nn.Parameter:
                    p.__class__ = ArenaParameter   # (reads of .grad wait for overlapped backward passes: see the class)
                p._uc2_store = self
                p._uc2_gepoch = -1
                if raw_grad(p) is not None:       # keep an existing gradient (folded into the arena)
                    self.grad_buf(p)
        self.shadow = None
        self._gviews, self._cviews, self._spans = {}, {}, {}      # cached views of the gradient / bf16 arenas (host time: see grad_buf)
        self.version = 1
        self.shadow_version = 0
        self.auto_sync = True         # re-cast the bf16 copies at every top-level forward (safe default)
        _STORES.add(self)

    # ---- views ----
    def view(self, flat, p):
        o = self.offsets[id(p)]
        return flat[o:o + p.numel()].view(p.shape)

    def padded(self, flat, p):
        """p's slot INCLUDING its zero padding rows, as [rows_padded, ...] (== view() for unpadded parameters)"""
        o = self.offsets[id(p)]
        n = _slot_numel(p)
        rows = n // (p.numel() // p.shape[0])
        return flat[o:o + n].view((rows,) + tuple(p.shape[1:]))

    def owns(self, p):
        return getattr(p, "_uc2_store", None) is self and id(p) in self.offsets and \
            p.data_ptr() == self.data.data_ptr() + 4 * self.offsets[id(p)]

    def span(self, flat, p_first, p_last, shape):
        """zero-copy view over adjacent parameters p_first..p_last (e.g. q,k,v -> [3H, H])"""
        o0 = self.offsets[id(p_first)]
        o1 = self.offsets[id(p_last)] + p_last.numel()
        return flat[o0:o1].view(shape)

    # ---- gradients ----
    def _ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            for p in self.params:
                p._uc2_gepoch = self.grad_epoch

    def grad_buf(self, p):
        """fp32 accumulation buffer for p (a view of the gradient arena), installed as p.grad.  The view objects are cached (a
        BertLayer backward asks for 16 of them, 3 us each to build: a third of its host time at the reference's micro-batch size);
        a cached view is re-validated by its address, since it has been handed out as p.grad and `p.grad.data = t` would re-point it."""
        if self.grad is None:
            self._ensure_grad()
        c = self._gviews.get(id(p))
        if c is None or c[0].data_ptr() != c[1] or c[2] is not self.grad:
            v = self.view(self.grad, p)
            self._gviews[id(p)] = (v, v.data_ptr(), self.grad)
        else:
            v = c[0]
        g = raw_grad(p)
        if g is None:
            if p._uc2_gepoch != self.grad_epoch:      # slice not known to be zero
                v.zero_()
            _tensor_grad.__set__(p, v)
            p._uc2_gepoch = -1
        elif g.data_ptr() != v.data_ptr():            # foreign gradient tensor: fold it in
            v.copy_(g)
            _tensor_grad.__set__(p, v)
            p._uc2_gepoch = -1
        return v

    def span_view(self, flat, p_first, p_last, shape):
        key = (id(p_first), id(p_last), tuple(shape), flat.dtype)
        c = self._spans.get(key)
        if c is None or c[1] is not flat or c[0].data_ptr() != c[2]:
            v = self.span(flat, p_first, p_last, shape)
            self._spans[key] = c = (v, flat, v.data_ptr())
        return c[0]

    def grad_span(self, p_first, p_last, shape):
        for p in self.params[self.pos[id(p_first)]: self.pos[id(p_last)] + 1]:
            self.grad_buf(p)
        return self.span_view(self.grad, p_first, p_last, shape)

    def zero_grad(self):
        if self.grad is not None:
            if self.grad.is_cuda:
                from . import ops
                ops.join_side_streams()
            self.grad.zero_()
        self.grad_epoch += 1
        for p in self.params:
            _tensor_grad.__set__(p, None)
            p._uc2_gepoch = self.grad_epoch

    def pin_grad_accumulators(self):
        """keep every parameter's AccumulateGrad node alive, created under the stream that is current NOW.  The kernels write
        gradients straight into the arena and the autograd Functions return None for their parameter inputs, but the graph still
        ends in those nodes, and autograd makes the stream that called backward() wait for the stream each of them was created on
        (torch/csrc/autograd/engine.cpp, exec_post_processing: leaf streams).  A node is created at a parameter's first use in a
        graph and dies with the graph -- re-created inside a pass that ops.accum_pass runs on its own stream, it would make the
        caller's stream wait for that pass's whole backward and serialise the next forward behind it.  Pinned here, on the
        caller's stream, the wait is a no-op."""
        if getattr(self, "_accs_for", None) != len(self.params):
            with torch.enable_grad():
                self._accs = [p.expand_as(p).grad_fn.next_functions[0][0] for p in self.params if p.requires_grad]
            self._accs_for = len(self.params)

    # ---- bf16 compute copies ----
    def mark_dirty(self):
        self.version += 1

    def sync_shadow(self):
        if self.shadow is None:
            self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
            self.shadow_version = 0
        if self.shadow_version != self.version:
            _lib.call("uc2_cast", 0, 1, self.total, _lib.ptr(self.data), _lib.ptr(self.shadow), _lib.stream())
            self.shadow_version = self.version

    def compute(self, p, dtype):
        if dtype == torch.float32:
            return p.data
        c = self._cviews.get(id(p))
        if c is None or c[1] is not self.shadow or c[0].data_ptr() != c[2]:
            v = self.view(self.shadow, p)
            self._cviews[id(p)] = c = (v, self.shadow, v.data_ptr())
        return c[0]

    def compute_span(self, p_first, p_last, shape, dtype):
        return self.span_view(self.data if dtype == torch.float32 else self.shadow, p_first, p_last, shape)

    # ---- k-contiguous copies W^T of the layer weights (bf16), for the input-gradient GEMMs dX = dY W ----
    def compute_t(self, p_first, p_last=None, shape=None):
        """bf16 [cols, rows] transpose of the 2-D parameter p_first (or of the adjacent parameters p_first..p_last viewed as
        `shape`, e.g. q|k|v -> [3H, H]), kept in a second bf16 arena that spans ONLY the encoder layers' slice of the parameter
        arena (170 MB for uc2-base, not the 0.55 GB the vocabulary tables would add; same relative offsets) and refreshed with
        ONE batched launch whenever the weights changed (once per optimizer step).  None if the shape is not made of whole
        64 x 64 tiles or the parameter lies outside that slice.  Call prepare_t() outside the step (warm-up / model set-up) to
        allocate the arena before the first backward needs it."""
        import ctypes
        rows, cols = (tuple(shape) if shape is not None else tuple(p_first.shape))
        if rows % 64 or cols % 64:
            return None
        self.sync_shadow()
        self.prepare_t()
        o = self.offsets[id(p_first)]
        if o < self._t_lo or o + rows * cols > self._t_hi:
            return None
        new = o not in self._t_items
        if new:
            self._t_items[o] = (rows, cols)
        if self._t_version != self.version or new:
            todo = list(self._t_items.items()) if self._t_version != self.version else [(o, (rows, cols))]

            class _Item(ctypes.Structure):
                _fields_ = [("offset", ctypes.c_size_t), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]
            arr = (_Item * len(todo))(*[_Item(oo, r, c) for oo, (r, c) in todo])
            # (the kernel addresses source and destination with the same element offset: the destination base is shifted by _t_lo)
            _lib.call("uc2_transpose_batch", len(todo), arr, _lib.ptr(self.shadow), _lib.ptr(self.shadow_t) - 2 * self._t_lo,
                      _lib.stream())
            self._t_version = self.version
        return self.shadow_t[o - self._t_lo:o - self._t_lo + rows * cols].view(cols, rows)

    # ---- head-interleaved copies of the fused QKV projections (bf16 W, W^T; fp32 bias) ----
    def qkv_interleaved(self, qw, vw, qb, nh, want_t=True):
        """(W' [3H, H] bf16, W'^T [H, 3H] bf16 or None, b' [3H] fp32): the adjacent query | key | value weights and biases with
        row w nh D + h D + d moved to row h 3D + w D + d.  The QKV GEMM on W' writes a head's q | k | v adjacent per token, which
        is what the attention kernels want to read (uc2_amd/ops/layer.py::BertLayerFn).  Copies of all registered blocks are refreshed
        by one interleave launch + one transpose launch whenever the weights changed (once per optimizer step)."""
        import ctypes
        self.sync_shadow()
        H3, cols = 3 * qw.shape[0], qw.shape[1]
        D = qw.shape[0] // nh
        if getattr(self, "_ilv_w", None) is None:
            n = sum(1 for nm in self.names if nm.endswith("query.weight"))
            self._ilv_shape = (H3, cols, nh, D)
            self._ilv_w = torch.empty(max(n, 1) * H3 * cols, dtype=torch.bfloat16, device=self.device)
            self._ilv_wt = torch.empty_like(self._ilv_w)
            self._ilv_b = torch.empty(max(n, 1) * H3, dtype=torch.float32, device=self.device)
            self._ilv_items, self._ilv_version = {}, -1
        if self._ilv_shape != (H3, cols, nh, D):
            return None
        o = self.offsets[id(qw)]
        new = o not in self._ilv_items
        if new:
            if len(self._ilv_items) * H3 * cols >= self._ilv_w.numel():
                return None
            self._ilv_items[o] = (len(self._ilv_items), self.offsets[id(qb)])
        if self._ilv_version != self.version or new:
            todo = list(self._ilv_items.items()) if self._ilv_version != self.version else [(o, self._ilv_items[o])]

            class _Ilv(ctypes.Structure):
                _fields_ = [("w_src", ctypes.c_size_t), ("w_dst", ctypes.c_size_t), ("b_src", ctypes.c_size_t), ("b_dst", ctypes.c_size_t)]

            class _Tr(ctypes.Structure):
                _fields_ = [("offset", ctypes.c_size_t), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]
            a1 = (_Ilv * len(todo))(*[_Ilv(wo, slot * H3 * cols, bo, slot * H3) for wo, (slot, bo) in todo])
            _lib.call("uc2_qkv_interleave_batch", len(todo), a1, nh, D, cols, _lib.ptr(self.shadow), _lib.ptr(self._ilv_w),
                      _lib.ptr(self.data), _lib.ptr(self._ilv_b), _lib.stream())
            if H3 % 64 == 0 and cols % 64 == 0:
                a2 = (_Tr * len(todo))(*[_Tr(slot * H3 * cols, H3, cols) for _, (slot, _) in todo])
                _lib.call("uc2_transpose_batch", len(todo), a2, _lib.ptr(self._ilv_w), _lib.ptr(self._ilv_wt), _lib.stream())
            self._ilv_version = self.version
        slot = self._ilv_items[o][0]
        w = self._ilv_w[slot * H3 * cols:(slot + 1) * H3 * cols].view(H3, cols)
        wt = self._ilv_wt[slot * H3 * cols:(slot + 1) * H3 * cols].view(cols, H3) if (want_t and H3 % 64 == 0 and cols % 64 == 0) else None
        return w, wt, self._ilv_b[slot * H3:(slot + 1) * H3]

    def table_t(self, p):
        """bf16 [cols, rows_padded] transpose of a padded vocabulary table (the word embeddings = the tied MLM decoder weight
        [250 112, H]): the k-contiguous operand of the decoder's input-gradient GEMM dz = dlogits E (contraction over the
        vocabulary), which otherwise reads E through the transposing LDS read.  Its own buffer (384 MB for uc2-base -- the encoder's
        W^T arena deliberately leaves the vocabulary tables out), allocated at first use, refreshed by one transpose launch when the
        weights changed; None when the slot is not made of whole 64 x 64 tiles."""
        import ctypes
        self.sync_shadow()
        o = self.offsets[id(p)]
        n = _slot_numel(p)
        cols = p.numel() // p.shape[0]
        rows = n // cols
        if rows % 64 or cols % 64:
            return None
        tabs = self.__dict__.setdefault("_tab_t", {})
        ent = tabs.get(o)
        if ent is None:
            ent = tabs[o] = [torch.empty(rows * cols, dtype=torch.bfloat16, device=self.device), -1]
        if ent[1] != self.version:
            class _Item(ctypes.Structure):
                _fields_ = [("offset", ctypes.c_size_t), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]
            arr = (_Item * 1)(_Item(o, rows, cols))
            # (the kernel addresses source and destination with the same element offset: the destination base is shifted by it)
            _lib.call("uc2_transpose_batch", 1, arr, _lib.ptr(self.shadow), _lib.ptr(ent[0]) - 2 * o, _lib.stream())
            ent[1] = self.version
        return ent[0].view(cols, rows)

    def prepare_t(self):
        """allocate the W^T arena: the slice of the parameter arena from the first to the last 2-D parameter of the encoder
        layers (names containing 'encoder.layer.' / 'layer.<i>.'; a bare BertLayer: all its 2-D parameters)"""
        if getattr(self, "shadow_t", None) is not None:
            return
        cand = [(self.offsets[id(p)], self.offsets[id(p)] + p.numel()) for n, p in zip(self.names, self.params)
                if p.dim() == 2 and _slot_numel(p) == p.numel() and ("encoder.layer." in n or n.startswith("layer."))]
        if not cand:
            cand = [(self.offsets[id(p)], self.offsets[id(p)] + p.numel()) for p in self.params
                    if p.dim() == 2 and _slot_numel(p) == p.numel()]
        self._t_lo = min(c[0] for c in cand) if cand else 0
        self._t_hi = max(c[1] for c in cand) if cand else 0
        self.shadow_t = torch.empty(max(self._t_hi - self._t_lo, 1), dtype=torch.bfloat16, device=self.device)
        self._t_items, self._t_version = {}, -1


def store_of(module):
    """the store that owns `module`'s parameters, built (or rebuilt after .to()/.cuda()) on demand"""
    d = module.__dict__
    st = d.get("_uc2_store_cache")
    probe = d.get("_uc2_store_probe")                  # (owner module, leaf name, parameter) of the first parameter
    if st is not None and probe is not None and probe[0]._parameters.get(probe[1]) is probe[2] and st.owns(probe[2]):
        return st                                      # (module.parameters() builds a generator chain: ~6 us per call, ~120 calls per forward)
    first = None
    for name, first in module.named_parameters():
        break
    if first is None:
        return None
    owner_path, _, leaf = name.rpartition(".")
    d["_uc2_store_probe"] = (module.get_submodule(owner_path) if owner_path else module, leaf, first)
    if st is not None and st.owns(first):
        return st
    st = getattr(first, "_uc2_store", None)
    if st is not None and st.owns(first):
        # parameters already re-homed by a parent module's store
        ps = [p for _, p in _unique_named_params(module)]
        if all(st.owns(p) for p in ps):
            module.__dict__["_uc2_store_cache"] = st
            return st
    st = ParamStore(module)
    module.__dict__["_uc2_store_cache"] = st
    return st


def mark_all_dirty():
    """call after changing parameter values outside AdamW.step (load_state_dict, broadcast, manual edits)"""
    for st in list(_STORES):
        st.mark_dirty()


def set_compute_dtype(module, dtype):
    """float32 = parity mode, bfloat16 = throughput mode (fp32 masters, bf16 MFMA GEMMs, fp32 statistics)"""
    assert dtype in (torch.float32, torch.bfloat16)
    for m in module.modules():
        m.__dict__["compute_dtype"] = dtype
    return module


def compute_dtype_of(module):
    return module.__dict__.get("compute_dtype", torch.float32)


def compute_dtype_of_store(st):
    """bfloat16 once the store keeps bf16 compute copies of its parameters (throughput mode), float32 otherwise (parity mode)"""
    return torch.bfloat16 if st.shadow is not None else torch.float32


def set_fp8(module, on=True):
    """fp8 (e4m3) operands for the forward and input-gradient GEMMs of every BertLayer under `module` (bf16 compute
    dtype required; weight gradients, attention, LayerNorm and the heads stay bf16 / fp32): BASELINE.json configs[4]"""
    for m in module.modules():
        if m.__class__.__name__ == "BertLayer":
            m.__dict__["uc2_fp8"] = bool(on)
    return module
