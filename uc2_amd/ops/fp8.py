"""fp8 (e4m3) mode state: per-tensor scales, delayed-scaling histories per tensor role and task, e4m3 weight copies, the e4m3 GEMM
wrappers.  Part of uc2_amd.ops."""
import ctypes

import torch

from .. import _lib
from .._lib import call, dt, ptr, stream
from ..config import cfg as knobs, state
from .base import EPI_NONE


# --------------------------------------------------------------------------------------
# fp8 (e4m3) inputs for the forward / input-gradient GEMMs (BASELINE.json configs[4]); weight gradients stay bf16
# --------------------------------------------------------------------------------------
_FP8_CELLS = {}


def _fp8_cell(device):
    """a zeroed 4-byte amax cell + a 4-byte scale cell.  Cells come from a pool that is zero-filled once per 4096
    quantisations (a torch.zeros per tensor was one fill launch each, 384 per uc2-large step)"""
    pool = _FP8_CELLS.get(device)
    if pool is None or pool[1] >= pool[0].numel():
        pool = [torch.zeros(4096, dtype=torch.int32, device=device), 0, torch.empty(4096, dtype=torch.float32, device=device)]
        _FP8_CELLS[device] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1], pool[2][i:i + 1]


def fp8_amax(x2, amax=None):
    """amax cell (int32 bit pattern of the running maximum of |x|) of a contiguous tensor; pass `amax` to keep accumulating"""
    if amax is None:
        amax = _fp8_cell(x2.device)[0]
    assert x2.is_contiguous()
    call("uc2_fp8_amax", dt(x2.dtype), x2.numel(), ptr(x2), ptr(amax), stream())
    return amax


def fp8_quantize(x2, transpose=False, amax=None):
    """per-tensor power-of-two scaling, all on the device, two launches (amax, then scale + quantise):
    (x8 uint8 [rows, cols] or [cols, rows], scale fp32 [1]).  `amax`: a cell already holding the maximum (weights are
    quantised in both orientations from one amax pass)"""
    rows, cols = x2.shape
    assert x2.is_contiguous()
    if amax is None:
        amax = fp8_amax(x2)
    scale = _fp8_cell(x2.device)[1]
    out = torch.empty((cols, rows) if transpose else (rows, cols), dtype=torch.uint8, device=x2.device)
    call("uc2_fp8_quant_amax", dt(x2.dtype), rows, cols, ptr(x2), x2.stride(0), ptr(amax), ptr(scale), ptr(out), out.stride(0),
         int(transpose), stream())
    return out, scale
_FP8_HIST = {}             # tensor role -> [three amax cells (int32), index of the cell holding the previous maximum]
AMAX_CELLS = 16            # include/uc2_hip.h UC2_AMAX_CELLS: a maximum is kept in 16 cells (producers spread their atomics), three groups per role


def _fp8_rotate(h):
    """(previous, next, clear) device pointers of a role's three cell groups, and advance the role's history by one use"""
    cells, i = h
    base, step = cells.data_ptr(), 4 * AMAX_CELLS
    h[1] = (i + 1) % 3
    return base + i * step, base + ((i + 1) % 3) * step, base + ((i + 2) % 3) * step


_ST_UID = [0]


def _st_uid(st):
    """a number that names this parameter store for the life of the process (id() of a collected store can come back with another model)"""
    u = st.__dict__.get("_fp8_uid")
    if u is None:
        _ST_UID[0] += 1
        u = st.__dict__["_fp8_uid"] = _ST_UID[0]
    return u


_FP8_PREQ = {}             # data_ptr of a layer output -> (consumer layer id, (e4m3 copy, scale)) written by that layer's last LayerNorm for the
                           # next layer's QKV GEMM.  Cleared at the start of every top-level model forward (fp8_new_forward): an entry a
                           # forward that raised left behind must not meet the next step's tensor at a recycled address


def fp8_new_forward():
    """called by VLXLMRModel.forward before the first layer: drops a hand-over left by a forward pass that did not finish"""
    _FP8_PREQ.clear()


def fp8_quantize_act(x2, key=None):
    """e4m3 copy + scale of an activation.  key = the tensor's role (store, layer, name): the first use computes the maximum just in
    time (two passes); every later use quantises with half the scale of the PREVIOUS use's maximum while accumulating its own for the
    next one -- one pass, no amax launch (uc2_fp8_quant_delayed)."""
    if key is None or not knobs.fp8_delayed or torch.cuda.is_current_stream_capturing():
        return fp8_quantize(x2)
    h = _FP8_HIST.get(key)
    if h is None:
        cells = torch.zeros(3 * AMAX_CELLS, dtype=torch.int32, device=x2.device)
        x8, scale = fp8_quantize(x2, amax=fp8_amax(x2, cells[0:1]))
        _FP8_HIST[key] = [cells, 0]
        return x8, scale
    rows, cols = x2.shape
    assert x2.is_contiguous()
    scale = _fp8_cell(x2.device)[1]
    out = torch.empty((rows, cols), dtype=torch.uint8, device=x2.device)
    prev, nxt, clr = _fp8_rotate(h)
    call("uc2_fp8_quant_delayed", dt(x2.dtype), rows, cols, ptr(x2), x2.stride(0), prev, nxt, clr, ptr(scale), ptr(out), out.stride(0), stream())
    return out, scale


def gemm_fp8(a8, sa, b8, sb, bias=None, epi=EPI_NONE, aux_in=None, aux_out=None, flags=0):
    """bf16 C[M,N] = epi((A8 . B8^T) / (sa * sb) + bias); A8 [M,K], B8 [N,K] e4m3 bytes, K % 128 == 0"""
    M, K = a8.shape
    N = b8.shape[0]
    assert b8.shape[1] == K
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    ldaux = 0
    for x in (aux_in, aux_out):
        if x is not None and x.dim() == 2:
            ldaux = x.stride(0)
    timer = state.gemm_timer
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("uc2_gemm_fp8", M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), ptr(sa), ptr(sb), ptr(out), out.stride(0),
         ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, flags, stream())
    if timer is not None:
        e1.record()
        timer.add(("fp8", int(epi)), 2.0 * M * N * K, e0, e1)
    return out


def gemm_fp8_q(a8, sa, b8, sb, q_key, bias=None, epi=EPI_NONE, aux_in=None, aux_out=None, flags=0):
    """gemm_fp8 whose epilogue also writes the e4m3 copy of its output for the next GEMM (uc2_gemm_fp8_q; delayed scaling on the
    history of the CONSUMER's tensor role q_key).  -> (out, (q8, scale)), or None when that role has no history yet (its first use
    initialises it just in time, fp8_quantize_act) or the ping-pong kernel does not take the call."""
    h = _FP8_HIST.get(q_key) if (knobs.fp8_delayed and q_key is not None and not torch.cuda.is_current_stream_capturing()) else None
    if h is None:
        return None
    M, K = a8.shape
    N = b8.shape[0]
    if M % 256 or N % 256 or K % 256:                     # (the ping-pong kernel's shapes: do not advance the history for a call that cannot run)
        return None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    q8 = torch.empty((M, N), dtype=torch.uint8, device=a8.device)
    scale = _fp8_cell(a8.device)[1]
    ldaux = 0
    for x in (aux_in, aux_out):
        if x is not None and x.dim() == 2:
            ldaux = x.stride(0)
    timer = state.gemm_timer
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    i_was = h[1]
    prev, nxt, clr = _fp8_rotate(h)
    rc = _lib.load().uc2_gemm_fp8_q(M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), ptr(sa), ptr(sb), ptr(out), out.stride(0),
                                    ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, flags, ptr(q8), q8.stride(0), prev, nxt, clr, ptr(scale), stream())
    if rc == -2:
        h[1] = i_was
        return None
    _lib.check(rc)
    if timer is not None:
        e1.record()
        timer.add(("fp8", int(epi)), 2.0 * M * N * K, e0, e1)
    return out, (q8, scale)


class _Fp8WeightItem(ctypes.Structure):          # Uc2Fp8WeightItem
    _fields_ = [("w", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("out", ctypes.c_void_p), ("out_t", ctypes.c_void_p),
                ("amax", ctypes.c_void_p), ("scale", ctypes.c_void_p)]


# (knobs.fp8_attn_fused -- the attention kernels write the e4m3 copies of ctx / dqkv themselves, uc2_attn_fwd_q / uc2_attn_bwd_q -- is
#  OFF: measured break-even on uc2-large, config.py)
def _fp8_weight(st, p_first, p_last, shape, transpose):
    """e4m3 copy (+ scale) of a weight span, re-quantised when the parameters change (AdamW step, load_state_dict).
    Every span a forward / backward has asked for is remembered; when the store's version moves, ALL of them are quantised again,
    both orientations, by one call (uc2_fp8_quant_weights_batch: 3 launches per 32 weights instead of 3 launches per weight)."""
    cache = st.__dict__.setdefault("_fp8_cache", {})
    key = (st.offsets[id(p_first)], st.offsets[id(p_last)], bool(transpose))
    hit = cache.get(key)
    if hit is not None and hit[2] == st.version:
        return hit[0], hit[1]
    w = st.span(st.data, p_first, p_last, shape)
    rows, cols = w.shape
    if knobs.fp8_weight_batch and rows % 64 == 0 and cols % 64 == 0 and w.is_contiguous() and not torch.cuda.is_current_stream_capturing():
        spans = st.__dict__.setdefault("_fp8_spans", {})
        skey = key[:2]
        if skey not in spans:
            dev = w.device
            spans[skey] = (p_first, p_last, shape, torch.empty((rows, cols), dtype=torch.uint8, device=dev),
                           torch.empty((cols, rows), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                           torch.empty(1, dtype=torch.float32, device=dev))
            todo = [skey]                                 # a span seen for the first time: quantise it alone, now
        else:
            todo = list(spans)                            # the parameters changed: every known span in one batch
        arr = (_Fp8WeightItem * len(todo))()
        for n_, k_ in enumerate(todo):
            pf, pl, shp, w8_, wt8_, am_, sc_ = spans[k_]
            wk = st.span(st.data, pf, pl, shp)
            arr[n_] = _Fp8WeightItem(wk.data_ptr(), wk.shape[0], wk.shape[1], w8_.data_ptr(), wt8_.data_ptr(), am_.data_ptr(), sc_.data_ptr())
        rc = _lib.load().uc2_fp8_quant_weights_batch(len(todo), arr, stream())
        if rc == 0:
            for k_ in todo:
                _, _, _, w8_, wt8_, _, sc_ = spans[k_]
                cache[(k_[0], k_[1], False)] = (w8_, sc_, st.version)
                cache[(k_[0], k_[1], True)] = (wt8_, sc_, st.version)
            hit = cache[key]
            return hit[0], hit[1]
        if rc != -2:
            _lib.check(rc)
        del spans[skey]
    akey = (key[0], key[1], "amax")
    ahit = cache.get(akey)
    if ahit is None or ahit[1] != st.version:                # one amax pass serves both orientations
        ahit = (fp8_amax(w), st.version)
        cache[akey] = ahit
    w8, sc = fp8_quantize(w, transpose, amax=ahit[0])
    cache[key] = (w8, sc, st.version)
    return w8, sc


def linear_fwd_fp8(x2, st, p_first, p_last, shape, bias, epi=EPI_NONE, aux_out=None, flags=0, role=None, tag=None, pre_q=None, q_key=None):
    """pre_q: (x8, scale) already produced by the GEMM that made x2 (its fused e4m3 stream); q_key: the consumer's tensor role of THIS
    GEMM's output -- returns (y, (y8, scale) or None) then"""
    w8, sw = _fp8_weight(st, p_first, p_last, shape, False)
    x8, sx = pre_q if pre_q is not None else fp8_quantize_act(x2, None if role is None else (_st_uid(st), st.offsets[id(p_first)], "fwd", role, tag))
    if q_key is not None:
        r = gemm_fp8_q(x8, sx, w8, sw, q_key, bias=bias, epi=epi, aux_out=aux_out, flags=flags)
        return r if r is not None else (gemm_fp8(x8, sx, w8, sw, bias=bias, epi=epi, aux_out=aux_out, flags=flags), None)
    return gemm_fp8(x8, sx, w8, sw, bias=bias, epi=epi, aux_out=aux_out, flags=flags)


def linear_drop_residual_fp8(x2, st, p_first, p_last, shape, bias, res2, drop_p, seed, seed_imm, role=None, tag=None, pre_q=None):
    """-> (s, None) with s = dropout(x2 w^T + bias) + res2 from ONE e4m3 GEMM launch (uc2_gemm_fp8_drop_residual: the fp8 form of
    linear.linear_drop_residual, same mask as ln_fwd / ln_bwd), or (None, (x8, scale) or pre_q) with nothing launched when the ping-pong
    e4m3 kernel does not take the shape: the caller keeps linear_fwd_fp8 (handing it that e4m3 copy, so the activation's history
    advances once) + the LayerNorm kernel's own dropout / residual"""
    M, K = x2.shape
    N = shape[0]
    if M % 256 or N % 256 or K % 256 or x2.dtype != torch.bfloat16:
        return None, pre_q
    w8, sw = _fp8_weight(st, p_first, p_last, shape, False)
    x8, sx = pre_q if pre_q is not None else fp8_quantize_act(x2, None if role is None else (_st_uid(st), st.offsets[id(p_first)], "fwd", role, tag))
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    timer = state.gemm_timer
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.load().uc2_gemm_fp8_drop_residual(M, N, K, ptr(x8), x8.stride(0), ptr(w8), w8.stride(0), ptr(sx), ptr(sw), ptr(out), N,
                                                ptr(bias), ptr(res2), res2.stride(0), float(drop_p), ptr(seed), seed_imm, stream())
    if rc == -2:
        # (the activation was quantised above with its role's history advanced exactly as linear_fwd_fp8 would have: hand the copy on)
        return None, (x8, sx)
    _lib.check(rc)
    if timer is not None:
        e1.record()
        timer.add(("fp8", 10), 2.0 * M * N * K, e0, e1)
    return out, None


def linear_dgrad_fp8(dy2, st, p_first, p_last, shape, epi=EPI_NONE, aux_in=None, colsum_out=None, flags=0, role=None, tag=None, pre_q=None, q_key=None):
    """dX = epi(dY W): the k-contiguous operand is the transposed e4m3 copy of W ([in, out])"""
    wt8, sw = _fp8_weight(st, p_first, p_last, shape, True)
    d8, sd = pre_q if pre_q is not None else fp8_quantize_act(dy2, None if role is None else (_st_uid(st), st.offsets[id(p_first)], "bwd", role, tag))
    if q_key is not None:
        r = gemm_fp8_q(d8, sd, wt8, sw, q_key, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)
        return r if r is not None else (gemm_fp8(d8, sd, wt8, sw, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags), None)
    return gemm_fp8(d8, sd, wt8, sw, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)


def _fp8_hist_for(q_key, device):
    """the history of a tensor role if a producer may fuse its quantisation now (fp8 delayed scaling on, the role has been used)"""
    if q_key is None or not knobs.fp8_delayed or torch.cuda.is_current_stream_capturing():
        return None
    return _FP8_HIST.get(q_key)
