"""Raw wrappers of the HBM-bound kernels: LayerNorm forward / backward (+ deferred reductions), attention forward / backward,
casts.  Part of uc2_amd.ops."""
import ctypes
import math

import torch

from .. import _lib
from .._lib import call, dt, ptr, stream
from ..config import cfg as knobs
from .base import _Timed
from .streams import _on_side_stream, _queue_pass_callback, _side_route
from .gemm import _gemm_queue
from .fp8 import _fp8_cell, _fp8_hist_for, _fp8_rotate


def ln_fwd(x2, res2, gamma, beta, eps, drop_p=0.0, seed=None, seed_imm=0, want_stats=True, drop_after=False, q_key=None):
    """y = LN(dropout(x) + res) (drop_after False: the encoder's dense->dropout->LN tails) or
    y = dropout(LN(x + res)) (drop_after True: the embedding tails, model/model.py:331-333,361-363).
    q_key (fp8 mode): the tensor role of y at the GEMM that reads it -- returns a 4th value, (y8, scale) written by the same kernel
    (uc2_ln_fwd_q, delayed scaling) or None when that role has no history yet / the kernel does not take the shape"""
    M, H = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(M, dtype=torch.float32, device=x2.device) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device) if want_stats else None
    h = _fp8_hist_for(q_key, x2.device) if x2.dtype == torch.bfloat16 and H % 8 == 0 and H <= 1024 else None
    q = None
    with _Timed("ln_fwd", M * H * x2.element_size() * (3 if res2 is not None else 2)):
        if h is not None:
            y8 = torch.empty((M, H), dtype=torch.uint8, device=x2.device)
            scale = _fp8_cell(x2.device)[1]
            i_was = h[1]
            prev, nxt, clr = _fp8_rotate(h)
            rc = _lib.load().uc2_ln_fwd_q(dt(x2.dtype), M, H, ptr(x2), ptr(res2), ptr(gamma), ptr(beta), eps, drop_p, int(drop_after),
                                          ptr(seed), seed_imm, ptr(y), ptr(mean), ptr(rstd), ptr(y8), prev, nxt, clr, ptr(scale), stream())
            if rc == -2:
                h[1] = i_was
                h = None
            else:
                _lib.check(rc)
                q = (y8, scale)
        if h is None:
            call("uc2_ln_fwd", dt(x2.dtype), M, H, ptr(x2), ptr(res2), ptr(gamma), ptr(beta), eps, drop_p, int(drop_after),
                 ptr(seed), seed_imm, ptr(y), ptr(mean), ptr(rstd), stream())
    return (y, mean, rstd, q) if q_key is not None else (y, mean, rstd)


def ln_bwd(dy2, x2, res2, gamma, mean, rstd, dgamma, dbeta, drop_p=0.0, seed=None, seed_imm=0, need_dres=True,
           dbias=None, drop_after=False, q_key=None):
    """returns (dx, dres); with drop_p == 0 they are the same tensor.  dbias (optional, fp32 [H]) accumulates
    the column sum of dx: the bias gradient of the dense layer that produced x, for free in the same pass.
    Two kernels: the streaming pass (dx, dres, per-workgroup partial column sums) and a small reduction of the partials into
    dgamma / dbeta / dbias.  Nothing in the backward chain reads those three, so where the weight gradients run on the side
    stream the reduction goes there too: beside a persistent weight-gradient GEMM that owns every CU, the 5 us kernel waited
    ~115 us for a CU with the whole input-gradient chain queued behind it (24 times per step)."""
    M, H = x2.shape
    lib = _lib.load()
    ws = torch.empty(lib.uc2_ln_bwd_workspace(M, H) // 4, dtype=torch.float32, device=x2.device)
    dx = torch.empty_like(x2)
    dres = torch.empty_like(x2) if (drop_p > 0.0 and need_dres and drop_after != 1) else None       # (drop_after: False / True / 2)
    streams = 3 + (1 if res2 is not None else 0) + (1 if dres is not None else 0)     # dy, x, (res) in; dx, (dres) out
    d = dt(x2.dtype)
    h = _fp8_hist_for(q_key, x2.device) if x2.dtype == torch.bfloat16 else None
    q = None
    with _Timed("ln_bwd", M * H * x2.element_size() * streams):
        if h is not None:                  # fp8 mode: the same pass writes the e4m3 copy of dx the input-gradient GEMM reads
            d8 = torch.empty((M, H), dtype=torch.uint8, device=x2.device)
            scale = _fp8_cell(x2.device)[1]
            i_was = h[1]
            prev, nxt, clr = _fp8_rotate(h)
            rc = lib.uc2_ln_bwd_partial_q(d, M, H, ptr(dy2), ptr(x2), ptr(res2), ptr(gamma), ptr(mean), ptr(rstd), drop_p,
                                          int(drop_after), ptr(seed), seed_imm, ptr(dx), ptr(dres), int(dbias is not None), ptr(ws),
                                          ptr(d8), prev, nxt, clr, ptr(scale), stream())
            if rc == -2:
                h[1] = i_was
                h = None
            else:
                _lib.check(rc)
                q = (d8, scale)
        if h is None:
            call("uc2_ln_bwd_partial", d, M, H, ptr(dy2), ptr(x2), ptr(res2), ptr(gamma), ptr(mean), ptr(rstd), drop_p,
                 int(drop_after), ptr(seed), seed_imm, ptr(dx), ptr(dres), int(dbias is not None), ptr(ws), stream())

    _ln_bwd_second_stage(d, M, H, ws, dgamma, dbeta, dbias, x2.device)
    if q_key is not None:
        return dx, (dres if dres is not None else dx), q
    return dx, (dres if dres is not None else dx)


def _ln_bwd_second_stage(d, M, H, ws, dgamma, dbeta, dbias, device):
    """the reduction of a LayerNorm backward's partial column sums (ws) into dgamma / dbeta / dbias: on the side stream, queued for
    the end of the pass, or right away (see ln_bwd)"""
    def reduce():
        call("uc2_ln_bwd_reduce", d, M, H, ptr(ws), ptr(dgamma), ptr(dbeta), ptr(dbias), stream())
    if dgamma is not None or dbeta is not None or dbias is not None:
        if knobs.ln_reduce_side and _side_route(M):      # (below WGRAD_SIDE_MIN_ROWS tokens it made the regime erratic: 27.0-30.8 ms against 27.0-27.1)
            _on_side_stream(device, reduce, (ws,))
        elif not (knobs.ln_reduce_batch and M < knobs.wgrad_side_min_rows and _defer_ln_reduction(d, M, H, ws, dgamma, dbeta, dbias)):
            reduce()


# Small token counts (the reference's 104-pair micro-batches): the second stage of every LayerNorm backward of a pass goes out as ONE
# launch at the end of the pass (uc2_ln_bwd_reduce_batch) -- 28 launches of 5.7 us on the input-gradient chain otherwise.  The
# pending list belongs to one autograd graph task; entries a failed pass left behind are dropped, not reduced.
_LN_BATCH_MAX = 32
_ln_pending = []
_ln_pending_task = [-1]


class _LnReduceItem(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("ws", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p)]


def _defer_ln_reduction(d, M, H, ws, dgamma, dbeta, dbias):
    """queue the reduction for the end of the current backward pass; False outside one (the caller reduces right away)"""
    task = torch._C._current_graph_task_id()
    if task < 0 or torch.cuda.is_current_stream_capturing():
        return False
    if _ln_pending_task[0] != task:
        del _ln_pending[:]                       # (left by a pass that raised)
        try:
            _queue_pass_callback(flush_ln_reductions)
        except RuntimeError:
            return False
        _ln_pending_task[0] = task
    _ln_pending.append((d, H, M, ws, dgamma, dbeta, dbias))
    if len(_ln_pending) >= _LN_BATCH_MAX:
        _flush_ln(keep_task=True)
    return True


def _flush_ln(keep_task):
    items, _ln_pending[:] = list(_ln_pending), []
    if not keep_task:
        _ln_pending_task[0] = -1
    groups = {}
    for it in items:
        groups.setdefault(it[:2], []).append(it)
    for (d, H), its in groups.items():
        arr = (_LnReduceItem * len(its))(*[_LnReduceItem(M, ptr(ws), ptr(dg), ptr(dbt), ptr(dbs)) for (_, _, M, ws, dg, dbt, dbs) in its])
        call("uc2_ln_bwd_reduce_batch", d, len(its), arr, H, stream())


def flush_ln_reductions(end_of_pass=True):
    """reduce every pending LayerNorm backward now: the end-of-backward callback, and (end_of_pass False) BertLayerFn.backward
    before it hands a layer's gradients to GradSync's all-reduce hook"""
    if _ln_pending:
        _flush_ln(keep_task=not end_of_pass)
    elif end_of_pass:
        _ln_pending_task[0] = -1


ATTN_QKV_INTERLEAVED = 16          # include/uc2_hip.h UC2_ATTN_QKV_INTERLEAVED, OR-ed into `impl`


def _attn_q(q_key, qkv, impl, ilv):
    """the delayed-scaling history of the tensor role `q_key` if the attention kernel may write the e4m3 copy itself"""
    if q_key is None or not knobs.fp8_attn_fused or qkv.dtype != torch.bfloat16 or ilv or (knobs.attn_impl if impl is None else impl) == 1:
        return None
    return _fp8_hist_for(q_key, qkv.device)


def attn_fwd(qkv, mask2d, B, L, nh, D, drop_p=0.0, seed=None, seed_imm=0, impl=None, want_lse=True, ilv=False, q_key=None, rows=None):
    """ilv: qkv is [B L, nh, 3, D] (q|k|v of a head adjacent per token) instead of [B L, 3, nh, D].
    q_key (fp8 mode): the tensor role of ctx at the GEMM that reads it -- returns a third value, (ctx8, scale) written by the same
    kernel (uc2_attn_fwd_q, delayed scaling) or None when that role has no history yet / the MFMA kernels do not take the shape"""
    H = nh * D
    rows = B * L if rows is None else rows           # rows > B L: qkv / ctx carry zero-filled rows beyond the batch (BertLayerFn, padded rows)
    ctx = torch.empty((rows, H), dtype=qkv.dtype, device=qkv.device)
    if rows > B * L:
        ctx[B * L:].zero_()
    lse = torch.empty((B, nh, L), dtype=torch.float32, device=qkv.device) if want_lse else None
    h = _attn_q(q_key, qkv, impl, ilv) if rows == B * L else None
    if h is not None:
        c8 = torch.empty((B * L, H), dtype=torch.uint8, device=qkv.device)
        scale = _fp8_cell(qkv.device)[1]
        i_was = h[1]
        prev, nxt, clr = _fp8_rotate(h)
        with _Timed("attn_fwd", B * L * H * qkv.element_size() * 4 + B * nh * L * 4 + B * L * H):
            rc = _lib.load().uc2_attn_fwd_q(B, L, nh, D, ptr(qkv), ptr(mask2d), 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(lse),
                                            ptr(c8), prev, nxt, clr, ptr(scale), stream())
        if rc == 0:
            return ctx, lse, (c8, scale)
        h[1] = i_was
        if rc != -2:
            _lib.check(rc)
    with _Timed("attn_fwd", B * L * H * qkv.element_size() * 4 + B * nh * L * 4):       # q,k,v in; ctx, lse out
        call("uc2_attn_fwd", dt(qkv.dtype), (knobs.attn_impl if impl is None else impl) | (ATTN_QKV_INTERLEAVED if ilv else 0), B, L, nh, D, ptr(qkv), ptr(mask2d),
             1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(lse), stream())
    return (ctx, lse, None) if q_key is not None else (ctx, lse)


def attn_bwd(qkv, mask2d, ctx, dctx, lse, B, L, nh, D, drop_p=0.0, seed=None, seed_imm=0, impl=None, dbias=None, ilv=False, q_key=None):
    """dqkv; with dbias (fp32 [3H]) also dbias += column sums of dqkv = the gradient of the fused q|k|v bias (always in the
    reference order q | k | v).  ilv: qkv and dqkv are in the head-interleaved layout (see attn_fwd).
    q_key (fp8 mode): returns (dqkv, (dqkv8, scale) or None), the e4m3 copy written by the same kernel (uc2_attn_bwd_q)"""
    dqkv = torch.empty_like(qkv)
    if qkv.shape[0] > B * L:                         # padded rows: their gradient is exactly zero (it is contracted over tokens in dWqkv)
        dqkv[B * L:].zero_()
    h = _attn_q(q_key, qkv, impl, ilv) if qkv.shape[0] == B * L else None
    if h is not None:
        d8 = torch.empty(qkv.shape, dtype=torch.uint8, device=qkv.device)
        scale = _fp8_cell(qkv.device)[1]
        i_was = h[1]
        prev, nxt, clr = _fp8_rotate(h)
        with _Timed("attn_bwd", B * L * nh * D * qkv.element_size() * 8 + B * nh * L * 4 + B * L * nh * D * 3):
            rc = _lib.load().uc2_attn_bwd_q(B, L, nh, D, ptr(qkv), ptr(mask2d), 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx),
                                            ptr(lse), ptr(dqkv), ptr(dbias), ptr(_gemm_queue(qkv.device)[12:14]) if knobs.gemm_queue else None,
                                            ptr(d8), prev, nxt, clr, ptr(scale), stream())
        if rc == 0:
            return dqkv, (d8, scale)
        h[1] = i_was
        if rc != -2:
            _lib.check(rc)
    impl = (knobs.attn_impl if impl is None else impl) | (ATTN_QKV_INTERLEAVED if ilv else 0)
    with _Timed("attn_bwd", B * L * nh * D * qkv.element_size() * 8 + B * nh * L * 4):   # qkv, ctx, dctx, lse in; dqkv out
        if knobs.gemm_queue and qkv.dtype == torch.bfloat16:       # N > 1: the persistent kernels share the chip with the all-reduce kernels
            call("uc2_attn_bwd_queued", dt(qkv.dtype), impl, B, L, nh, D, ptr(qkv), ptr(mask2d),
                 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias),
                 ptr(_gemm_queue(qkv.device)[12:14]), stream())
        else:
            call("uc2_attn_bwd", dt(qkv.dtype), impl, B, L, nh, D, ptr(qkv), ptr(mask2d),
                 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias), stream())
    return (dqkv, None) if q_key is not None else dqkv


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("uc2_cast", dt(x.dtype), dt(dtype), x.numel(), ptr(x), ptr(out), stream())
    return out


def _mask2d(attention_mask, B, L):
    """[B,1,1,L] additive float mask (model/model.py:433-436) -> contiguous fp32 [B, L]"""
    m = attention_mask
    if m.dim() == 4:
        if m.shape[1] != 1 or m.shape[2] != 1:
            raise _lib.Uc2Error("only key masks of shape [B,1,1,L] are supported, got %s" % (tuple(m.shape),))
        m = m.reshape(m.shape[0], m.shape[3])
    if m.shape != (B, L):
        raise _lib.Uc2Error("attention mask shape %s does not match hidden states [%d,%d]" % (tuple(m.shape), B, L))
    if m.dtype != torch.float32:
        m = m.float()
    return m.contiguous()
