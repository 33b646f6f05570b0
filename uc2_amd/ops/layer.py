"""BertLayerFn: one autograd node per encoder layer (model/layer.py:159-170 of the reference), per-kernel route and native
per-layer entry.  Part of uc2_amd.ops."""
import ctypes

import torch

from .. import _lib
from .._lib import dt, ptr, stream
from ..config import cfg as knobs, state
from ..store import store_of
from .base import EPI_ADD, EPI_DGELU, EPI_GELU, EPI_NONE, GEMM_AUX_DERIV, _FORCED, rng
from .streams import _on_side_stream, _side_route
from .gemm import _gemm_queue, _plan_fits, gemm, gemm_plan
from .fp8 import _FP8_PREQ, _fp8_weight, _st_uid, fp8_quantize_act, gemm_fp8, linear_dgrad_fp8, linear_drop_residual_fp8, linear_fwd_fp8
from .linear import _num_cus, linear_dgrad, linear_drop_residual, linear_fwd, linear_wgrad, wgrad_group
from .kernels import _ln_bwd_second_stage, attn_bwd, attn_fwd, flush_ln_reductions, ln_bwd, ln_fwd


# --------------------------------------------------------------------------------------
# one BertLayer = one autograd node (reference model/layer.py:159-170)
# --------------------------------------------------------------------------------------
_P_NAMES = ("qw", "qb", "kw", "kb", "vw", "vb", "ow", "ob", "g1", "b1", "iw", "ib", "fw", "fb", "g2", "b2")


def layer_params(layer):
    a, it, o = layer.attention, layer.intermediate, layer.output
    s = a.self
    return (s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
            a.output.dense.weight, a.output.dense.bias, a.output.LayerNorm.weight, a.output.LayerNorm.bias,
            it.dense.weight, it.dense.bias, o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)


def _ilv_wgrad_plan(n_out, n_in, rows, device):
    """(variant, split_k) of the two-stage ping-pong weight-gradient GEMM dW[n_out, n_in] += dY^T X over `rows` tokens, or None if
    that kernel cannot take the shape.  The interleaved route needs this path: its reduction pass is what puts the rows of dWqkv
    back into the parameter arena's order."""
    key = (True, True, n_out, n_in, rows, True)
    v, sp = gemm_plan(torch.bfloat16, True, True, n_out, n_in, rows, True)
    if v in (8, 12) and sp > 1 and _plan_fits((v, sp), key):
        return v, sp
    tiles = (n_out // 256) * (n_in // 256)
    if n_out % 256 or n_in % 256 or tiles == 0:
        return None
    valid = [s_ for s_ in range(2, 129) if s_ * 256 <= rows and _plan_fits((12, s_), key)]
    if not valid:
        return None
    cus = _num_cus(device)
    return 12, min(valid, key=lambda s_: (abs(tiles * s_ - cus), s_))


# One C call per layer and direction (include/uc2_hip.h: uc2_bert_layer_fwd / _bwd) for the plain route -- no fp8, no head-interleaved
# q|k|v, no fused dropout-residual tails, no k-contiguous W^T copies, i.e. the reference's micro-batch sizes: the same kernels with
# the same arguments in the same order as the per-kernel calls of BertLayerFn below, so the same bits; what it saves is host time
# (~20 ctypes calls, their argument marshalling and the timing hooks per layer).  knobs.native_layer; _native_layer_ok says when.
class _GemmPlanC(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int), ("split_k", ctypes.c_int), ("flags", ctypes.c_int)]


_VP, _CI, _CF, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64


class _BertLayerC(ctypes.Structure):            # Uc2BertLayer
    _fields_ = ([(n, _CI) for n in ("dtype", "B", "L", "H", "nh", "I", "attn_impl")] + [(n, _CF) for n in ("eps", "p_hidden", "p_attn")]
                + [("seed", _VP)] + [(n, _U64) for n in ("site_attn", "site_ln1", "site_ln2")]
                + [(n, _VP) for n in ("wqkv", "wo", "wi", "wf", "bqkv", "bo", "g1", "b1", "bi", "bf", "g2", "b2", "mask", "x", "qkv", "ctx",
                                      "lse", "o1", "mean1", "rstd1", "a", "pre", "u", "o2", "mean2", "rstd2", "y")]
                + [(n, _GemmPlanC) for n in ("plan_qkv", "plan_o", "plan_i", "plan_f")] + [("queue", _VP)])


class _BertLayerGradC(ctypes.Structure):        # Uc2BertLayerGrad
    _fields_ = ([(n, _VP) for n in ("dy", "d_o2", "dz2", "d_pre", "da", "d_o1", "dz1", "dctx", "dqkv", "dx", "ws1", "ws2", "dbi", "dbqkv",
                                    "attn_queue")]
                + [(n, _GemmPlanC) for n in ("plan_df", "plan_di", "plan_do", "plan_dqkv")])


def _plan_c(dtype, tb, M, N, K, epi=EPI_NONE, flags=0):
    """the (variant, split_k, flags) _gemm_planned + gemm would pass for this GEMM"""
    v, sp = gemm_plan(dtype, False, tb, M, N, K, False)
    if knobs.pp_skew and v in (8, 9, 12):
        flags |= (knobs.pp_skew.get(epi, 0) & 15) << 4
    return _GemmPlanC(v, sp, flags | knobs.gemm_extra_flags)


def _native_layer_ok(dtype, M, fp8, ilv):
    return (knobs.native_layer and not fp8 and ilv is None and state.gemm_timer is None and state.hbm_timer is None and _FORCED[0] is None
            and not (knobs.ln_fuse and dtype == torch.bfloat16 and M >= knobs.ln_fuse_min_rows)
            and not (knobs.dgrad_transposed_w and dtype == torch.bfloat16 and M >= knobs.dgrad_wt_min_rows))


class BertLayerFn(torch.autograd.Function):
    """x -> LN(x + Wo.Attn(x)) -> LN(a + W2.gelu(W1.a)).  10 kernel launches forward, 21 backward.
    Weight/bias/LN gradients are accumulated by the kernels directly into the fp32 gradient arena
    (installed as .grad); autograd only carries dx."""

    @staticmethod
    def forward(ctx, x, mask2d, layer, cfg, *params):
        st = store_of(layer)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        P = dict(zip(_P_NAMES, params))
        # x is [B, L, H] or -- inside VLXLMREncoder when B L is not a whole number of 256-row GEMM tiles -- [rows, H] with
        # rows = B L rounded up to 256 (functions.PadRowsFn): every GEMM / LayerNorm of the layer then runs over `rows` token rows (the
        # planned ping-pong kernels instead of the generic kernel a ragged token count falls back to), attention over the B x L
        # structure of the mask.  Rows beyond B L hold finite junk in the forward (bias, LayerNorm beta) and exact zeros in every
        # gradient that is contracted over tokens (the attention tails are zero-filled), so they add nothing to any weight gradient.
        rows2d = x.dim() == 2
        if rows2d:
            (M, H), (B, L) = x.shape, mask2d.shape
            assert M >= B * L
        else:
            B, L, H = x.shape
            M = B * L
        nh = cfg["nh"]
        D = H // nh
        x2 = x.reshape(M, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        training = cfg["training"]
        p_h = cfg["p_hidden"] if training else 0.0
        p_a = cfg["p_attn"] if training else 0.0
        seed = rng.snapshot(x.device) if (p_h > 0 or p_a > 0) else None
        sid = cfg["layer_id"] * 16
        s_attn, s_ln1, s_ln2 = rng.site(sid + 1), rng.site(sid + 2), rng.site(sid + 3)      # dropout sites of this layer

        wqkv = st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype)
        bqkv = st.span_view(st.data, P["qb"], P["vb"], (3 * H,))
        # Head-interleaved q|k|v (round 4): the QKV GEMM runs on a row-permuted copy of [Wq; Wk; Wv] (store.qkv_interleaved), so a
        # head's q | k | v is ONE 384-byte segment per token instead of three 128-byte segments 1536 bytes apart -- the attention
        # kernels' access pattern is what bounds them (forward 4.2 -> 4.7 TB/s).  Same dot products in the same order: qkv, dqkv,
        # ctx and every gradient are bit-identical to the plain layout.  Needs the MFMA attention kernels and, for the backward,
        # the two-stage ping-pong weight-gradient GEMM (its reduction pass un-permutes the rows of dWqkv).
        ilv = None
        if (knobs.qkv_interleaved and dtype == torch.bfloat16 and M >= knobs.qkv_ilv_min_rows and M % 128 == 0 and not cfg.get("fp8")
                and knobs.attn_impl in (0, 2) and D in (32, 64) and _lib.load().uc2_attn_mfma_supported(L, D)
                and not torch.cuda.is_current_stream_capturing()):
            plan = _ilv_wgrad_plan(3 * H, H, M, x.device) if any(ctx.needs_input_grad) or training else (12, 2)
            pack = st.qkv_interleaved(P["qw"], P["vw"], P["qb"], nh) if plan is not None else None
            if pack is not None:
                ilv = (pack, plan)
                wqkv, bqkv = pack[0], pack[2]
        if not any(ctx.needs_input_grad):
            # forward-only (retrieval scoring, validation, the hard-negative scoring pass): nothing is kept for a
            # backward -- no gelu' stream out of the FFN1 GEMM, no LayerNorm statistics, no log-sum-exp
            qkv = linear_fwd(x2, wqkv, bqkv)
            ctxv, _ = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, want_lse=False, ilv=ilv is not None, rows=M)
            del qkv
            fuse = int(knobs.ln_fuse) if (dtype == torch.bfloat16 and M >= knobs.ln_fuse_min_rows) else 0
            o1 = linear_drop_residual(ctxv, st.compute(P["ow"], dtype), P["ob"].data, x2, p_h, seed, s_ln1) if fuse & 1 else None
            if o1 is not None:
                a, _, _ = ln_fwd(o1, None, P["g1"].data, P["b1"].data, 1e-12, want_stats=False)
            else:
                o1 = linear_fwd(ctxv, st.compute(P["ow"], dtype), P["ob"].data)
                a, _, _ = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1, want_stats=False)
            del o1, ctxv
            u = linear_fwd(a, st.compute(P["iw"], dtype), P["ib"].data, EPI_GELU, None)
            o2 = linear_drop_residual(u, st.compute(P["fw"], dtype), P["fb"].data, a, p_h, seed, s_ln2) if fuse & 2 else None
            if o2 is not None:
                y, _, _ = ln_fwd(o2, None, P["g2"].data, P["b2"].data, 1e-12, want_stats=False)
            else:
                o2 = linear_fwd(u, st.compute(P["fw"], dtype), P["fb"].data)
                y, _, _ = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2, want_stats=False)
            del u
            return y if rows2d else y.view(B, L, H)
        fp8 = bool(cfg.get("fp8")) and dtype == torch.bfloat16 and H % 128 == 0 and P["iw"].shape[0] % 128 == 0
        I_ = P["iw"].shape[0]
        pre = torch.empty((M, I_), dtype=dtype, device=x.device)
        native = _native_layer_ok(dtype, M, fp8, ilv) and M == B * L
        if native:
            dev = x.device
            e = lambda *shape, dt_=dtype: torch.empty(shape, dtype=dt_, device=dev)
            f32 = torch.float32
            qkv, ctxv, lse = e(M, 3 * H), e(M, H), e(B, nh, L, dt_=f32)
            o1, mean1, rstd1, a = e(M, H), e(M, dt_=f32), e(M, dt_=f32), e(M, H)
            u, o2, mean2, rstd2, y = e(M, I_), e(M, H), e(M, dt_=f32), e(M, dt_=f32), e(M, H)
            c = _BertLayerC()
            c.dtype, c.B, c.L, c.H, c.nh, c.I, c.attn_impl = dt(dtype), B, L, H, nh, I_, knobs.attn_impl
            c.eps, c.p_hidden, c.p_attn = 1e-12, p_h, p_a
            c.seed, c.site_attn, c.site_ln1, c.site_ln2 = ptr(seed), s_attn, s_ln1, s_ln2
            c.wqkv, c.wo, c.wi, c.wf = wqkv.data_ptr(), st.compute(P["ow"], dtype).data_ptr(), st.compute(P["iw"], dtype).data_ptr(), st.compute(P["fw"], dtype).data_ptr()
            c.bqkv, c.bo, c.g1, c.b1 = bqkv.data_ptr(), P["ob"].data.data_ptr(), P["g1"].data.data_ptr(), P["b1"].data.data_ptr()
            c.bi, c.bf, c.g2, c.b2 = P["ib"].data.data_ptr(), P["fb"].data.data_ptr(), P["g2"].data.data_ptr(), P["b2"].data.data_ptr()
            c.mask, c.x = mask2d.data_ptr(), x2.data_ptr()
            c.qkv, c.ctx, c.lse, c.o1, c.mean1, c.rstd1, c.a = (qkv.data_ptr(), ctxv.data_ptr(), lse.data_ptr(), o1.data_ptr(), mean1.data_ptr(),
                                                                 rstd1.data_ptr(), a.data_ptr())
            c.pre, c.u, c.o2, c.mean2, c.rstd2, c.y = pre.data_ptr(), u.data_ptr(), o2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), y.data_ptr()
            c.plan_qkv, c.plan_o = _plan_c(dtype, False, M, 3 * H, H), _plan_c(dtype, False, M, H, H)
            c.plan_i, c.plan_f = _plan_c(dtype, False, M, I_, H, EPI_GELU, GEMM_AUX_DERIV), _plan_c(dtype, False, M, H, I_)
            c.queue = ptr(_gemm_queue(dev)) if (knobs.gemm_queue and dtype == torch.bfloat16) else None
            _lib.check(_lib.load().uc2_bert_layer_fwd(ctypes.byref(c), stream()))
            ctx.native_c = c
            fused1 = fused2 = False
        elif fp8:
            # e4m3 operands for the four forward GEMMs (per-tensor scales computed on the device), bf16 outputs
            # tensor roles (keys of the delayed-scaling histories): a role is named by its CONSUMER; the layer input's by the layer id,
            # so that the layer above can write the e4m3 copy from its last LayerNorm (handed over through _FP8_PREQ)
            kx = (_st_uid(st), ("layer", cfg["layer_id"]), "fwd", "x", state.fp8_tag)
            ky = (_st_uid(st), ("layer", cfg["layer_id"] + 1), "fwd", "x", state.fp8_tag)
            ka = (_st_uid(st), st.offsets[id(P["iw"])], "fwd", "a", state.fp8_tag)
            xq = _FP8_PREQ.pop(x2.data_ptr(), None)
            if xq is not None:                         # written for THIS layer (id) by the layer below, same shape -- or not used
                xq = xq[1] if (xq[0] == cfg["layer_id"] and tuple(xq[1][0].shape) == (M, H)) else None
            w8_, sw_ = _fp8_weight(st, P["qw"], P["vw"], (3 * H, H), False)
            x8_, sx_ = xq if xq is not None else fp8_quantize_act(x2, kx)
            qkv = gemm_fp8(x8_, sx_, w8_, sw_, bias=bqkv)
            # (the attention kernel writes the e4m3 copy of ctx the output projection reads; same role key as the stand-alone pass)
            ctxv, lse, cq = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, q_key=(_st_uid(st), st.offsets[id(P["ow"])], "fwd", "ctx", state.fp8_tag), rows=M)
            # the dense -> dropout -> + residual tails as in the bf16 branch below: from LN_FUSE_MIN_ROWS tokens the e4m3 GEMM writes the
            # pre-LayerNorm sum (uc2_gemm_fp8_drop_residual), the LayerNorm reads one tensor and hashes no mask
            fuse = int(knobs.ln_fuse) if M >= knobs.ln_fuse_min_rows else 0
            o1, cq = (linear_drop_residual_fp8(ctxv, st, P["ow"], P["ow"], (H, H), P["ob"].data, x2, p_h, seed, s_ln1, role="ctx", tag=state.fp8_tag,
                                               pre_q=cq) if fuse & 1 else (None, cq))
            fused1 = o1 is not None
            if fused1:
                a, mean1, rstd1, aq = ln_fwd(o1, None, P["g1"].data, P["b1"].data, 1e-12, q_key=ka)
            else:
                o1 = linear_fwd_fp8(ctxv, st, P["ow"], P["ow"], (H, H), P["ob"].data, role="ctx", tag=state.fp8_tag, pre_q=cq)
                a, mean1, rstd1, aq = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1, q_key=ka)
            # (the FFN1 GEMM's epilogue writes the e4m3 copy of u that FFN2 reads: no quantisation pass over [tokens, 4H])
            u, uq = linear_fwd_fp8(a, st, P["iw"], P["iw"], (I_, H), P["ib"].data, EPI_GELU, pre, flags=GEMM_AUX_DERIV, role="a", tag=state.fp8_tag,
                                   pre_q=aq, q_key=(_st_uid(st), st.offsets[id(P["fw"])], "fwd", "u", state.fp8_tag))
            o2, uq = (linear_drop_residual_fp8(u, st, P["fw"], P["fw"], (H, I_), P["fb"].data, a, p_h, seed, s_ln2, role="u", tag=state.fp8_tag,
                                               pre_q=uq) if fuse & 2 else (None, uq))
            fused2 = o2 is not None
            if not fused2:
                o2 = linear_fwd_fp8(u, st, P["fw"], P["fw"], (H, I_), P["fb"].data, role="u", tag=state.fp8_tag, pre_q=uq)
        else:
            qkv = linear_fwd(x2, wqkv, bqkv)
            ctxv, lse = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, ilv=ilv is not None, rows=M)
            # The dense -> dropout -> + residual tails: with LN_FUSE the Wo / FFN2 GEMM writes the pre-LayerNorm SUM (dropout mask and
            # residual in its epilogue), the LayerNorm reads one tensor and hashes no mask; o1 / o2 then hold the sums and the
            # backward runs the LayerNorm in its drop_after = 2 form (fused1 / fused2 say which form each tail took)
            fuse = int(knobs.ln_fuse) if M >= knobs.ln_fuse_min_rows else 0
            o1 = linear_drop_residual(ctxv, st.compute(P["ow"], dtype), P["ob"].data, x2, p_h, seed, s_ln1) if fuse & 1 else None
            fused1 = o1 is not None
            if fused1:
                a, mean1, rstd1 = ln_fwd(o1, None, P["g1"].data, P["b1"].data, 1e-12)
            else:
                o1 = linear_fwd(ctxv, st.compute(P["ow"], dtype), P["ob"].data)
                a, mean1, rstd1 = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1)
            # `pre` holds gelu'(a W1^T + b1), not the pre-activation itself (UC2_GEMM_AUX_DERIV): one more exp2 beside
            # the forward's Phi(x) there, and the backward's dGELU epilogue becomes a plain multiply
            u = linear_fwd(a, st.compute(P["iw"], dtype), P["ib"].data, EPI_GELU, pre, flags=GEMM_AUX_DERIV)
            o2 = linear_drop_residual(u, st.compute(P["fw"], dtype), P["fb"].data, a, p_h, seed, s_ln2) if fuse & 2 else None
            fused2 = o2 is not None
            if not fused2:
                o2 = linear_fwd(u, st.compute(P["fw"], dtype), P["fb"].data)
        if native:
            pass
        elif fp8:
            if fused2:
                y, mean2, rstd2, yq = ln_fwd(o2, None, P["g2"].data, P["b2"].data, 1e-12, q_key=ky)
            else:
                y, mean2, rstd2, yq = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2, q_key=ky)
            _FP8_PREQ.clear()                          # (at most one hand-over alive: the last layer's copy has no fp8 consumer)
            if yq is not None:
                _FP8_PREQ[y.data_ptr()] = (cfg["layer_id"] + 1, yq)
        elif fused2:
            y, mean2, rstd2 = ln_fwd(o2, None, P["g2"].data, P["b2"].data, 1e-12)
        else:
            y, mean2, rstd2 = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2)
        ctx.ln_fused = (fused1, fused2)

        ctx.save_for_backward(x2, mask2d, qkv, ctxv, lse, o1, mean1, rstd1, a, pre, u, o2, mean2, rstd2, seed)
        ctx.layer, ctx.cfg, ctx.shape, ctx.p = layer, cfg, (B, L, H, nh, D), (p_h, p_a, (s_attn, s_ln1, s_ln2))
        ctx.params, ctx.fp8, ctx.fp8_tag = params, fp8, state.fp8_tag
        ctx.native = native
        ctx.rows2d = rows2d
        ctx.ilv_plan = ilv[1] if ilv is not None else None
        return y if rows2d else y.view(B, L, H)

    @staticmethod
    def backward(ctx, dy):
        x2, mask2d, qkv, ctxv, lse, o1, mean1, rstd1, a, pre, u, o2, mean2, rstd2, seed = ctx.saved_tensors
        B, L, H, nh, D = ctx.shape
        p_h, p_a, (s_attn, s_ln1, s_ln2) = ctx.p
        P = dict(zip(_P_NAMES, ctx.params))
        st = store_of(ctx.layer)
        dtype = x2.dtype
        M = x2.shape[0]                         # B L, or B L rounded up to 256 rows (see forward)
        rows2d = ctx.rows2d
        out_shape = (M, H) if rows2d else (B, L, H)
        dy2 = dy.reshape(M, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        G = st.grad_buf

        # LN2 and FFN
        fp8 = ctx.fp8
        I_ = P["iw"].shape[0]
        nat = None
        if ctx.native:
            # everything of this layer's backward except the weight gradients and the LayerNorm reductions: one C call
            dev = x2.device
            e = lambda *shape: torch.empty(shape, dtype=dtype, device=dev)
            lib = _lib.load()
            nws = lib.uc2_ln_bwd_workspace(M, H) // 4
            ws1, ws2 = torch.empty(nws, dtype=torch.float32, device=dev), torch.empty(nws, dtype=torch.float32, device=dev)
            d_o2, d_pre, da, d_o1, dctx, dqkv = e(M, H), e(M, I_), e(M, H), e(M, H), e(M, H), e(M, 3 * H)
            dz2, dz1 = (e(M, H), e(M, H)) if p_h > 0.0 else (None, None)
            dxn = e(M, H) if ctx.needs_input_grad[0] else None
            g = _BertLayerGradC()
            g.dy, g.d_o2, g.dz2, g.d_pre, g.da = dy2.data_ptr(), d_o2.data_ptr(), ptr(dz2), d_pre.data_ptr(), da.data_ptr()
            g.d_o1, g.dz1, g.dctx, g.dqkv, g.dx = d_o1.data_ptr(), ptr(dz1), dctx.data_ptr(), dqkv.data_ptr(), ptr(dxn)
            g.ws1, g.ws2 = ws1.data_ptr(), ws2.data_ptr()
            g.dbi = G(P["ib"]).data_ptr()
            g.dbqkv = st.grad_span(P["qb"], P["vb"], (3 * H,)).data_ptr()
            g.attn_queue = ptr(_gemm_queue(dev)[12:14]) if (knobs.gemm_queue and dtype == torch.bfloat16) else None
            g.plan_df = _plan_c(dtype, True, M, I_, H, EPI_DGELU, GEMM_AUX_DERIV)
            g.plan_di, g.plan_do, g.plan_dqkv = _plan_c(dtype, True, M, H, I_, EPI_ADD), _plan_c(dtype, True, M, H, H), _plan_c(dtype, True, M, H, 3 * H, EPI_ADD)
            c = ctx.native_c
            # (the parameter pointers are taken again here, like the per-kernel route does: a store re-created between forward and
            #  backward -- set_compute_dtype, load_state_dict into a new arena -- must not leave this call with stale addresses)
            c.wqkv = st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype).data_ptr()
            c.wo, c.wi, c.wf = st.compute(P["ow"], dtype).data_ptr(), st.compute(P["iw"], dtype).data_ptr(), st.compute(P["fw"], dtype).data_ptr()
            c.g1, c.g2 = P["g1"].data.data_ptr(), P["g2"].data.data_ptr()
            _lib.check(lib.uc2_bert_layer_bwd(ctypes.byref(c), ctypes.byref(g), stream()))
            d_ = dt(dtype)
            _ln_bwd_second_stage(d_, M, H, ws2, G(P["g2"]), G(P["b2"]), G(P["fb"]), dev)
            _ln_bwd_second_stage(d_, M, H, ws1, G(P["g1"]), G(P["b1"]), G(P["ob"]), dev)
            nat = (d_o2, d_pre, d_o1, dqkv, dxn)
        if nat is not None:
            pass
        elif fp8:
            d_o2, dz2, dq2 = ln_bwd(dy2, o2, None if ctx.ln_fused[1] else a, P["g2"].data, mean2, rstd2, G(P["g2"]), G(P["b2"]), p_h, seed, s_ln2,
                                    dbias=G(P["fb"]), drop_after=2 if ctx.ln_fused[1] else False,
                                    q_key=(_st_uid(st), st.offsets[id(P["fw"])], "bwd", "d_o2", ctx.fp8_tag))
        else:
            d_o2, dz2 = ln_bwd(dy2, o2, None if ctx.ln_fused[1] else a, P["g2"].data, mean2, rstd2, G(P["g2"]), G(P["b2"]), p_h, seed, s_ln2,
                               dbias=G(P["fb"]), drop_after=2 if ctx.ln_fused[1] else False)
        I_ = P["iw"].shape[0]
        # k-contiguous copies W^T for the input-gradient GEMMs (bf16; refreshed once per optimizer step, one launch for all)
        # (from DGRAD_WT_MIN_ROWS tokens)
        use_wt = knobs.dgrad_transposed_w and dtype == torch.bfloat16 and not fp8 and M >= knobs.dgrad_wt_min_rows
        WT = (lambda pf, pl=None, shp=None: st.compute_t(pf, pl, shp)) if use_wt else (lambda *a_: None)
        # small token counts: the four weight gradients go out as ONE grouped launch at the end (wgrad_group)
        grouped = [] if (knobs.wgrad_group and dtype == torch.bfloat16 and M < knobs.wgrad_side_min_rows and M % 128 == 0) else None
        wgrad = (lambda dyv, xv, dwv: grouped.append((dyv, xv, dwv))) if grouped is not None else (lambda dyv, xv, dwv: linear_wgrad(dyv, xv, dwv, None))
        wgrad(d_o2, u, G(P["fw"]))
        if nat is not None:
            pass
        elif fp8:
            d_pre, dq = linear_dgrad_fp8(d_o2, st, P["fw"], P["fw"], (H, I_), EPI_DGELU, pre, colsum_out=G(P["ib"]), flags=GEMM_AUX_DERIV, role="d_o2",
                                         tag=ctx.fp8_tag, pre_q=dq2, q_key=(_st_uid(st), st.offsets[id(P["iw"])], "bwd", "d_pre", ctx.fp8_tag))
        else:
            d_pre = linear_dgrad(d_o2, st.compute(P["fw"], dtype), EPI_DGELU, pre, colsum_out=G(P["ib"]),
                                 flags=GEMM_AUX_DERIV, wt=WT(P["fw"]))                         # + d(intermediate bias)
        wgrad(d_pre, a, G(P["iw"]))
        if nat is not None:
            pass
        elif fp8:
            da = linear_dgrad_fp8(d_pre, st, P["iw"], P["iw"], (I_, H), EPI_ADD, dz2, role="d_pre", tag=ctx.fp8_tag, pre_q=dq)
        else:
            da = linear_dgrad(d_pre, st.compute(P["iw"], dtype), EPI_ADD, dz2, wt=WT(P["iw"]))
        # LN1, output projection, attention, fused QKV
        if nat is not None:
            pass
        elif fp8:
            d_o1, dz1, dq1 = ln_bwd(da, o1, None if ctx.ln_fused[0] else x2, P["g1"].data, mean1, rstd1, G(P["g1"]), G(P["b1"]), p_h, seed, s_ln1,
                                    dbias=G(P["ob"]), drop_after=2 if ctx.ln_fused[0] else False,
                                    q_key=(_st_uid(st), st.offsets[id(P["ow"])], "bwd", "d_o1", ctx.fp8_tag))
        else:
            d_o1, dz1 = ln_bwd(da, o1, None if ctx.ln_fused[0] else x2, P["g1"].data, mean1, rstd1, G(P["g1"]), G(P["b1"]), p_h, seed, s_ln1,
                               dbias=G(P["ob"]), drop_after=2 if ctx.ln_fused[0] else False)
            dq1 = None
        wgrad(d_o1, ctxv, G(P["ow"]))
        dwqkv = st.grad_span(P["qw"], P["vw"], (3 * H, H))
        if nat is not None:
            wgrad(dqkv, x2, dwqkv)
            return BertLayerFn._finish_backward(ctx, grouped, dy2, None if dxn is None else dxn.view(out_shape))
        dctx = linear_dgrad_fp8(d_o1, st, P["ow"], P["ow"], (H, H), role="d_o1", tag=ctx.fp8_tag, pre_q=dq1) if fp8 else linear_dgrad(d_o1, st.compute(P["ow"], dtype), wt=WT(P["ow"]))
        dwqkv = st.grad_span(P["qw"], P["vw"], (3 * H, H))
        dbqkv = st.grad_span(P["qb"], P["vb"], (3 * H,))
        # d(q|k|v bias) comes out of attn_bwd: column sums of the dQ/dK/dV accumulators, added up per workgroup in LDS and
        # flushed with one global atomic per column and workgroup (the separate column-sum pass re-read dqkv: 97 us)
        # (the fp32-math kernels run the column-sum pass inside uc2_attn_bwd)
        ilv_plan = ctx.ilv_plan
        dqq = None
        if fp8:
            dqkv, dqq = attn_bwd(qkv, mask2d, ctxv, dctx, lse, B, L, nh, D, p_a, seed, s_attn, dbias=dbqkv,
                                 q_key=(_st_uid(st), st.offsets[id(P["qw"])], "bwd", "dqkv", ctx.fp8_tag))
        else:
            dqkv = attn_bwd(qkv, mask2d, ctxv, dctx, lse, B, L, nh, D, p_a, seed, s_attn, dbias=dbqkv, ilv=ilv_plan is not None)
        if ilv_plan is not None:
            # dqkv is head-interleaved: dWqkv comes out with its rows in that order, the split-K reduction puts them back
            # (dqkv^T x on the two-stage ping-pong kernel: _ilv_wgrad_plan); the input gradient contracts over the interleaved
            # index on both operands (W' / W'^T)
            v_, sp_ = ilv_plan
            w_ilv, wt_ilv, _ = st.qkv_interleaved(P["qw"], P["vw"], P["qb"], nh)

            def _wg():
                gemm(dqkv, x2, 3 * H, H, M, ta=True, tb=True, out=dwqkv, accumulate=True, split_k=sp_, variant=v_, qkv_rows_d=D)
            if _side_route(M):
                _on_side_stream(dqkv.device, _wg, (dqkv, x2))
            else:
                _wg()
            dx = None
            if ctx.needs_input_grad[0]:
                dx = linear_dgrad(dqkv, w_ilv, EPI_ADD, dz1, wt=wt_ilv if use_wt else None).view(out_shape)
        else:
            wgrad(dqkv, x2, dwqkv)
            dx = None
            if ctx.needs_input_grad[0]:
                if fp8:
                    dx = linear_dgrad_fp8(dqkv, st, P["qw"], P["vw"], (3 * H, H), EPI_ADD, dz1, role="dqkv", tag=ctx.fp8_tag, pre_q=dqq).view(out_shape)
                else:
                    dx = linear_dgrad(dqkv, st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype), EPI_ADD, dz1,
                                      wt=WT(P["qw"], P["vw"], (3 * H, H))).view(out_shape)
        return BertLayerFn._finish_backward(ctx, grouped, dy2, dx)

    @staticmethod
    def _finish_backward(ctx, grouped, dy2, dx):
        """the layer's grouped weight-gradient launch and the gradient-ready hook (both routes of backward end here)"""
        if grouped:
            if knobs.wgrad_group_side and knobs.wgrad_side_stream and not torch.cuda.is_current_stream_capturing():
                # the layer's grouped weight-gradient launch beside the next layer's backward (small token counts: the main chain's
                # kernels leave CUs idle at their round tails and between launches)
                trip = list(grouped)
                _on_side_stream(dy2.device, lambda: wgrad_group(trip), [t for tr in trip for t in tr[:2]])
            else:
                wgrad_group(grouped)
        hook = ctx.cfg.get("grad_ready_hook")
        if hook is not None:
            # this layer's gradients are all enqueued: the main stream holds the bias / LayerNorm gradients, the side stream the
            # four dW GEMMs.  The hook's all-reduce waits for both streams itself (ops.pending_side_stream ->
            # uc2_comm_allreduce_bucket_after); joining the side stream into the main stream here (round 3) serialised every
            # layer boundary of the main stream behind that layer's weight-gradient GEMMs exactly when N > 1.
            # (small batches: the layer's LayerNorm sums are still pending -- but only flush them per layer when the hook is about to
            #  start a reduction: with an unarmed GradSync, or one rank and no communicator, that would undo the one-launch batching
            #  of every non-final micro-step)
            owner = getattr(hook, "__self__", None)
            will = getattr(owner, "will_reduce", None)
            if will is None or will():
                flush_ln_reductions(end_of_pass=False)
            hook(ctx.layer)
        return (dx, None, None, None) + (None,) * len(ctx.params)
