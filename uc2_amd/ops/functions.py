"""torch.autograd.Function wrappers of the remaining fused ops (linear, LayerNorm, embeddings, row selection, losses, tied
decoder + cross-entropy, OT, attention).  Part of uc2_amd.ops."""
import math

import torch

from .. import _lib
from .._lib import call, dt, ptr, stream
from ..config import cfg as knobs
from ..store import store_of
from .base import EPI_GELU, EPI_TANH, rng
from .gemm import _gemm_planned, _wgrad_split, gemm
from .linear import colsum_accum, linear_dgrad, linear_fwd, linear_wgrad
from .kernels import attn_bwd, attn_fwd, ln_bwd, ln_fwd


# --------------------------------------------------------------------------------------
# generic building blocks for embeddings and heads
# --------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); `owner` is the module that owns weight/bias (for the store lookup).
    act: EPI_NONE | EPI_GELU | EPI_TANH.  weight_t=True means `weight` is stored [in, out] and used
    transposed (RegionFeatureRegression: F.linear(h, W_img^T), model/model.py:1155)."""

    @staticmethod
    def forward(ctx, x, owner, act, weight_t, weight, bias, rows=None):
        """rows = (r0, r1): use only output rows r0..r1-1 of weight / bias (the q, k or v third of a packed in_proj)"""
        st = store_of(owner)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = st.compute(weight, dtype)
        b = bias.data if bias is not None else None
        if rows is not None:
            assert not weight_t
            w = w[rows[0]:rows[1]]
            b = b[rows[0]:rows[1]] if b is not None else None
        M, K = x2.shape
        N = w.shape[1] if weight_t else w.shape[0]
        pre = torch.empty((M, N), dtype=dtype, device=x.device) if act == EPI_GELU else None
        if weight_t:
            y = gemm(x2, w, M, N, K, tb=True, bias=b, epi=act, aux_out=pre)
        else:
            y = gemm(x2, w, M, N, K, bias=b, epi=act, aux_out=pre)
        ctx.save_for_backward(x2, pre if act == EPI_GELU else (y if act == EPI_TANH else None))
        ctx.owner, ctx.act, ctx.weight_t, ctx.wb, ctx.shp, ctx.rows = owner, act, weight_t, (weight, bias), shp, rows
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        weight, bias = ctx.wb
        st = store_of(ctx.owner)
        dtype = x2.dtype
        M, K = x2.shape
        N = dy.shape[-1]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.act == EPI_GELU:                 # dpre = dy * gelu'(pre)
            dpre = _dgelu(dy2, aux)
        elif ctx.act == EPI_TANH:
            dpre = torch.empty_like(dy2)
            call("uc2_dtanh", dt(dtype), dy2.numel(), ptr(aux), ptr(dy2), ptr(dpre), stream())
        else:
            dpre = dy2
        w = st.compute(weight, dtype)
        dw = st.grad_buf(weight)
        db = st.grad_buf(bias) if bias is not None else None
        if ctx.rows is not None:
            r0, r1 = ctx.rows
            w, dw = w[r0:r1], dw[r0:r1]
            db = db[r0:r1] if db is not None else None
        if ctx.weight_t:                        # weight [K_in, N_out]: dW[K,N] += X^T dPre
            gemm(x2, dpre, K, N, M, ta=True, tb=True, out=dw, accumulate=True,
                 split_k=_wgrad_split(dtype, K, N, M))
            if db is not None:
                colsum_accum(dpre, db)
        else:
            linear_wgrad(dpre, x2, dw, db)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.weight_t:                    # dX[M,K] = dPre[M,N] W[K,N]^T
                dx = gemm(dpre, w, M, K, N)
            else:
                dx = linear_dgrad(dpre, w)
            dx = dx.view(ctx.shp)
        # one gradient slot per input actually passed (`rows` is only ever passed as a tuple, never as an explicit None)
        return (dx, None, None, None, None, None) + ((None,) if ctx.rows is not None else ())


def _dgelu(dy2, pre):
    """dy * gelu'(pre), elementwise (head transforms; the encoder FFN fuses this into its dgrad GEMM)"""
    out = torch.empty_like(dy2)
    call("uc2_dgelu", dt(dy2.dtype), dy2.numel(), ptr(pre), ptr(dy2), ptr(out), stream())
    return out


class GeluFn(torch.autograd.Function):
    """x * 0.5 * (1 + erf(x / sqrt 2)) as a stand-alone activation (model/layer.py:31-37)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        call("uc2_gelu", dt(x.dtype), x.numel(), ptr(x), ptr(y), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _dgelu(dy.contiguous(), x)


class FusedQKVFn(torch.autograd.Function):
    """q|k|v = x [Wq;Wk;Wv]^T + [bq;bk;bv] in ONE GEMM over the adjacent arena slices of the three nn.Linear
    parameters (model/layer.py:76-78 runs three); gradients go straight into the matching gradient-arena span"""

    @staticmethod
    def forward(ctx, x, owner, qw, qb, kw, kb, vw, vb):
        st = store_of(owner)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        H = qw.shape[1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        wqkv = st.compute_span(qw, vw, (3 * qw.shape[0], H), dtype)
        bqkv = st.span_view(st.data, qb, vb, (3 * qw.shape[0],))
        qkv = linear_fwd(x2, wqkv, bqkv)
        ctx.save_for_backward(x2)
        ctx.owner, ctx.ps, ctx.shp = owner, (qw, qb, kw, kb, vw, vb), x.shape
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        (x2,) = ctx.saved_tensors
        qw, qb, kw, kb, vw, vb = ctx.ps
        st = store_of(ctx.owner)
        H = qw.shape[1]
        dqkv = dqkv.contiguous()
        linear_wgrad(dqkv, x2, st.grad_span(qw, vw, (3 * qw.shape[0], H)), st.grad_span(qb, vb, (3 * qw.shape[0],)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dqkv, st.compute_span(qw, vw, (3 * qw.shape[0], H), x2.dtype)).view(ctx.shp)
        return (dx,) + (None,) * 7


class TiedSubsetDecoderFn(torch.autograd.Function):
    """logits over a SUBSET of the tied decoder's columns: (z E^T + bias)[:, ids] == z E[ids]^T + bias[ids]
    (forward_mmxlm_soft, model/model.py:639-642, keeps 2857 of 250 002 columns): the full-vocabulary logits are
    never formed; dE rows / dbias entries of the subset are accumulated into the gradient arena (ids unique)."""

    @staticmethod
    def forward(ctx, z, owner, weight, bias, ids):
        st = store_of(owner)
        dtype = z.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        n, H = z.shape
        nv = ids.numel()
        z = z.contiguous()
        wsub = torch.empty((nv, H), dtype=dtype, device=z.device)
        wc = st.compute(weight, dtype)
        call("uc2_select_rows", dt(dtype), nv, H, ptr(wc), wc.stride(0), ptr(ids), ptr(wsub), H, 0, stream())
        bsub = torch.empty(nv, dtype=torch.float32, device=z.device)
        call("uc2_gather_f32", nv, ptr(bias.data), ptr(ids), ptr(bsub), 0, stream())
        y = gemm(z, wsub, n, nv, H, bias=bsub)
        ctx.save_for_backward(z, wsub, ids)
        ctx.owner, ctx.wb = owner, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, wsub, ids = ctx.saved_tensors
        weight, bias = ctx.wb
        st = store_of(ctx.owner)
        n, H = z.shape
        nv = ids.numel()
        dy = dy.contiguous()
        dwsub = torch.zeros((nv, H), dtype=torch.float32, device=z.device)
        gemm(dy, z, nv, H, n, ta=True, tb=True, out=dwsub, accumulate=True)
        dE = st.grad_buf(weight)
        call("uc2_select_rows", 0, nv, H, ptr(dwsub), H, ptr(ids), ptr(dE), dE.stride(0), 2, stream())
        dbsub = torch.zeros(nv, dtype=torch.float32, device=z.device)
        colsum_accum(dy, dbsub)
        call("uc2_gather_f32", nv, ptr(dbsub), ptr(ids), ptr(st.grad_buf(bias)), 1, stream())
        dz = gemm(dy, wsub, n, H, nv, tb=True)
        return dz, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """y = LN(dropout(x) + residual) * gamma + beta   (residual optional); with drop_after the dropout
    sits on the output instead: y = dropout(LN(x + residual) * gamma + beta)"""

    @staticmethod
    def forward(ctx, x, residual, owner, eps, drop_p, seed_imm, gamma, beta, beta_extra, drop_after=False):
        st = store_of(owner)
        shp = x.shape
        H = shp[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, H)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        seed = rng.snapshot(x.device) if drop_p > 0 else None
        seed_imm = rng.site(seed_imm)
        b = beta.data if beta_extra is None else (beta.data + beta_extra.data)
        y, mean, rstd = ln_fwd(x2, r2, gamma.data, b, eps, drop_p, seed, seed_imm, drop_after=drop_after)
        ctx.save_for_backward(x2, r2, mean, rstd, seed)
        ctx.owner, ctx.gb, ctx.cfg, ctx.shp = owner, (gamma, beta, beta_extra), (drop_p, seed_imm, drop_after), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, mean, rstd, seed = ctx.saved_tensors
        gamma, beta, beta_extra = ctx.gb
        drop_p, seed_imm, drop_after = ctx.cfg
        st = store_of(ctx.owner)
        H = x2.shape[1]
        dy2 = dy.reshape(-1, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dbeta = st.grad_buf(beta)
        dx, dres = ln_bwd(dy2, x2, r2, gamma.data, mean, rstd, st.grad_buf(gamma), dbeta, drop_p, seed, seed_imm,
                          need_dres=r2 is not None, drop_after=drop_after)
        dextra = None
        if beta_extra is not None and ctx.needs_input_grad[8]:
            # d(beta + extra) flows to both; beta got it through the arena, extra gets a fresh column sum
            if drop_after and drop_p > 0:
                raise _lib.Uc2Error("beta_extra with output dropout is not supported")
            dextra = torch.zeros(H, dtype=torch.float32, device=dy.device)
            colsum_accum(dy2, dextra)
        return (dx.view(ctx.shp) if ctx.needs_input_grad[0] else None,
                dres.view(ctx.shp) if (r2 is not None and ctx.needs_input_grad[1]) else None,
                None, None, None, None, None, None, dextra, None)


class EmbedTextFn(torch.autograd.Function):
    """word[ids] + pos[pos_ids] + type[type_ids or 0]  (model/model.py:322-330), output in compute dtype"""

    @staticmethod
    def forward(ctx, owner, dtype, ids, pos_ids, type_ids, word, pos, typ, word_pad=-1, pos_pad=-1):
        B, T = ids.shape
        H = word.shape[1]
        out = torch.empty((B, T, H), dtype=dtype, device=ids.device)
        ids_c, pos_c = ids.contiguous(), pos_ids.contiguous()
        ty_c = type_ids.contiguous() if type_ids is not None else None
        call("uc2_embed_fwd", dt(dtype), B * T, H, ptr(ids_c), ptr(pos_c), ptr(ty_c), 0, ptr(word.data), ptr(pos.data),
             ptr(typ.data), ptr(out), stream())
        ctx.save_for_backward(ids_c, pos_c, ty_c)
        ctx.owner, ctx.tabs, ctx.H, ctx.pads = owner, (word, pos, typ), H, (int(word_pad), int(pos_pad))
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, pos_ids, type_ids = ctx.saved_tensors
        word, pos, typ = ctx.tabs
        st = store_of(ctx.owner)
        H = ctx.H
        d2 = dout.reshape(-1, H)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        rows = d2.shape[0]
        dtyp = st.grad_buf(typ)
        # ids are [B, T]: position / type rows repeat down the batch and are summed in registers (uc2_embed_bwd_seq); -2 = shape not
        # taken, the row-per-wave kernel then adds every token's row with atomics
        Bn, Tn = ids.shape
        rc = _lib.load().uc2_embed_bwd_seq(dt(d2.dtype), Bn, Tn, H, ptr(ids), ptr(pos_ids), ptr(type_ids), ptr(d2), ptr(st.grad_buf(word)),
                                           ptr(st.grad_buf(pos)), ptr(dtyp), ctx.pads[0], ctx.pads[1], stream()) if knobs.embed_bwd_seq else -2
        if rc == -2:
            call("uc2_embed_bwd", dt(d2.dtype), rows, H, ptr(ids), ptr(pos_ids), ptr(type_ids), ptr(d2),
                 ptr(st.grad_buf(word)), ptr(st.grad_buf(pos)), ptr(dtyp), ctx.pads[0], ctx.pads[1], stream())
        else:
            _lib.check(rc)
        if type_ids is None:         # constant type 0: its row gets the column sum (no atomic pile-up on one row)
            colsum_accum(d2, dtyp[0])
        return (None,) * 10


def padded_rows(M, dtype):
    """token rows an encoder pass over M = B L tokens runs on: M rounded up to a whole number of 256-row GEMM tiles (bf16, from
    1 024 tokens, UC2_PAD_ROWS=0 off), else M.  The reference's token-bucket batches have a new B x L every step
    (data/sampler.py:11-59); a ragged M used to push every GEMM of the step off the ping-pong kernels' plans (bench.py
    itm_rank_finetune: 612 re-routed calls per 8 steps) and to tune a new shape per step."""
    if dtype != torch.bfloat16 or not knobs.pad_rows or M < 1024 or M % 256 == 0:
        return M
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return M
    return (M + 255) // 256 * 256


class PadRowsFn(torch.autograd.Function):
    """[B, L, H] -> [rows, H], rows >= B L: the tokens followed by zero rows (see padded_rows, BertLayerFn)"""

    @staticmethod
    def forward(ctx, x, rows):
        B, L, H = x.shape
        M = B * L
        out = torch.empty((rows, H), dtype=x.dtype, device=x.device)
        out[:M].copy_(x.reshape(M, H))
        out[M:].zero_()
        ctx.shape = (B, L, H)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, L, H = ctx.shape
        return dy[:B * L].view(B, L, H), None


class UnpadRowsFn(torch.autograd.Function):
    """[rows, H] -> the first B L rows as [B, L, H] (a view); the gradient of the dropped rows is zero"""

    @staticmethod
    def forward(ctx, x2, B, L):
        ctx.rows = x2.shape[0]
        return x2[:B * L].view(B, L, x2.shape[1])

    @staticmethod
    def backward(ctx, dy):
        B, L, H = dy.shape
        out = torch.empty((ctx.rows, H), dtype=dy.dtype, device=dy.device)
        out[:B * L].copy_(dy.reshape(B * L, H))
        out[B * L:].zero_()
        return out, None, None


class GatherRowsFn(torch.autograd.Function):
    """torch.gather(src, 1, index[..., None].expand(H)) (model/model.py:420-425)"""

    @staticmethod
    def forward(ctx, src, index):
        B, S, H = src.shape
        L = index.shape[1]
        src = src.contiguous()
        idx = index.contiguous()
        out = torch.empty((B, L, H), dtype=src.dtype, device=src.device)
        call("uc2_gather_rows_fwd", dt(src.dtype), B, S, L, H, ptr(src), ptr(idx), ptr(out), stream())
        ctx.save_for_backward(idx)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, L, H = dout.shape
        dout = dout.contiguous()
        dsrc = torch.empty((B, ctx.S, H), dtype=dout.dtype, device=dout.device)
        call("uc2_gather_rows_bwd", dt(dout.dtype), B, ctx.S, L, H, ptr(dout), ptr(idx), ptr(dsrc), stream())
        return dsrc, None


class GatherCatRowsFn(torch.autograd.Function):
    """torch.gather(torch.cat([a, b], 1), 1, index[..., None].expand(H)) (model/model.py:412-425) without the concatenated tensor:
    the gather reads the text and image embeddings in place; the backward writes their two gradients as separate tensors"""

    @staticmethod
    def forward(ctx, a, b, index):
        B, S1, H = a.shape
        S2 = b.shape[1]
        L = index.shape[1]
        a, b = a.contiguous(), b.contiguous()
        idx = index.contiguous()
        out = torch.empty((B, L, H), dtype=a.dtype, device=a.device)
        call("uc2_gather_rows2_fwd", dt(a.dtype), B, S1, S2, L, H, ptr(a), ptr(b), ptr(idx), ptr(out), stream())
        ctx.save_for_backward(idx)
        ctx.S = (S1, S2)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, L, H = dout.shape
        S1, S2 = ctx.S
        dout = dout.contiguous()
        da = torch.empty((B, S1, H), dtype=dout.dtype, device=dout.device)
        db = torch.empty((B, S2, H), dtype=dout.dtype, device=dout.device)
        call("uc2_gather_rows2_bwd", dt(dout.dtype), B, S1, S2, L, H, ptr(dout), ptr(idx), ptr(da), ptr(db), stream())
        return da, db, None


class SelectRowsFn(torch.autograd.Function):
    """hidden[mask] for a boolean mask over rows (model/model.py:653-657); rows = flat row indices"""

    @staticmethod
    def forward(ctx, hidden2, rows):
        R, H = hidden2.shape
        n = rows.numel()
        out = torch.empty((n, H), dtype=hidden2.dtype, device=hidden2.device)
        call("uc2_select_rows", dt(hidden2.dtype), n, H, ptr(hidden2), hidden2.stride(0), ptr(rows), ptr(out), H, 0,
             stream())
        ctx.save_for_backward(rows)
        ctx.R = R
        return out

    @staticmethod
    def backward(ctx, dout):
        (rows,) = ctx.saved_tensors
        n, H = dout.shape
        dout = dout.contiguous()
        dsrc = torch.zeros((ctx.R, H), dtype=dout.dtype, device=dout.device)
        call("uc2_select_rows", dt(dout.dtype), n, H, ptr(dout), H, ptr(rows), ptr(dsrc), H, 1, stream())
        return dsrc, None


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits, labels, ignore_index, reduction='none'); logits are consumed (overwritten
    by dlogits in backward).  Also returns argmax (int64) as a non-differentiable side output."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index, n_valid_cols):
        n = logits.shape[0]
        V = n_valid_cols
        labels = labels.contiguous()
        loss = torch.empty(n, dtype=torch.float32, device=logits.device)
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        am = torch.empty(n, dtype=torch.int64, device=logits.device)
        call("uc2_ce_fwd", dt(logits.dtype), n, V, ptr(logits), logits.stride(0), ptr(labels), ignore_index, ptr(loss),
             ptr(lse), ptr(am), stream())
        ctx.save_for_backward(logits, labels, lse)
        ctx.cfg = (ignore_index, V)
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, gloss, _gam):
        logits, labels, lse = ctx.saved_tensors
        ignore_index, V = ctx.cfg
        n = logits.shape[0]
        g = gloss.contiguous().float()
        dlog = logits            # in place: the logits buffer becomes dlogits
        call("uc2_ce_bwd", dt(logits.dtype), n, V, ptr(dlog), dlog.stride(0), ptr(labels), ignore_index, ptr(lse),
             ptr(g), stream())
        return dlog, None, None, None


class KLDivFn(torch.autograd.Function):
    """F.kl_div(F.log_softmax(pred, -1), target, reduction='none') (model/model.py:764-768)"""

    @staticmethod
    def forward(ctx, pred, target, n_valid_cols):
        n, V = pred.shape[0], n_valid_cols
        target = target.contiguous().float()
        lse = torch.empty(n, dtype=torch.float32, device=pred.device)
        call("uc2_ce_fwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), None, -100, None, ptr(lse), None, stream())
        loss = torch.empty((n, V), dtype=torch.float32, device=pred.device)
        call("uc2_kl_fwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), ptr(target), ptr(lse), ptr(loss), stream())
        ctx.save_for_backward(pred, target, lse)
        ctx.V = V
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, target, lse = ctx.saved_tensors
        n, V = pred.shape[0], ctx.V
        g = gloss.contiguous().float()
        dpred = torch.zeros_like(pred) if pred.shape[1] > V else torch.empty_like(pred)
        call("uc2_kl_bwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), ptr(target), ptr(lse), ptr(g), ptr(dpred),
             stream())
        return dpred, None, None


class MSEFn(torch.autograd.Function):
    """F.mse_loss(pred, target, reduction='none') (model/model.py:684-686)"""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        target = target.contiguous().float()
        loss = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
        call("uc2_mse", dt(pred.dtype), pred.numel(), ptr(pred), ptr(target), None, ptr(loss), None, stream())
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, target = ctx.saved_tensors
        g = gloss.contiguous().float()
        dpred = torch.empty_like(pred)
        call("uc2_mse", dt(pred.dtype), pred.numel(), ptr(pred), ptr(target), ptr(g), None, ptr(dpred), stream())
        return dpred, None


class TripletFn(torch.autograd.Function):
    """sigmoid -> view(-1, sample_size) -> clamp(margin + neg - pos, 0) (model/itm.py:45-53)"""

    @staticmethod
    def forward(ctx, scores, sample_size, margin):
        s = scores.contiguous().view(-1)
        n = s.numel() // sample_size
        loss = torch.empty((n, sample_size - 1), dtype=torch.float32, device=s.device)
        call("uc2_triplet", dt(s.dtype), n, sample_size, margin, ptr(s), None, ptr(loss), None, stream())
        ctx.save_for_backward(s)
        ctx.cfg = (n, sample_size, margin, scores.shape)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (s,) = ctx.saved_tensors
        n, ss, margin, shp = ctx.cfg
        g = gloss.contiguous().float()
        ds = torch.empty_like(s)
        call("uc2_triplet", dt(s.dtype), n, ss, margin, ptr(s), ptr(g), None, ptr(ds), stream())
        return ds.view(shp), None, None


def add_rowvec(a, b, vec, rowmask, out_dtype):
    """out = a + b + (rowmask ? vec : 0) row-wise; a may be fp32 while out is the compute dtype"""
    H = a.shape[-1]
    a2 = a.reshape(-1, H)
    if not a2.is_contiguous():
        a2 = a2.contiguous()
    b2 = None
    if b is not None:
        b2 = b.reshape(-1, H)
        if not b2.is_contiguous():
            b2 = b2.contiguous()
        assert b2.dtype == out_dtype
    out = torch.empty(a2.shape, dtype=out_dtype, device=a.device)
    call("uc2_add_rowvec", dt(a2.dtype), dt(out_dtype), a2.shape[0], H, ptr(a2), ptr(b2), ptr(vec), ptr(rowmask),
         ptr(out), stream())
    return out.view(a.shape)


class MaskEmbedFn(torch.autograd.Function):
    """cast(img_feat) + mask_embedding(img_masks) with row 0 == 0 and no gradient to row 0
    (nn.Embedding(2, img_dim, padding_idx=0), model/model.py:347,353-356)"""

    @staticmethod
    def forward(ctx, owner, img_feat, img_masks, weight, out_dtype):
        m8 = img_masks.reshape(-1).to(torch.uint8).contiguous()
        out = add_rowvec(img_feat, None, weight.data[1], m8, out_dtype)
        ctx.save_for_backward(m8)
        ctx.owner, ctx.weight = owner, weight
        return out

    @staticmethod
    def backward(ctx, dout):
        (m8,) = ctx.saved_tensors
        st = store_of(ctx.owner)
        d2 = dout.reshape(-1, dout.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        colsum_accum(d2, st.grad_buf(ctx.weight)[1], m8)
        return None, None, None, None, None


class AddRowFn(torch.autograd.Function):
    """a + b + table[row]  (transformed_im + transformed_pos + type embedding, model/model.py:360)"""

    @staticmethod
    def forward(ctx, a, b, table, row):
        out = add_rowvec(a, b, table.data[row], None, a.dtype)
        ctx.table, ctx.row = table, row
        return out

    @staticmethod
    def backward(ctx, dout):
        st = getattr(ctx.table, "_uc2_store", None)
        d2 = dout.reshape(-1, dout.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        if st is not None:
            colsum_accum(d2, st.grad_buf(ctx.table)[ctx.row])
            dtab = None
        else:
            dtab = torch.zeros_like(ctx.table.data)
            colsum_accum(d2, dtab[ctx.row])
        return dout, dout, dtab, None


_DEC_ROWS = 8192          # rows per decoder chunk: 2 * rows * 250 112 bytes of logits must stay below 2^32 (the ping-pong
                          # kernel addresses its operands with 32-bit byte offsets), and a chunk's logits are 4.1 GB


def _dec_chunks(npad):
    """equal row chunks (multiples of 256, at most _DEC_ROWS): 9216 masked rows are 2 x 4608, not 8192 + 1024 -- the
    remainder chunk ran the vocabulary-long GEMMs on a handful of tiles"""
    if npad <= 0:
        return []
    n = (npad + _DEC_ROWS - 1) // _DEC_ROWS
    rows = ((npad + n - 1) // n + 255) // 256 * 256 if npad % 256 == 0 else _DEC_ROWS
    return [(r0, min(npad, r0 + rows)) for r0 in range(0, npad, rows)]


class DecoderCEFn(torch.autograd.Function):
    """tied-decoder logits + cross entropy in one node (model/layer.py:257-265, model/model.py:590-596).
    The vocabulary tables are padded to whole 256-row GEMM tiles inside the arena (store.padded) and the masked rows
    to a multiple of 256 (zero rows, ignored labels), so the three decoder GEMMs -- logits = z E^T + bias,
    dE += dlogits^T z, dz = dlogits E -- all run on the persistent 256x256 MFMA kernel; logits live in ONE
    [rows, 250112] bf16 buffer per chunk of 8192 rows that the backward overwrites with dlogits, and dE goes straight
    into the word-embedding gradient arena."""

    @staticmethod
    def forward(ctx, z, owner, weight, bias, labels, ignore_index):
        st = store_of(owner)
        dtype = z.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        n, H = z.shape
        V = weight.shape[0]
        in_arena = st.owns(weight) and st.owns(bias)
        Wp = st.padded(st.data if dtype == torch.float32 else st.shadow, weight) if in_arena else st.compute(weight, dtype)
        bp = st.padded(st.data, bias) if in_arena else bias.data
        Vp = Wp.shape[0] if Wp.shape[0] % 8 == 0 else (V + 7) // 8 * 8
        tile = 256 if (dtype == torch.bfloat16 and n >= 256) else 1
        npad = (n + tile - 1) // tile * tile
        z = z.contiguous()
        labels = labels.contiguous()
        if npad != n:                                 # zero rows / ignored labels up to whole tiles
            zp = torch.zeros((npad, H), dtype=dtype, device=z.device)
            zp[:n].copy_(z)
            lp = torch.full((npad,), ignore_index, dtype=labels.dtype, device=z.device)
            lp[:n].copy_(labels)
            z, labels = zp, lp
        loss = torch.empty(npad, dtype=torch.float32, device=z.device)
        lse = torch.empty(npad, dtype=torch.float32, device=z.device)
        am = torch.empty(npad, dtype=torch.int64, device=z.device)
        chunks = []
        for r0, r1 in _dec_chunks(npad):
            m = r1 - r0
            logits = torch.empty((m, Vp), dtype=dtype, device=z.device)
            if Wp.shape[0] == Vp and Vp != V:         # whole padded tiles: N = Vp (padding columns = padding bias = 0)
                _gemm_planned(z[r0:r1], Wp, m, Vp, H, False, False, out=logits, bias=bp)
            else:
                gemm(z[r0:r1], Wp, m, V, H, out=logits, bias=bp)
            call("uc2_ce_fwd", dt(dtype), m, V, ptr(logits), Vp, ptr(labels[r0:r1]), ignore_index, ptr(loss[r0:r1]),
                 ptr(lse[r0:r1]), ptr(am[r0:r1]), stream())
            chunks.append(logits)
        ctx.save_for_backward(z, labels, lse, *chunks)
        ctx.owner, ctx.wb, ctx.cfg = owner, (weight, bias), (ignore_index, V, Vp, n, npad)
        loss, am = loss[:n], am[:n]
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, gloss, _g):
        z, labels, lse = ctx.saved_tensors[:3]
        chunks = ctx.saved_tensors[3:]
        weight, bias = ctx.wb
        ignore_index, V, Vp, n, npad = ctx.cfg
        st = store_of(ctx.owner)
        dtype = z.dtype
        H = z.shape[1]
        g = gloss.contiguous().float()
        if npad != n:
            gp = torch.zeros(npad, dtype=torch.float32, device=g.device)
            gp[:n].copy_(g)
            g = gp
        in_arena = st.owns(weight) and st.owns(bias)
        Wp = st.padded(st.data if dtype == torch.float32 else st.shadow, weight) if in_arena else st.compute(weight, dtype)
        st.grad_buf(weight)
        st.grad_buf(bias)
        dE = st.padded(st.grad, weight) if in_arena else st.grad_buf(weight)
        db = st.padded(st.grad, bias) if in_arena else st.grad_buf(bias)
        full = Wp.shape[0] == Vp and Vp != V
        dz = torch.empty((npad, H), dtype=dtype, device=z.device)
        for ci, (r0, r1) in enumerate(_dec_chunks(npad)):
            m = r1 - r0
            dlog = chunks[ci]                          # in place: the logits buffer becomes dlogits (padding columns zeroed)
            # dlogits in place, with dbias[V] += colsum(dlogits) from the same pass when the rows are vectorisable
            rc = _lib.load().uc2_ce_bwd_colsum(dt(dtype), m, V, ptr(dlog), Vp, ptr(labels[r0:r1]), ignore_index,
                                               ptr(lse[r0:r1]), ptr(g[r0:r1]), ptr(db), Vp if full else V, stream())
            have_db = rc == 0
            if rc not in (0, -2):
                _lib.check(rc)
            if not have_db:
                call("uc2_ce_bwd", dt(dtype), m, V, ptr(dlog), Vp, ptr(labels[r0:r1]), ignore_index, ptr(lse[r0:r1]),
                     ptr(g[r0:r1]), stream())
            # dE[V,H] += dlogits^T z ; dz = dlogits E
            if full:
                _gemm_planned(dlog, z[r0:r1], Vp, H, m, True, True, wgrad=True, out=dE, accumulate=True, lda=Vp)
                if not have_db:
                    call("uc2_colsum_accum", dt(dtype), m, Vp, ptr(dlog), Vp, None, ptr(db), stream())
                if dtype == torch.bfloat16 and m >= 256:
                    # m x H is only (m/256) x 3 tiles (96 at 8192 rows) under a contraction of 250 112: split it over the
                    # vocabulary like a weight gradient (fp32 partial tiles + one reduction pass), then round once
                    dz32 = torch.zeros((m, H), dtype=torch.float32, device=z.device)
                    Et = st.table_t(weight) if (knobs.decoder_wt and in_arena) else None
                    if Et is not None:
                        # k-contiguous E^T [H, Vp]: the layout of a forward GEMM (both operands read with plain ds_read_b128; the NT form
                        # ran 3 032 us per 7 680-row chunk = 0.97 PF/s).  Same box, alternating: 385.5 / 385.9 -> 382.9 / 383.1 ms per
                        # 6144-pair MLM step (profiles/r06_experiments.md section 6)
                        _gemm_planned(dlog, Et, m, H, Vp, False, False, wgrad=True, out=dz32, accumulate=True, lda=Vp)
                    else:
                        _gemm_planned(dlog, Wp, m, H, Vp, False, True, wgrad=True, out=dz32, accumulate=True, lda=Vp)
                    call("uc2_cast", dt(torch.float32), dt(dtype), m * H, ptr(dz32), ptr(dz[r0:r1]), stream())
                else:
                    _gemm_planned(dlog, Wp, m, H, Vp, False, True, out=dz[r0:r1], lda=Vp)
            else:
                gemm(dlog, z[r0:r1], V, H, m, ta=True, tb=True, out=dE, accumulate=True, lda=Vp,
                     split_k=_wgrad_split(dtype, V, H, m))
                if not have_db:
                    call("uc2_colsum_accum", dt(dtype), m, V, ptr(dlog), Vp, None, ptr(db), stream())
                gemm(dlog, Wp, m, H, V, tb=True, out=dz[r0:r1], lda=Vp)
        return dz[:n], None, None, None, None, None


class OTDistFn(torch.autograd.Function):
    """optimal_transport_dist of the scattered-back text / image embeddings (model/ot.py:66-82, model/model.py:701-720):
    dist [B] fp32; the transport plan is a constant of the backward (the reference detaches it)"""

    @staticmethod
    def forward(ctx, seq, scatter, txt_pad, img_pad, T, R, beta, iters):
        B, L, H = seq.shape
        seq = seq.contiguous()
        scatter = scatter.contiguous()
        tp = txt_pad.to(torch.uint8).contiguous()
        ip = img_pad.to(torch.uint8).contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.uc2_ot_workspace(B, T, R, H), dtype=torch.uint8, device=seq.device)
        dist = torch.empty(B, dtype=torch.float32, device=seq.device)
        Tm = torch.empty((B, R, T), dtype=torch.float32, device=seq.device)
        call("uc2_ot_fwd", dt(seq.dtype), B, L, T, R, H, ptr(seq), ptr(scatter), ptr(tp), ptr(ip), float(beta), int(iters),
             ptr(dist), ptr(Tm), ptr(ws), stream())
        ctx.save_for_backward(Tm, ws)
        ctx.cfg = (B, L, T, R, H, seq.dtype)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        Tm, ws = ctx.saved_tensors
        B, L, T, R, H, dtype = ctx.cfg
        g = gdist.contiguous().float()
        dseq = torch.zeros((B, L, H), dtype=dtype, device=Tm.device)
        call("uc2_ot_bwd", dt(dtype), B, L, T, R, H, ptr(Tm), ptr(ws), ptr(g), ptr(dseq), stream())
        return dseq, None, None, None, None, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(QK^T/sqrt(d) + mask) V over a packed [B*L, 3H] projection (one node; used by MultiheadAttention)"""

    @staticmethod
    def forward(ctx, qkv2, mask2d, B, L, nh, D, drop_p, seed_imm):
        seed = rng.snapshot(qkv2.device) if drop_p > 0 else None
        seed_imm = rng.site(seed_imm)
        ctxv, lse = attn_fwd(qkv2, mask2d, B, L, nh, D, drop_p, seed, seed_imm)
        ctx.save_for_backward(qkv2, mask2d, ctxv, lse, seed)
        ctx.cfg = (B, L, nh, D, drop_p, seed_imm)
        return ctxv

    @staticmethod
    def backward(ctx, dctx):
        qkv2, mask2d, ctxv, lse, seed = ctx.saved_tensors
        B, L, nh, D, drop_p, seed_imm = ctx.cfg
        dqkv = attn_bwd(qkv2, mask2d, ctxv, dctx.contiguous(), lse, B, L, nh, D, drop_p, seed, seed_imm)
        return dqkv, None, None, None, None, None, None, None


class AttentionGeneralFn(torch.autograd.Function):
    """softmax(scale q k^T + key_mask + attn_mask) v with separate q / k / v [B*L, nh*D] tensors (cross-attention,
    additive attn_mask): the general form behind MultiheadAttention (model/attention.py:12-264); fp32 math; dropout on the
    probabilities with counter-based masks (the seed copy is a third, non-differentiable output for need_weights)"""

    @staticmethod
    def forward(ctx, q2, k2, v2, key_mask, attn_mask, B, Lq, Lk, nh, D, drop_p=0.0, seed=None, seed_imm=0):
        """seed: the caller's dropout seed copy (rng.snapshot; it also feeds attn_general_probs_mean), seed_imm: its site number"""
        q2, k2, v2 = q2.contiguous(), k2.contiguous(), v2.contiguous()
        H = nh * D
        out = torch.empty((B * Lq, H), dtype=q2.dtype, device=q2.device)
        lse = torch.empty((B, nh, Lq), dtype=torch.float32, device=q2.device)
        call("uc2_attn_general_fwd", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(v2), H, ptr(key_mask),
             ptr(attn_mask), 1.0 / math.sqrt(D), ptr(out), H, ptr(lse), float(drop_p), ptr(seed), seed_imm, stream())
        ctx.save_for_backward(q2, k2, v2, key_mask, attn_mask, out, lse, seed)
        ctx.cfg = (B, Lq, Lk, nh, D, float(drop_p), seed_imm)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q2, k2, v2, key_mask, attn_mask, out, lse, seed = ctx.saved_tensors
        B, Lq, Lk, nh, D, drop_p, seed_imm = ctx.cfg
        H = nh * D
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        delta = torch.empty((B, nh, Lq), dtype=torch.float32, device=q2.device)
        call("uc2_attn_general_bwd", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(v2), H, ptr(key_mask),
             ptr(attn_mask), 1.0 / math.sqrt(D), ptr(out), ptr(dout), H, ptr(lse), ptr(delta), ptr(dq), H, ptr(dk), H,
             ptr(dv), H, drop_p, ptr(seed), seed_imm, stream())
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None


def attn_general_probs_mean(q2, k2, key_mask, attn_mask, lse, B, Lq, Lk, nh, D, drop_p=0.0, seed=None, seed_imm=0):
    out = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q2.device)
    H = nh * D
    call("uc2_attn_general_probs_mean", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(key_mask), ptr(attn_mask),
         1.0 / math.sqrt(D), ptr(lse), ptr(out), float(drop_p), ptr(seed), seed_imm, stream())
    return out


def attn_probs_mean(qkv2, mask2d, B, L, nh, D):
    out = torch.empty((B, L, L), dtype=torch.float32, device=qkv2.device)
    call("uc2_attn_probs_mean", dt(qkv2.dtype), B, L, nh, D, ptr(qkv2), ptr(mask2d), 1.0 / math.sqrt(D), ptr(out),
         stream())
    return out
