"""MultiheadAttention with the reference's signature and parameters (reference model/attention.py:267-401:
packed `in_proj_weight [3E, E]`, `(L, N, E)` layout, boolean `key_padding_mask`), on the uc2 HIP kernels.

The reference uses this module only in the NLVR2 head (model/nlvr2.py), always as self-attention with the
packed projection.  That case maps one-to-one onto the fused path of the encoder: one QKV GEMM, the fused
attention kernel with an additive key mask, one output GEMM.  Separate query / key / value inputs
(cross-attention) and an additive float attn_mask [L, S] run on the general kernels (attention_general.hip): three
projections over row slices of the packed in_proj_weight, then uc2_attn_general_*.  What stays unimplemented
(and raises): kdim / vdim != embed_dim, bias_kv, zero_attn, boolean or 3-D attn_mask.  Attention dropout works on both paths
(counter-based masks regenerated in the backward; the returned head-averaged weights are the dropped ones, like the reference's).
"""
import torch
from torch import nn
from torch.nn import Parameter
from torch.nn.init import constant_, xavier_uniform_

from .. import ops
from ..store import compute_dtype_of
from .layer import Linear


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self._qkv_same_embed_dim = self.kdim == embed_dim and self.vdim == embed_dim
        if not self._qkv_same_embed_dim or add_bias_kv or add_zero_attn:
            raise NotImplementedError("uc2_amd.MultiheadAttention implements the packed self-attention case "
                                      "the reference uses (kdim = vdim = embed_dim, no bias_kv / zero_attn)")
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = Linear(embed_dim, embed_dim, bias=bias)
        self.bias_k = self.bias_v = None
        self.add_zero_attn = False
        self._reset_parameters()

    def _reset_parameters(self):
        xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            constant_(self.in_proj_bias, 0.)
            constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        """query = key = value: (L, N, E); key_padding_mask: (N, L) bool, True = ignore that key.
        Returns (attn_output (L, N, E), head-averaged weights (N, L, L) or None)."""
        if not (query is key and key is value) or attn_mask is not None:
            return self._forward_general(query, key, value, key_padding_mask, need_weights, attn_mask)
        L, N, E = query.shape
        cd = compute_dtype_of(self)
        x = query.transpose(0, 1).contiguous()                                   # (N, L, E): token-major rows
        if x.dtype != cd:
            x = ops.cast(x, cd)
        qkv = ops.LinearFn.apply(x, self, ops.EPI_NONE, False, self.in_proj_weight, self.in_proj_bias)
        mask2d = torch.zeros((N, L), dtype=torch.float32, device=query.device)
        if key_padding_mask is not None:
            mask2d.masked_fill_(key_padding_mask.to(torch.bool), -1e30)          # -inf in the reference: weight exactly 0
        qkv2 = qkv.reshape(N * L, 3 * E)
        p = self.dropout if self.training else 0.0
        ctx = ops.AttentionFn.apply(qkv2, mask2d, N, L, self.num_heads, self.head_dim, p, 0x4D48)
        out = self.out_proj(ctx.view(N, L, E)).transpose(0, 1)
        weights = None
        if need_weights:
            with torch.no_grad():
                weights = ops.attn_probs_mean(qkv2.detach(), mask2d, N, L, self.num_heads, self.head_dim)
        return out, weights

    def _forward_general(self, query, key, value, key_padding_mask, need_weights, attn_mask):
        """cross-attention / additive attn_mask (model/attention.py:130-264 with the packed in_proj split in thirds)"""
        L, N, E = query.shape
        S = key.shape[0]
        if key.shape != value.shape or key.shape[1] != N or key.shape[2] != E:
            raise NotImplementedError("key / value must be (S, N, embed_dim)")
        if attn_mask is not None:
            if attn_mask.dim() != 2 or tuple(attn_mask.shape) != (L, S) or not attn_mask.is_floating_point():
                raise NotImplementedError("attn_mask must be an additive float mask of shape (L, S)")
            attn_mask = attn_mask.to(torch.float32).contiguous()
        cd = compute_dtype_of(self)

        def rows(x):                                          # (L, N, E) -> token-major (N*L, E)
            x = x.transpose(0, 1).contiguous()
            return ops.cast(x, cd) if x.dtype != cd else x
        W, bvec = self.in_proj_weight, self.in_proj_bias
        q = ops.LinearFn.apply(rows(query), self, ops.EPI_NONE, False, W, bvec, (0, E))
        k = ops.LinearFn.apply(rows(key), self, ops.EPI_NONE, False, W, bvec, (E, 2 * E))
        v = ops.LinearFn.apply(rows(value), self, ops.EPI_NONE, False, W, bvec, (2 * E, 3 * E))
        kmask = None
        if key_padding_mask is not None:
            kmask = torch.zeros((N, S), dtype=torch.float32, device=query.device)
            kmask.masked_fill_(key_padding_mask.to(torch.bool), -1e30)
        # attention dropout (the reference's NLVR2 head calls this path with attention_probs_dropout_prob, model/nlvr2.py:120-125)
        p = float(self.dropout) if self.training else 0.0
        seed = ops.rng.snapshot(q.device) if p > 0 else None
        site = ops.rng.site(0x4D49)
        ctx, lse = ops.AttentionGeneralFn.apply(q.reshape(N * L, E), k.reshape(N * S, E), v.reshape(N * S, E), kmask, attn_mask,
                                                N, L, S, self.num_heads, self.head_dim, p, seed, site)
        out = self.out_proj(ctx.view(N, L, E)).transpose(0, 1)
        weights = None
        if need_weights:
            with torch.no_grad():
                weights = ops.attn_general_probs_mean(q.detach().reshape(N * L, E), k.detach().reshape(N * S, E), kmask,
                                                      attn_mask, lse, N, L, S, self.num_heads, self.head_dim, p, seed, site)
        return out, weights
