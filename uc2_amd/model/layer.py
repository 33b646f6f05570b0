"""Transformer block and head pieces with the reference's class names, constructor arguments,
forward() signatures and state_dict keys (reference model/layer.py), backed by the HIP kernels.

Modules hold ordinary fp32 nn.Parameters (the master weights); forward() dispatches to
uc2_amd.ops.  Inputs/outputs are in the module's compute dtype (float32 = parity mode,
bfloat16 = throughput mode; see uc2_amd.set_compute_dtype).
"""
import torch
from torch import nn

from .. import ops
from ..store import compute_dtype_of, store_of

_LAYER_COUNTER = [0]


class BertLayerNorm(nn.Module):
    """apex FusedLayerNorm replacement (model/layer.py:25): same parameters (`weight`, `bias`), same math."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.eps = eps
        self.normalized_shape = (hidden_size,)

    def forward(self, x, residual=None, drop_p=0.0, seed_imm=0, beta_extra=None, drop_after=False):
        """LN(dropout_p(x) + residual); drop_after=True puts the dropout on the output (the embedding tails)"""
        x = ops.cast(x, compute_dtype_of(self)) if x.dtype != compute_dtype_of(self) else x
        return ops.LayerNormFn.apply(x, residual, self, self.eps, drop_p, seed_imm, self.weight, self.bias, beta_extra,
                                     drop_after)


LayerNorm = BertLayerNorm


class Linear(nn.Linear):
    """nn.Linear whose forward/backward run on the uc2 GEMM kernels (same parameters and keys)."""

    def forward(self, x, act=ops.EPI_NONE):
        cd = compute_dtype_of(self)
        if x.dtype != cd:
            x = ops.cast(x, cd)
        return ops.LinearFn.apply(x, self, act, False, self.weight, self.bias)


def gelu(x):
    """model/layer.py:31-37 (erf form)"""
    return ops.GeluFn.apply(x)


class GELU(nn.Module):
    """model/layer.py:40-50 (GELU module wrapping gelu()): erf-form GELU as its own kernel.  The heads call their
    Linear with act=EPI_GELU (fused epilogue) and skip this module; calling the Sequential the plain way
    (`net(x)`) runs Linear -> this -> LayerNorm and gives the same result."""

    def forward(self, x):
        cd = compute_dtype_of(self)
        if x.dtype != cd:
            x = ops.cast(x, cd)
        return ops.GeluFn.apply(x)


class BertSelfAttention(nn.Module):
    """model/layer.py:53-101.  Inside a BertLayer the math runs in BertLayerFn (one autograd node); called on its
    own, forward() composes the same kernels: one fused QKV GEMM over the adjacent q|k|v parameters, then the
    fused attention kernel (scores never reach HBM), context written head-merged."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask):
        """hidden_states [B,L,H], attention_mask [B,1,1,L] additive -> context [B,L,H] (model/layer.py:75-101)"""
        cd = compute_dtype_of(self)
        x = hidden_states if hidden_states.dtype == cd else ops.cast(hidden_states, cd)
        B, L, H = x.shape
        mask2d = ops._mask2d(attention_mask, B, L)
        q, k, v = self.query, self.key, self.value
        qkv = ops.FusedQKVFn.apply(x, self, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias)
        p = self.dropout.p if self.training else 0.0
        ctx = ops.AttentionFn.apply(qkv, mask2d, B, L, self.num_attention_heads, self.attention_head_size, p, 0x5341)
        return ctx.view(B, L, H)


class BertSelfOutput(nn.Module):
    """model/layer.py:104-115: LayerNorm(dropout(dense(hidden_states)) + input_tensor)"""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        p = self.dropout.p if self.training else 0.0
        return self.LayerNorm(self.dense(hidden_states), input_tensor, p, 0x534F)


class BertAttention(nn.Module):
    """model/layer.py:118-127"""

    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output = self.self(input_tensor, attention_mask)
        return self.output(self_output, input_tensor)


class BertIntermediate(nn.Module):
    """model/layer.py:130-142: gelu(dense(x)), GELU fused into the GEMM epilogue"""

    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "gelu":
            raise ValueError("uc2_amd implements the erf-GELU feed-forward only (config/uc2-base.json)")
        self.dense = Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return self.dense(hidden_states, act=ops.EPI_GELU)


class BertOutput(nn.Module):
    """model/layer.py:145-156: LayerNorm(dropout(dense(hidden_states)) + input_tensor)"""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        p = self.dropout.p if self.training else 0.0
        return self.LayerNorm(self.dense(hidden_states), input_tensor, p, 0x424F)


class BertLayer(nn.Module):
    """model/layer.py:159-170.  forward(hidden_states [B,L,H], attention_mask [B,1,1,L] additive) -> [B,L,H]"""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self._layer_id = _LAYER_COUNTER[0]
        _LAYER_COUNTER[0] += 1
        self.grad_ready_hook = None

    def __deepcopy__(self, memo):
        # VLXLMREncoder deep-copies one template layer (model/model.py:369-371); give each copy its own id
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_layer_id"] = _LAYER_COUNTER[0]
        new.__dict__.pop("_uc2_store_cache", None)
        _LAYER_COUNTER[0] += 1
        return new

    def forward_unfused(self, hidden_states, attention_mask):
        """the reference's composition (model/layer.py:166-170) through the sub-modules' own forward()s -- same
        kernels, one autograd node per sub-module instead of one for the layer; used by the parity tests"""
        attention_output = self.attention(hidden_states, attention_mask)
        intermediate_output = self.intermediate(attention_output)
        return self.output(intermediate_output, attention_output)

    def forward(self, hidden_states, attention_mask):
        cd = compute_dtype_of(self)
        x = hidden_states if hidden_states.dtype == cd else ops.cast(hidden_states, cd)
        if x.dim() == 2:
            # [rows, H] with rows >= B L: the padded-rows form VLXLMREncoder.forward passes from layer to layer when B L is not a
            # whole number of GEMM tiles (ops.PadRowsFn); B and L are the mask's
            B, L = attention_mask.shape[0], attention_mask.shape[-1]
        else:
            B, L, H = x.shape
        mask2d = ops._mask2d(attention_mask, B, L)
        a = self.attention
        cfg = {"nh": a.self.num_attention_heads, "training": self.training,
               "p_hidden": a.output.dropout.p, "p_attn": a.self.dropout.p,
               "layer_id": self._layer_id, "grad_ready_hook": self.grad_ready_hook,
               "fp8": self.__dict__.get("uc2_fp8", False)}
        return ops.BertLayerFn.apply(x, mask2d, self, cfg, *ops.layer_params(self))


class BertPooler(nn.Module):
    """model/layer.py:173-185: tanh(W h[:,0] + b), tanh fused in the GEMM epilogue"""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        first = hidden_states[:, 0].contiguous()
        return self.dense(first, act=ops.EPI_TANH)


class RobertaLMHead(nn.Module):
    """model/layer.py:236-265: dense -> gelu -> LN(eps) -> tied decoder (+ bias)."""

    def __init__(self, config, roberta_model_embedding_weights):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.decoder = Linear(roberta_model_embedding_weights.size(1), roberta_model_embedding_weights.size(0),
                              bias=False)
        self.decoder.weight = roberta_model_embedding_weights
        self.bias = nn.Parameter(torch.zeros(roberta_model_embedding_weights.size(0)))
        self.decoder.bias = self.bias

    def transform(self, features):
        return self.layer_norm(self.dense(features, act=ops.EPI_GELU))

    def forward(self, features, **kwargs):
        return self.decoder(self.transform(features))


class VisualRobertaLMHead(nn.Module):
    """model/layer.py:267-294: constructed by the reference but never called;
    parameters kept for checkpoint compatibility."""

    def __init__(self, config, roberta_model_embedding_weights, valid_roberta_model_embedding_index):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        n = len(valid_roberta_model_embedding_index)
        self.decoder = Linear(roberta_model_embedding_weights.size(1), n, bias=False)
        self.decoder.weight = nn.Parameter(
            roberta_model_embedding_weights.data[valid_roberta_model_embedding_index].clone())
        self.bias = nn.Parameter(torch.zeros(n))
        self.decoder.bias = self.bias

    def forward(self, features, **kwargs):
        return self.decoder(self.layer_norm(self.dense(features, act=ops.EPI_GELU)))
