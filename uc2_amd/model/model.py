"""VL-XLM-R model shells with the reference's names, signatures, batch-dict keys and state_dict
layout (reference model/model.py:1-776, 1143-1169), computing on the uc2 HIP kernels.

Drop-in for:  VLXLMRConfig, VLXLMRPreTrainedModel (from_pretrained / init_weights),
VLXLMRTextEmbeddings, VLXLMRImageEmbeddings, VLXLMREncoder, VLXLMRModel,
VLXLMRForPretraining, RegionFeatureRegression, RegionClassification, pad_tensor_to_mul.
Every task of VLXLMRForPretraining.forward is provided, including the *-soft tasks and the OT
regulariser of the ITM head.
"""
import copy
import json
import logging
from collections import defaultdict
from io import open

import os

import torch
from torch import nn

from .. import ops
from ..config import cfg as knobs, state
from ..store import compute_dtype_of, mark_all_dirty, store_of
from .layer import (GELU, BertLayer, BertLayerNorm, BertPooler, Linear, RobertaLMHead, VisualRobertaLMHead)

logger = logging.getLogger(__name__)
LayerNorm = BertLayerNorm

# the reference derives this list from the XLM-R tokenizer at import time (model/const_variable.py);
# it only sizes the never-called vis_cls head.  Override before constructing a model if needed.
VALID_XLMR_TOKEN_IDS = list(range(5, 50))


def xlmr_sublayer_loading(state_dict, load_embedding_only=False, load_layer=0):
    """Partial XLM-R loads (model/model.py:24-41): keys that must NOT be loaded are renamed to `not_load.<key>` in
    place, so that the loader reports them as unexpected instead of copying them.  load_embedding_only keeps only
    `roberta.embeddings.*`; load_layer = n keeps encoder layers 0..n."""
    def skipped(key):
        if load_embedding_only:
            return "roberta.embeddings" not in key
        if load_layer:
            return "roberta.encoder" in key and int(key.split(".")[3]) > load_layer
        return False
    if load_layer and not load_embedding_only:
        assert isinstance(load_layer, int) and load_layer > 0
    for key in [k for k in state_dict.keys() if skipped(k)]:
        state_dict["not_load." + key] = state_dict.pop(key)


_CONFIG_FIELDS = (("hidden_size", 768), ("num_hidden_layers", 12), ("num_attention_heads", 12),
                  ("intermediate_size", 3072), ("hidden_act", "gelu"), ("hidden_dropout_prob", 0.1),
                  ("attention_probs_dropout_prob", 0.1), ("max_position_embeddings", 514), ("type_vocab_size", 2),
                  ("initializer_range", 0.02), ("layer_norm_eps", 1e-5), ("pad_token_id", 1))


class VLXLMRConfig(object):
    """Model geometry with the reference's field names and constructors (model/model.py:45-141): an int first
    argument is the vocabulary size and the keywords fill the rest; a str is the path of a JSON file whose keys
    become attributes (config/uc2-base.json)."""

    _POSITIONAL = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "hidden_act",
                   "hidden_dropout_prob", "attention_probs_dropout_prob", "max_position_embeddings", "type_vocab_size",
                   "initializer_range", "output_past", "layer_norm_eps", "pad_token_id")     # the reference's argument order

    def __init__(self, vocab_size_or_config_json_file, *args, **fields):
        src = vocab_size_or_config_json_file
        fields.update(zip(self._POSITIONAL, args))
        fields.pop("output_past", None)                   # accepted and ignored, like the reference
        if isinstance(src, str):
            with open(src, "r", encoding="utf-8") as f:
                self.__dict__.update(json.load(f))
        elif isinstance(src, int):
            unknown = set(fields) - {k for k, _ in _CONFIG_FIELDS}
            if unknown:
                raise TypeError("unexpected config field(s): %s" % sorted(unknown))
            self.vocab_size = src
            for key, default in _CONFIG_FIELDS:
                setattr(self, key, fields.get(key, default))
        else:
            raise ValueError("First argument must be either a vocabulary size (int) or the path to a "
                             "pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = cls(-1)
        config.__dict__.update(json_object)
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def __repr__(self):
        return self.to_json_string()


def _legacy_key(key):
    """TF-era LayerNorm names in old checkpoints: gamma -> weight, beta -> bias (model/model.py:213-226)"""
    if "gamma" in key:
        return key.replace("gamma", "weight")
    if "beta" in key:
        return key.replace("beta", "bias")
    return None


class VLXLMRPreTrainedModel(nn.Module):
    """model/model.py:143-278: weight init and from_pretrained key surgery."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, VLXLMRConfig):
            raise ValueError("Parameter config in `{}(config)` should be an instance of class `VLXLMRConfig`."
                             .format(self.__class__.__name__))
        self.config = config

    def init_weights(self, module):
        """N(0, initializer_range) for Linear / Embedding weights (padding rows included), LayerNorm = (1, 0),
        Linear biases 0 (model/model.py:159-172)"""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def zero_grad(self, set_to_none=True):
        st = store_of(self)
        if st is not None:
            st.zero_grad()
        else:
            super().zero_grad(set_to_none)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        mark_all_dirty()                   # the bf16 compute copies are stale now
        return r

    @classmethod
    def from_pretrained(cls, config_file, state_dict, load_embedding_only=False, load_layer=None, *inputs, **kwargs):
        """Build the model from a JSON config and copy a checkpoint dict into it (model/model.py:174-278):
        legacy LayerNorm key names are translated, partial XLM-R loads rename the skipped keys, a `roberta.bert.`
        prefix in the checkpoint is honoured, missing / unexpected keys are logged, shape errors raise."""
        config = VLXLMRConfig.from_json_file(config_file)
        logger.info("Model config {}".format(config))
        model = cls(config, *inputs, **kwargs)
        if load_embedding_only:
            xlmr_sublayer_loading(state_dict, load_embedding_only=True)
        elif load_layer is not None:
            xlmr_sublayer_loading(state_dict, load_layer=load_layer)
        else:
            for old in [k for k in state_dict.keys() if _legacy_key(k)]:
                state_dict[_legacy_key(old)] = state_dict.pop(old)
        metadata = getattr(state_dict, "_metadata", None)
        sd = state_dict.copy()
        if metadata is not None:
            sd._metadata = metadata
        missing, unexpected, errors = [], [], []
        prefix0 = "roberta.bert." if any(k.startswith("roberta.bert.") for k in sd.keys()) else ""
        stack = [(model, prefix0)]
        while stack:                       # every module copies its own direct parameters / buffers
            module, prefix = stack.pop()
            local = {} if metadata is None else metadata.get(prefix[:-1], {})
            module._load_from_state_dict(sd, prefix, local, True, missing, unexpected, errors)
            stack.extend((child, prefix + name + ".") for name, child in module._modules.items() if child is not None)
        if missing:
            logger.info("Weights of {} not initialized from pretrained model: {}".format(cls.__name__, missing))
        if unexpected:
            logger.info("Weights from pretrained model not used in {}: {}".format(cls.__name__, unexpected))
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(cls.__name__, "\n\t".join(errors)))
        mark_all_dirty()
        return model


def create_position_ids_from_input_ids(input_ids, padding_idx):
    """model/model.py:280-290 on the device: cumsum of non-pad tokens (+ pad id)."""
    ids = input_ids.contiguous()
    out = torch.empty_like(ids)
    B, T = ids.shape
    ops.call("uc2_position_ids", B, T, ops.ptr(ids), int(padding_idx), ops.ptr(out), ops.stream())
    return out


class VLXLMRTextEmbeddings(nn.Module):
    """model/model.py:292-335"""

    def __init__(self, config):
        super().__init__()
        self.padding_idx = config.pad_token_id
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size,
                                                padding_idx=config.pad_token_id)
        self.new_token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids=None, position_ids=None, token_type_ids=None):
        if position_ids is None:
            position_ids = create_position_ids_from_input_ids(input_ids, self.padding_idx)
        elif position_ids.shape != input_ids.shape:
            position_ids = position_ids.expand_as(input_ids)
        e = ops.EmbedTextFn.apply(self, compute_dtype_of(self), input_ids, position_ids, token_type_ids,
                                  self.word_embeddings.weight, self.position_embeddings.weight,
                                  self.new_token_type_embeddings.weight,
                                  -1 if self.word_embeddings.padding_idx is None else self.word_embeddings.padding_idx,
                                  -1 if self.position_embeddings.padding_idx is None else self.position_embeddings.padding_idx)
        p = self.dropout.p if self.training else 0.0
        return self.LayerNorm(e, None, p, 0x7E01, drop_after=True)       # dropout(LayerNorm(.)), model/model.py:331-333


class VLXLMRImageEmbeddings(nn.Module):
    """model/model.py:339-364"""

    def __init__(self, config, img_dim):
        super().__init__()
        self.img_linear = Linear(img_dim, config.hidden_size)
        self.img_layer_norm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.pos_layer_norm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.pos_linear = Linear(7, config.hidden_size)
        self.mask_embedding = nn.Embedding(2, img_dim, padding_idx=0)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, img_feat, img_pos_feat, type_embeddings, img_masks=None):
        """type_embeddings: the new_token_type_embeddings Parameter (image rows use row 1,
        model/model.py:404) or an explicit [B,R,H] tensor when img_type_ids were given."""
        cd = compute_dtype_of(self)
        if img_masks is not None:
            self.mask_embedding.weight.data[0, :].fill_(0)            # model/model.py:354 (in place)
            img_feat = ops.MaskEmbedFn.apply(self, img_feat, img_masks, self.mask_embedding.weight, cd)
        ti = self.img_layer_norm(self.img_linear(ops.cast(img_feat, cd)))
        tp = self.pos_layer_norm(self.pos_linear(ops.cast(img_pos_feat, cd)))
        p = self.dropout.p if self.training else 0.0
        if isinstance(type_embeddings, nn.Parameter):
            # + type row 1 for every region == a constant added to the affine shift of the last LN input
            s = ops.AddRowFn.apply(ti, tp, type_embeddings, 1)
        else:
            s = ti + tp + ops.cast(type_embeddings, cd)
        return self.LayerNorm(s, None, p, 0x7E02, drop_after=True)       # dropout(LayerNorm(.)), model/model.py:361-363


class VLXLMREncoder(nn.Module):
    """model/model.py:366-383"""

    def __init__(self, config):
        super().__init__()
        layer = BertLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(config.num_hidden_layers)])

    def forward(self, input_, attention_mask, output_all_encoded_layers=True):
        all_encoder_layers = []
        hidden_states = input_
        # A ragged token count (the reference's token-bucket batches have a new B x L every step, data/sampler.py:11-59) would take
        # every GEMM of every layer off its planned kernel: in bf16 the layers then run on B L rounded up to whole 256-row tiles --
        # zero rows appended ONCE here, carried from layer to layer as [rows, H], dropped at the end (ops.padded_rows)
        B, L = input_.shape[0], input_.shape[1]
        rows = ops.padded_rows(B * L, input_.dtype) if input_.dim() == 3 and len(self.layer) else B * L
        padded = rows != B * L
        if padded:
            hidden_states = ops.PadRowsFn.apply(hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous(), rows)
        for layer_module in self.layer:
            hidden_states = layer_module(hidden_states, attention_mask)
            if output_all_encoded_layers:
                all_encoder_layers.append(ops.UnpadRowsFn.apply(hidden_states, B, L) if padded else hidden_states)
        if not output_all_encoded_layers:
            all_encoder_layers.append(ops.UnpadRowsFn.apply(hidden_states, B, L) if padded else hidden_states)
        return all_encoder_layers


class VLXLMRModel(VLXLMRPreTrainedModel):
    """model/model.py:385-458"""

    def __init__(self, config, img_dim):
        super().__init__(config)
        self.embeddings = VLXLMRTextEmbeddings(config)
        self.img_embeddings = VLXLMRImageEmbeddings(config, img_dim)
        self.encoder = VLXLMREncoder(config)
        self.pooler = BertPooler(config)
        self.apply(self.init_weights)

    def _compute_txt_embeddings(self, input_ids, position_ids, txt_type_ids=None):
        return self.embeddings(input_ids, position_ids, txt_type_ids)

    def _compute_img_embeddings(self, img_feat, img_pos_feat, img_masks=None, img_type_ids=None):
        if img_type_ids is None:
            type_emb = self.embeddings.new_token_type_embeddings.weight
        else:
            # explicit per-region type ids: an API corner no entry point uses (always None upstream)
            type_emb = nn.functional.embedding(img_type_ids, self.embeddings.new_token_type_embeddings.weight)
        return self.img_embeddings(img_feat, img_pos_feat, type_emb, img_masks)

    def _compute_img_txt_embeddings(self, input_ids, position_ids, img_feat, img_pos_feat, gather_index,
                                    img_masks=None, txt_type_ids=None, img_type_ids=None):
        txt_emb = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
        img_emb = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
        if txt_emb.dtype == img_emb.dtype and txt_emb.shape[0] == img_emb.shape[0]:
            return ops.GatherCatRowsFn.apply(txt_emb, img_emb, gather_index)       # the concatenation is never built
        return ops.GatherRowsFn.apply(torch.cat([txt_emb, img_emb], dim=1), gather_index)

    def forward(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index=None,
                img_masks=None, output_all_encoded_layers=True, txt_type_ids=None, img_type_ids=None):
        st = store_of(self)
        if st.auto_sync and compute_dtype_of(self) == torch.bfloat16:
            st.mark_dirty()
        # additive key mask, fp32 (the parameters are fp32 masters): model/model.py:433-436
        extended_attention_mask = attention_mask.unsqueeze(1).unsqueeze(2).to(dtype=torch.float32)
        extended_attention_mask = (1.0 - extended_attention_mask) * -10000.0
        ops.fp8_new_forward()
        with ops.rng.scope():           # one dropout seed copy for the whole forward; the sites are told apart by ops.rng.site()
            if input_ids is None:
                embedding_output = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
            elif img_feat is None:
                embedding_output = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
            else:
                embedding_output = self._compute_img_txt_embeddings(input_ids, position_ids, img_feat, img_pos_feat,
                                                                    gather_index, img_masks, txt_type_ids, img_type_ids)
            encoded_layers = self.encoder(embedding_output, extended_attention_mask,
                                          output_all_encoded_layers=output_all_encoded_layers)
        if not output_all_encoded_layers:
            encoded_layers = encoded_layers[-1]
        return encoded_layers


def _any_fp8(module):
    """True when a BertLayer under `module` runs its GEMMs on e4m3 operands (store.set_fp8); cached per module"""
    d = module.__dict__
    key = d.get("_uc2_fp8_probe")
    layers = key if key is not None else [m for m in module.modules() if m.__class__.__name__ == "BertLayer"]
    if key is None:
        d["_uc2_fp8_probe"] = layers
    return any(l.__dict__.get("uc2_fp8", False) for l in layers)


def pad_tensor_to_mul(tensor, dim=0, mul=8):
    """model/model.py:1051-1054: returns immediately in the reference (padding disabled)."""
    return tensor, 0


class RegionFeatureRegression(nn.Module):
    """model/model.py:1143-1156: Linear-GELU-LN(1e-12) then F.linear(h, W_img^T, bias) (weight tied to img_linear)"""

    def __init__(self, hidden_size, feat_dim, img_linear_weight):
        super().__init__()
        self.net = nn.Sequential(Linear(hidden_size, hidden_size), GELU(), LayerNorm(hidden_size, eps=1e-12))
        self.weight = img_linear_weight
        self.bias = nn.Parameter(torch.zeros(feat_dim))

    def forward(self, input_):
        hidden = self.net[2](self.net[0](input_, act=ops.EPI_GELU))
        return ops.LinearFn.apply(hidden, self, ops.EPI_NONE, True, self.weight, self.bias)


class RegionClassification(nn.Module):
    """model/model.py:1159-1169"""

    def __init__(self, hidden_size, label_dim):
        super().__init__()
        self.net = nn.Sequential(Linear(hidden_size, hidden_size), GELU(), LayerNorm(hidden_size, eps=1e-12),
                                 Linear(hidden_size, label_dim))

    def forward(self, input_):
        return self.net[3](self.net[2](self.net[0](input_, act=ops.EPI_GELU)))


class VLXLMRForPretraining(VLXLMRPreTrainedModel):
    """model/model.py:460-775: MLM / TLM / VMLM / MRFR / MRC / ITM heads; returns UNREDUCED losses
    (fp32) or raw scores exactly like the reference."""

    def __init__(self, config, img_dim, img_label_dim, nce_temp=1, ot_pos_only=False):
        super().__init__(config)
        self.roberta = VLXLMRModel(config, img_dim)
        self.cls = RobertaLMHead(config, self.roberta.embeddings.word_embeddings.weight)
        self.vis_cls = VisualRobertaLMHead(config, self.roberta.embeddings.word_embeddings.weight,
                                           VALID_XLMR_TOKEN_IDS)
        self.feat_regress = RegionFeatureRegression(config.hidden_size, img_dim,
                                                    self.roberta.img_embeddings.img_linear.weight)
        self.region_classifier = RegionClassification(config.hidden_size, img_label_dim)
        self.itm_output = Linear(config.hidden_size, 2)
        self.ot_pos_only = ot_pos_only
        self.apply(self.init_weights)
        self.vocab_pad = 0

    def pad_vocab(self):
        """model/model.py:486-493: padding is a no-op in the reference; only the tie is re-established."""
        self.cls.decoder.weight = self.roberta.embeddings.word_embeddings.weight
        self.vocab_pad = 0

    # ------------------------------------------------------------------ dispatch
    def forward(self, batch, task, compute_loss=True):
        """model/model.py:495-568.  A training forward of a small micro-batch runs on one of the two accumulation-overlap streams
        (ops.accum_pass: the loop's next forward then runs beside this one's backward); results and call sequence are unchanged."""
        st = store_of(self)                 # one arena for the whole model
        am = batch.get('attn_masks') if hasattr(batch, 'get') else None
        rows = am.numel() if (self.training and torch.is_tensor(am)) else 0
        with ops.accum_pass(st, rows, [v for v in batch.values() if torch.is_tensor(v)] if rows else (),
                            fp8=_any_fp8(self), bf16=compute_dtype_of(self) == torch.bfloat16) as ap:
            return ap.mark(self._forward(batch, task, compute_loss))

    def _forward(self, batch, task, compute_loss=True):
        # fp8 mode: delayed activation scales are kept per task (the gradient magnitudes of two tasks differ by orders of magnitude:
        # their losses average over different counts -- the reference keeps one amp loss scaler per task for the same reason,
        # pretrain.py:462-465)
        state.fp8_tag = (task, bool(compute_loss))
        batch = defaultdict(lambda: None, batch)
        input_ids = batch['input_ids']
        position_ids = batch['position_ids'] if task == 'tlm' else None
        img_feat = batch['img_feat']
        img_pos_feat = batch['img_pos_feat']
        attention_mask = batch['attn_masks']
        gather_index = batch['gather_index']
        if task in ['mlm', 'tlm']:
            return self.forward_mlm(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                                    batch['txt_labels'], compute_loss, n_masked=batch['n_txt_labels'])
        elif task == 'tlm-ni':
            return self.forward_mlm(input_ids, position_ids, None, None, attention_mask, None,
                                    batch['txt_labels'], compute_loss, n_masked=batch['n_txt_labels'])
        elif task in ['mmxlm', 'vmlm']:
            return self.forward_mmxlm(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                                      batch['img_masks'], batch['txt_labels'], compute_loss, n_masked=batch['n_txt_labels'])
        elif task in ["mmxlm-soft", 'vmlm-soft']:
            return self.forward_mmxlm_soft(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                           gather_index, batch['img_masks'], batch['tgt_masks'],
                                           batch['label_targets'], compute_loss)
        elif task == 'mrfr':
            return self.forward_mrfr(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                                     batch['img_masks'], batch['img_mask_tgt'], batch['feat_targets'], compute_loss,
                                     n_masked=batch['n_img_mask_tgt'])
        elif task == 'itm':
            return self.forward_itm(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                                    batch['targets'], batch['ot_inputs'], compute_loss)
        elif task.startswith('mrc'):
            return self.forward_mrc(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                                    batch['img_masks'], batch['img_mask_tgt'], batch['label_targets'], task,
                                    compute_loss, n_masked=batch['n_img_mask_tgt'])
        else:
            raise ValueError('invalid task')

    # ------------------------------------------------------------------ heads
    # Count hints ('n_txt_labels' / 'n_img_mask_tgt' in the batch) are the caller's promise about device data the host never reads
    # back.  A wrong one must not corrupt memory: padding entries of the fixed-size index list are -1, uc2_select_rows gathers them
    # as zero rows / skips them when scattering, their labels become ignore_index; and every mismatch (either direction: a hint
    # that is too small drops masked positions from the loss) is counted on the device in HINT_MISMATCH, which whoever syncs anyway
    # (a logging step, the loader, bench.py) reads with hint_mismatches().  UC2_CHECK_HINTS=1 asserts at once (one sync per call).
    HINT_MISMATCH = {}

    @classmethod
    def hint_mismatches(cls):
        """number of forward calls so far whose count hint disagreed with the mask on the device (one host sync)"""
        return int(sum(int(t.item()) for t in cls.HINT_MISMATCH.values()))

    @classmethod
    def _masked_rows(cls, mask, n_hint=None):
        """flat indices of the set entries of `mask`, in order.  Their number decides the shape of everything downstream, so
        finding it on the device costs a host sync (torch.nonzero; the reference's boolean indexing syncs the same way,
        model/model.py:653-657).  When the batch carries the count (`n_hint`, known to whoever built the labels on the host:
        uc2_amd/data/loader.py::assemble, bench.py) the index list is built without one; entries beyond the real count are -1."""
        flat = mask.reshape(-1)
        if n_hint is None:
            return torch.nonzero(flat, as_tuple=False).view(-1)
        n = int(n_hint)
        cnt = flat.sum()
        if knobs.check_hints:
            assert int(cnt.item()) == n, "batch count hint %d != %d masked entries" % (n, int(cnt.item()))
        key = str(flat.device)
        acc = cls.HINT_MISMATCH.get(key)
        if acc is None:
            acc = cls.HINT_MISMATCH[key] = torch.zeros((), dtype=torch.int64, device=flat.device)
        acc.add_((cnt != n).to(torch.int64))
        try:
            return torch.nonzero_static(flat, size=n, fill_value=-1).view(-1)
        except (NotImplementedError, RuntimeError):         # no device kernel in this torch build: stable sort, still no sync
            order = torch.argsort(~flat.bool(), stable=True)
            if n > order.numel():
                order = torch.cat([order, order.new_full((n - order.numel(),), -1)])
            order = order[:n]
            return torch.where(torch.arange(n, device=flat.device) < cnt, order, order.new_full((), -1))

    def _compute_masked_hidden(self, hidden, mask, n_hint=None, rows=None):
        """model/model.py:653-657: rows of `hidden` where mask is set (row compaction kernel)"""
        H = hidden.size(-1)
        if rows is None:
            rows = self._masked_rows(mask, n_hint)
        return ops.SelectRowsFn.apply(hidden.reshape(-1, H), rows)

    def _masked_labels(self, hidden, txt_labels, n_hint):
        """(hidden[txt_labels != -1], txt_labels[txt_labels != -1]) with one index list for both (model/model.py:583-596)"""
        rows = self._masked_rows(txt_labels != -1, n_hint)
        labels = txt_labels.reshape(-1).index_select(0, rows.clamp_min(0))
        if n_hint is not None:
            labels = torch.where(rows < 0, labels.new_full((), -100), labels)      # padding entries: ignore_index of the MLM loss
        return self._compute_masked_hidden(hidden, None, rows=rows), labels

    def _pad_layer_unpad(self, input_, layer):
        return layer(input_)

    def _mlm_scores_or_loss(self, masked_output, labels_flat, compute_loss):
        z = self.cls.transform(masked_output)
        if compute_loss:
            loss, _ = ops.DecoderCEFn.apply(z, self.cls, self.cls.decoder.weight, self.cls.bias, labels_flat, -100)
            return loss
        return self.cls.decoder(z)

    def forward_mlm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    txt_labels, compute_loss=True, n_masked=None):
        if gather_index is not None:
            sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                           gather_index, output_all_encoded_layers=False)
        else:
            sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                           output_all_encoded_layers=False)
        sequence_output = sequence_output[:, :input_ids.size(1), :]
        masked_output, labels = self._masked_labels(sequence_output.contiguous(), txt_labels, n_masked)
        return self._mlm_scores_or_loss(masked_output, labels, compute_loss)

    def forward_mmxlm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                      img_masks, txt_labels, compute_loss=True, n_masked=None):
        sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                       gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        if txt_labels.size(1) != sequence_output.size(1):   # labels cover the text positions of the joint sequence (model/model.py:615)
            masked_output = self._compute_masked_hidden(sequence_output, txt_labels != -1)
            return self._mlm_scores_or_loss(masked_output, txt_labels[txt_labels != -1], compute_loss)
        masked_output, labels = self._masked_labels(sequence_output, txt_labels, n_masked)
        return self._mlm_scores_or_loss(masked_output, labels, compute_loss)

    def forward_mmxlm_soft(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                           img_masks, tgt_masks, label_targets, compute_loss=True):
        """model/model.py:627-651: MLM head on the masked regions, restricted to the VALID_XLMR_TOKEN_IDS columns
        (computed as a GEMM against those rows of the tied decoder only), KL against the soft labels"""
        sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                       gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        masked_output = self._compute_masked_hidden(sequence_output, tgt_masks)
        z = self.cls.transform(masked_output)
        ids = self._valid_ids(z.device)
        prediction_soft_label = ops.TiedSubsetDecoderFn.apply(z, self.cls, self.cls.decoder.weight, self.cls.bias, ids)
        if compute_loss:
            return ops.KLDivFn.apply(prediction_soft_label, label_targets, prediction_soft_label.shape[-1])
        return prediction_soft_label

    def _valid_ids(self, device):
        ids = self.__dict__.get("_valid_ids_dev")
        if ids is None or ids.device != device or ids.numel() != len(VALID_XLMR_TOKEN_IDS):
            ids = torch.tensor(list(VALID_XLMR_TOKEN_IDS), dtype=torch.long, device=device)
            self.__dict__["_valid_ids_dev"] = ids
        return ids

    def forward_mrfr(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                     img_masks, img_mask_tgt, feat_targets, compute_loss=True, n_masked=None):
        sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                       gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        masked_output = self._compute_masked_hidden(sequence_output, img_mask_tgt, n_masked)
        prediction_feat = self.feat_regress(masked_output)
        if compute_loss:
            return ops.MSEFn.apply(prediction_feat, feat_targets)
        return prediction_feat

    def forward_itm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    targets, ot_inputs, compute_loss=True):
        sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                       gather_index, output_all_encoded_layers=False)
        pooled_output = self.roberta.pooler(sequence_output)
        rank_scores = self.itm_output(pooled_output)
        ot_loss = None
        if ot_inputs is not None:
            # OT regulariser (model/model.py:701-729, model/ot.py): the scatter back to the padded [txt | img] layout,
            # the cosine cost, 50 IPOT iterations and the trace run in one pass of uc2_ot_fwd per batch element
            tl, il = input_ids.size(1), img_feat.size(1)
            ot_dist = ops.OTDistFn.apply(sequence_output, ot_inputs['ot_scatter'], ot_inputs['txt_pad'],
                                         ot_inputs['img_pad'], tl, il, 0.5, 50)
            if self.ot_pos_only:
                ot_loss = ot_dist.masked_select(targets == 1)
            else:
                ot_loss = (ot_dist.masked_select(targets == 1), ot_dist.masked_select(targets == 0))
        if compute_loss:
            itm_loss, _ = ops.CrossEntropyFn.apply(rank_scores, targets, -100, rank_scores.shape[-1])
            return itm_loss, ot_loss
        return rank_scores, ot_loss

    def forward_mrc(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    img_masks, img_mask_tgt, label_targets, task, compute_loss=True, n_masked=None):
        sequence_output = self.roberta(input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                                       gather_index, output_all_encoded_layers=False, img_masks=img_masks)
        masked_output = self._compute_masked_hidden(sequence_output, img_mask_tgt, n_masked)
        prediction_soft_label = self.region_classifier(masked_output)
        if compute_loss:
            if "kl" in task:
                return ops.KLDivFn.apply(prediction_soft_label, label_targets, prediction_soft_label.shape[-1])
            # background class should not be the target (model/model.py:770)
            label_targets = torch.max(label_targets[:, 1:], dim=-1)[1] + 1
            loss, _ = ops.CrossEntropyFn.apply(prediction_soft_label, label_targets, 0,
                                               prediction_soft_label.shape[-1])
            return loss
        return prediction_soft_label
