"""Image-text retrieval finetune models (reference model/itm.py:12-55, and the in-model hard-negative miner the
reference ships for its Uniter twin, model/itm.py:105-186) on the uc2 HIP kernels."""
from collections import defaultdict

import torch

from .. import ops
from ..store import compute_dtype_of, mark_all_dirty, store_of
from .layer import Linear
from .model import VLXLMRModel, VLXLMRPreTrainedModel, _any_fp8


class VLXLMRForImageTextRetrieval(VLXLMRPreTrainedModel):
    """encoder -> pooler -> rank_output (H -> 1) -> sigmoid -> triplet margin loss"""

    def __init__(self, config, img_dim, margin=0.2):
        super().__init__(config)
        self.roberta = VLXLMRModel(config, img_dim)
        self.itm_output = Linear(config.hidden_size, 2)
        self.rank_output = Linear(config.hidden_size, 1)
        self.margin = margin
        self.apply(self.init_weights)

    def init_output(self):
        """model/itm.py:23-26: copy row 1 of the pretrain ITM head (values copied; the arena keeps
        rank_output's own storage, so later updates of itm_output do not alias)."""
        self.rank_output.weight.data.copy_(self.itm_output.weight.data[1:, :])
        self.rank_output.bias.data.copy_(self.itm_output.bias.data[1:])
        mark_all_dirty()

    def forward(self, batch, compute_loss=True):
        """model/itm.py:28-55 (training forwards of small micro-batches on the accumulation-overlap streams, ops.accum_pass)"""
        st = store_of(self)
        am = batch.get('attn_masks') if hasattr(batch, 'get') else None
        rows = am.numel() if (self.training and torch.is_tensor(am)) else 0
        with ops.accum_pass(st, rows, [v for v in batch.values() if torch.is_tensor(v)] if rows else (),
                            fp8=_any_fp8(self), bf16=compute_dtype_of(self) == torch.bfloat16) as ap:
            return ap.mark(self._forward(batch, compute_loss))

    def _forward(self, batch, compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        sequence_output = self.roberta(batch['input_ids'], None, batch['img_feat'], batch['img_pos_feat'],
                                       batch['attn_masks'], batch['gather_index'],
                                       output_all_encoded_layers=False)
        pooled_output = self.roberta.pooler(sequence_output)
        rank_scores = self.rank_output(pooled_output)
        if compute_loss:
            return ops.TripletFn.apply(rank_scores, batch['sample_size'], self.margin)
        return rank_scores


class VLXLMRForImageTextRetrievalHardNeg(VLXLMRForImageTextRetrieval):
    """The hard-negative pattern of BASELINE.json configs[3] on the VL-XLM-R retrieval model.  The reference implements
    it for its Uniter twin only (UniterForImageTextRetrievalHardNeg, model/itm.py:105-186; the script-level miner of
    itm.py is dead code): with one positive followed by N candidates sharing the text (sample_from='t')
    or the image ('i'), score every pair WITHOUT autograd state in eval mode (forward-only kernels, nothing saved),
    keep the positive + the `hard_size` best-scoring negatives (on-device top-k), and run the training forward on that
    sub-batch only (sample_size = hard_size + 1 -> triplet loss)."""

    def __init__(self, config, img_dim, margin=0.2, hard_size=16):
        super().__init__(config, img_dim, margin)
        self.hard_size = hard_size

    def forward(self, batch, sample_from='t', compute_loss=True):
        batch = dict(batch)
        n = batch['attn_masks'].size(0)
        if sample_from == 't':                             # same text for all pairs
            if batch['input_ids'].size(0) == 1:
                batch['input_ids'] = batch['input_ids'].expand(n, -1)
        elif sample_from == 'i':                           # same image for all pairs
            if batch['img_feat'].size(0) == 1:
                batch['img_feat'] = batch['img_feat'].expand(n, -1, -1)
            if batch['img_pos_feat'].size(0) == 1:
                batch['img_pos_feat'] = batch['img_pos_feat'].expand(n, -1, -1)
        else:
            raise ValueError()
        if self.training and compute_loss:
            with torch.no_grad():
                self.eval()
                scores = super().forward(batch, compute_loss=False)
                hard_batch = self._get_hard_batch(batch, scores, sample_from)
                self.train()
            return super().forward(hard_batch, compute_loss=True)
        return super().forward(batch, compute_loss)

    def _get_hard_batch(self, batch, scores, sample_from='t'):
        batch = defaultdict(lambda: None, batch)
        k = self.hard_size
        # the first example is the positive; top-k over the rest (model/itm.py:146-151)
        hard = scores.reshape(-1)[1:].topk(k, sorted=False)[1] + 1
        idx = torch.cat([torch.zeros(1, dtype=torch.long, device=hard.device), hard])
        attn = batch['attn_masks'].index_select(0, idx)
        gather = batch['gather_index'].index_select(0, idx)
        pos = batch['position_ids']
        if pos is not None and pos.size(0) != 1:
            pos = pos[:k + 1]
        input_ids, img_feat, img_pos_feat = batch['input_ids'], batch['img_feat'], batch['img_pos_feat']
        if sample_from == 't':
            max_len = int(attn.sum(dim=1).max().item())    # cut to minimum padding
            max_i = max_len - input_ids.size(1)
            attn, gather = attn[:, :max_len], gather[:, :max_len]
            img_feat = img_feat.index_select(0, idx)[:, :max_i, :]
            img_pos_feat = img_pos_feat.index_select(0, idx)[:, :max_i, :]
            input_ids = input_ids[:k + 1]
        else:
            input_ids = input_ids.index_select(0, idx)
            img_feat, img_pos_feat = img_feat[:k + 1], img_pos_feat[:k + 1]
        return {'sample_size': k + 1, 'input_ids': input_ids, 'position_ids': pos, 'img_feat': img_feat,
                'img_pos_feat': img_pos_feat, 'attn_masks': attn, 'gather_index': gather}
