"""Image-text retrieval finetune model (reference model/itm.py:12-55) on the uc2 HIP kernels."""
from collections import defaultdict

from .. import ops
from ..store import mark_all_dirty, store_of
from .layer import Linear
from .model import VLXLMRModel, VLXLMRPreTrainedModel


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
        store_of(self)
        batch = defaultdict(lambda: None, batch)
        sequence_output = self.roberta(batch['input_ids'], None, batch['img_feat'], batch['img_pos_feat'],
                                       batch['attn_masks'], batch['gather_index'],
                                       output_all_encoded_layers=False)
        pooled_output = self.roberta.pooler(sequence_output)
        rank_scores = self.rank_output(pooled_output)
        if compute_loss:
            return ops.TripletFn.apply(rank_scores, batch['sample_size'], self.margin)
        return rank_scores
