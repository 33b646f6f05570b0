"""debug: padded rows x native layer route, dropout on -- which combination changes the loss?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import uc2_oracle as O
from uc2_amd import ops
from uc2_amd.config import cfg as knobs
from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
from uc2_amd.store import set_compute_dtype
from uc2_amd.utils import synth
DEV = "cuda"
geom = dict(O.BASE, num_hidden_layers=int(sys.argv[1]) if len(sys.argv) > 1 else 1, vocab_size=2000)
d = dict(hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514,
         type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
d.update(geom)
batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth.make_batch(2000, 56, 40, 20, task="itm", seed=31, variable_len=True).items() if not k.startswith("_")}
res = {}
for pa in (0.0, 0.1):
  for ph in (0.0, 0.1):
    for pad in (False, True):
        for native in (False, True):
            knobs.pad_rows, knobs.native_layer = pad, native
            dd = dict(d, hidden_dropout_prob=ph, attention_probs_dropout_prob=pa)
            model = VLXLMRForPretraining(VLXLMRConfig.from_dict(dd), img_dim=2048, img_label_dim=1601)
            synth.det_init_(model)
            model.to(DEV).train()
            set_compute_dtype(model, torch.bfloat16)
            ops.rng.manual_seed(99, DEV)
            loss = model(batch, "itm", compute_loss=True)
            loss = (loss[0] if isinstance(loss, tuple) else loss).detach().float().clone()
            res[(pa, ph, pad, native)] = loss
            del model
    ref = res[(pa, ph, False, False)]
    for pad in (False, True):
        for native in (False, True):
            l = res[(pa, ph, pad, native)]
            print("p_attn %.1f p_hidden %.1f pad %d native %d: max |dloss| vs (pad 0, native 0) %.3e   loss[:4] %s" % (pa, ph, pad, native, float((l - ref).abs().max()), [round(float(v), 4) for v in l[:4]]), flush=True)
