"""debug: retrieval model, padded vs ragged rows, same model instance: losses and gradients"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import uc2_oracle as O
from uc2_amd import ops
from uc2_amd.config import cfg as knobs
from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
from uc2_amd.model.model import VLXLMRConfig
from uc2_amd.store import set_compute_dtype
from uc2_amd.utils import synth
from util import rel_err
DEV = "cuda"
geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
d = dict(hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=514,
         type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
d.update(geom)
b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synth.make_batch(2000, 30, 40, 50, task="itm", seed=5, variable_len=True).items() if not k.startswith("_")}
b.pop("targets", None); b["sample_size"] = 3
print("B x L", tuple(b["attn_masks"].shape))
for dtype in (torch.float32, torch.bfloat16):
    model = VLXLMRForImageTextRetrieval(VLXLMRConfig.from_dict(d), img_dim=2048)
    synth.det_init_(model); model.to(DEV).train(); set_compute_dtype(model, dtype)
    out = {}
    for tag, pad, native in (("ragged", False, True), ("ragged-nonative", False, False), ("padded", True, True), ("ragged2", False, True)):
        knobs.pad_rows, knobs.native_layer = pad, native
        model.zero_grad()
        loss = model(b, compute_loss=True)
        loss.mean().backward()
        ops.join_side_streams(); torch.cuda.synchronize()
        out[tag] = (loss.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
    for tag in ("ragged-nonative", "padded", "ragged2"):
        l, g = out[tag]; l0, g0 = out["ragged"]
        errs = sorted(((rel_err(g[n], g0[n]), n) for n in g0 if g0[n].norm() > 1e-7), reverse=True)[:4]
        print(dtype, tag, "vs ragged: max |dloss| %.3e" % float((l - l0).abs().max()), " worst grads:", [(round(e, 5), n.replace("roberta.", "")) for e, n in errs], flush=True)
