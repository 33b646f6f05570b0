"""diagnostic (not a test): per-parameter bf16 error of one BertLayer against the CPU oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from collections import OrderedDict
import torch
from oracle import uc2_oracle as O
from uc2_amd.model.layer import BertLayer
from uc2_amd.store import set_compute_dtype
from uc2_amd.utils import synth
from test_gpu_model import make_cfg
from util import rel_err, max_rel
for geom, B, L in [(O.TINY, 5, 68), (O.BASE, 2, 96)]:
    for dtype in (torch.float32, torch.bfloat16):
        cfg = make_cfg(geom); layer = BertLayer(cfg); synth.det_init_(layer)
        W = OrderedDict((n, p.detach().clone()) for n, p in layer.named_parameters())
        layer.to("cuda").train(); set_compute_dtype(layer, dtype)
        H = cfg.hidden_size
        x = synth.det_normal((B, L, H), 11); am = torch.ones(B, L, dtype=torch.long); am[0, L-7:] = 0
        ext = O.extended_mask(am); dy = synth.det_normal((B, L, H), 12)
        xo = x.clone().requires_grad_(True)
        Wg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in W.items())
        yo = O.bert_layer(xo, ext, Wg, "", cfg.num_attention_heads); yo.backward(dy)
        xd = x.to("cuda").to(dtype).requires_grad_(True)
        yd = layer(xd, ext.to("cuda")); yd.backward(dy.to("cuda").to(dtype))
        print("H=%d %s: y max_rel %.3e l2 %.3e | dx l2 %.3e" % (H, dtype, max_rel(yd.float().cpu(), yo.detach()), rel_err(yd.float().cpu(), yo.detach()), rel_err(xd.grad.float().cpu(), xo.grad)))
        for n, p in layer.named_parameters():
            print("   %-40s l2 %.3e  (ref norm %.3e)" % (n, rel_err(p.grad.cpu(), Wg[n].grad), Wg[n].grad.norm().item()))
