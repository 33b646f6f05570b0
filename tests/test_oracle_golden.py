"""The CPU oracle (oracle/uc2_oracle.py) pinned against golden vectors produced by the
REFERENCE's own code (tests/golden/make_golden.py).  CPU only."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import specs
from oracle import uc2_oracle as O
from uc2_amd.utils import synth
from util import check_against_golden, golden

TOL = 2e-5        # oracle vs reference, both fp32 on CPU (reference is reproducible to ~2e-6)


def make_weights(shapes):
    W = OrderedDict()
    for n, shp in shapes.items():
        W[n] = synth.det_fill_(n, torch.empty(shp))
    return W


def strip(b):
    return {k: v for k, v in b.items() if not k.startswith("_")}


def cfg_of(geom, drop=0.0):
    return O.Config.make(hidden_dropout_prob=drop, attention_probs_dropout_prob=drop, **geom)


def run_task(W, cfg, batch, task):
    """returns seq, scores, loss, grads -- what make_golden.py recorded for the reference"""
    b = strip(batch)
    seq = O.model_forward(W, cfg, b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"],
                          b["gather_index"], img_masks=b.get("img_masks"))
    scores = O.pretrain_forward(W, cfg, b, task, compute_loss=False)

    def loss_fn(Wg):
        l = O.pretrain_forward(Wg, cfg, b, task, compute_loss=True)
        l = l[0] if isinstance(l, tuple) else l
        loss_fn.loss = l.detach()
        return l.mean()
    _, grads = O.grads_of(loss_fn, W)
    if isinstance(scores, tuple):
        scores = scores[0]
    return seq, scores, loss_fn.loss, grads


CASES = [("tiny8", 8, False, ["itm", "mlm", "mrfr", "mrc", "mrc-kl", "vmlm"]),
         ("tiny8var", 8, True, ["itm", "mlm"]),
         ("tiny64", 64, False, ["itm", "mlm"])]


@pytest.mark.parametrize("tag,B,var,tasks", CASES)
def test_pretrain_tiny(tag, B, var, tasks):
    g = golden("tiny")
    cfg = cfg_of(O.TINY)
    W = make_weights(specs.pretrain_shapes(cfg))
    for task in tasks:
        batch = synth.make_batch(cfg.vocab_size, B, 32, 36, task=task, seed=1, variable_len=var)
        seq, scores, loss, grads = run_task(W, cfg, batch, task)
        key = "%s/%s" % (tag, task)
        check_against_golden(g, key + "/seq", seq, TOL)
        check_against_golden(g, key + "/scores", scores, TOL)
        check_against_golden(g, key + "/loss", loss, TOL * 5)
        if key + "/argmax" in g.files:
            assert np.array_equal(scores.argmax(-1).numpy(), g[key + "/argmax"])       # bit-exact labels
        if task == "itm":
            check_against_golden(g, key + "/pooled", O.pooler(seq, W), TOL)
        n = 0
        for name, gr in grads.items():
            k = "%s/grad/%s" % (key, name)
            if k + "/sum3" in g.files:
                if float(g[k + "/sum3"][2]) < 1e-7:          # mathematically zero (e.g. key.bias): noise only
                    assert gr.norm().item() < 1e-6
                else:
                    check_against_golden(g, k, gr, 2e-4)
                n += 1
        assert n > 40


def test_itm_rank():
    g = golden("tiny")
    cfg = cfg_of(O.TINY)
    W = make_weights(specs.itm_rank_shapes(cfg))
    b = strip(synth.make_batch(1000, 12, 32, 36, task="rank", seed=2, variable_len=True, sample_size=3))
    check_against_golden(g, "rank/scores", O.itm_rank_forward(W, cfg, b, compute_loss=False), TOL)

    def loss_fn(Wg):
        l = O.itm_rank_forward(Wg, cfg, b, margin=0.2)
        loss_fn.loss = l.detach()
        return l.mean()
    _, grads = O.grads_of(loss_fn, W)
    check_against_golden(g, "rank/loss", loss_fn.loss, TOL * 5)
    for name, gr in grads.items():
        k = "rank/grad/%s" % name
        if k + "/sum3" in g.files and float(g[k + "/sum3"][2]) > 1e-7:
            # the hinge is active on few triplets and the row sums cancel heavily: fp32 noise ~1e-3
            check_against_golden(g, k, gr, 5e-3)


def test_gather_index_and_padding():
    g = golden("gather")
    for key in g.files:
        if not key.startswith("gather/"):
            continue
        tls, nbs = [list(map(int, s.split("-"))) for s in key[len("gather/"):].split("_")]
        out_size = max(a + b for a, b in zip(tls, nbs))
        mine = O.get_gather_index(tls, nbs, len(tls), max(tls), out_size)
        assert np.array_equal(mine.numpy(), g[key])


def test_adamw_clip_sched():
    """3 optimizer steps (3 summed micro-batches each), both param groups, clipping, schedule."""
    g = golden("adamw")
    cfg = cfg_of(O.TINY)
    W = make_weights(specs.pretrain_shapes(cfg))
    names = [n for n in W if n not in ()]
    no_decay = [n for n in names if O.is_no_decay(n)]
    assert no_decay == list(g["adamw/no_decay_names"])                 # substring quirk (Q6) pinned
    assert [n for n in names if not O.is_no_decay(n)] == list(g["adamw/decay_names"])
    assert "roberta.img_embeddings.img_layer_norm.weight" not in no_decay
    M = {n: torch.zeros_like(w) for n, w in W.items()}
    V = {n: torch.zeros_like(w) for n, w in W.items()}
    steps = {n: 0 for n in W}
    for step in range(1, 4):
        lr = O.warmup_linear(step, 2, 10) * 4e-5 + 1e-5
        task = ["itm", "mlm", "mrfr"][step - 1]
        acc = {}
        for micro in range(3):
            b = strip(synth.make_batch(1000, 4, 32, 36, task=task, seed=10 * step + micro))

            def loss_fn(Wg):
                l = O.pretrain_forward(Wg, cfg, b, task)
                l = l[0] if isinstance(l, tuple) else l
                return l.mean()
            _, grads = O.grads_of(loss_fn, W)
            for n, gr in grads.items():
                acc[n] = acc.get(n, 0) + gr
        gl = [acc[n] for n in acc]
        gn = O.clip_grad_norm(gl, 5.0 if step > 1 else 0.05)
        assert abs(float(gn) - float(g["adamw/step%d/grad_norm" % step][0])) < 1e-4 * float(gn)
        if task == "mrfr":      # the forward zeroes mask_embedding.weight[0] in place (model/model.py:354);
            W["roberta.img_embeddings.mask_embedding.weight"][0].zero_()   # grads_of() works on copies
        for n in acc:                                   # params without a grad are skipped (Q7)
            steps[n] += 1
            O.adamw_step(W[n], acc[n], M[n], V[n], steps[n], lr, 0.9, 0.98, 1e-6,
                         0.0 if O.is_no_decay(n) else 0.01)
        for key in g.files:
            pre = "adamw/step%d/" % step
            if key.startswith(pre) and key.endswith("/sum3") and "param_norm" not in key and "grad_norm" not in key:
                n = key[len(pre):-len("/sum3")]
                check_against_golden(g, pre + n, W[n], 2e-6)
        tot = torch.sqrt(sum((w.double() ** 2).sum() for w in W.values())).item()
        assert abs(tot - float(g["adamw/step%d/param_norm" % step][0])) < 1e-7 * tot
    st = g["sched/steps"]
    for decay in ("linear", "invsqrt", "constant"):
        mine = [O.get_lr_sched(int(s), 4e-5, 10000, 200000, decay) for s in st]
        assert np.allclose(mine, g["sched/%s" % decay], rtol=1e-12, atol=0)


def test_allreduce_mean():
    ts = [synth.det_normal((1000,), 100 + r) for r in range(4)]
    out = O.allreduce_mean(ts, 1.0)
    assert torch.allclose(out, torch.stack(ts).mean(0), atol=1e-7)


@pytest.mark.slow
def test_pretrain_base_geometry():
    """12L/768H, vocab 250002, B=4 (BASELINE.json configs[1] geometry) -- ~1 min on 8 cores."""
    g = golden("base")
    cfg = cfg_of(O.BASE)
    W = make_weights(specs.pretrain_shapes(cfg))
    for task in ("itm", "mlm"):
        batch = synth.make_batch(cfg.vocab_size, 4, 60, 36, task=task, seed=1)
        seq, scores, loss, grads = run_task(W, cfg, batch, task)
        key = "base4/%s" % task
        check_against_golden(g, key + "/seq", seq, 5e-5)
        check_against_golden(g, key + "/loss", loss, 1e-4)
        assert np.array_equal(scores.argmax(-1).numpy(), g[key + "/argmax"])
        for name in ("roberta.encoder.layer.0.attention.self.query.weight",
                     "roberta.encoder.layer.11.output.dense.weight",
                     "roberta.img_embeddings.img_linear.weight"):
            check_against_golden(g, "%s/grad/%s" % (key, name), grads[name], 5e-4)
