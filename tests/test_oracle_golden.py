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


def _check_grads(g, key, grads, tol=2e-4, min_n=40):
    n = 0
    for name, gr in grads.items():
        k = "%s/grad/%s" % (key, name)
        if k + "/sum3" in g.files:
            if name.endswith("key.bias"):     # mathematically zero (softmax is shift invariant): rounding noise only
                qb = float(g[k.replace("key.bias", "query.bias") + "/sum3"][2])
                assert gr.norm().item() < 1e-3 * qb and float(g[k + "/sum3"][2]) < 1e-3 * qb
            elif float(g[k + "/sum3"][2]) < 1e-7:
                assert gr.norm().item() < 1e-6
            else:
                check_against_golden(g, k, gr, tol)
            n += 1
    assert n > min_n, n


def test_more_tasks_and_branches():
    """round-2 fixtures (tests/golden/golden_more.npz): tlm (batch position_ids), tlm-ni (text only), vmlm-soft,
    ITM + OT regulariser, the text-only / image-only encoder branches, all-layer output"""
    g = golden("more")
    cfg = cfg_of(O.TINY)
    W = make_weights(specs.pretrain_shapes(cfg))
    for task in ("tlm", "tlm-ni"):
        b = strip(synth.make_batch(1000, 8, 32, 36, task=task, seed=1, variable_len=True))
        key = "tiny8var/%s" % task
        assert np.array_equal(b["position_ids"].numpy(), g[key + "/position_ids"])
        scores = O.pretrain_forward(W, cfg, b, task, compute_loss=False)
        assert np.array_equal(scores.argmax(-1).numpy(), g[key + "/argmax"])
        check_against_golden(g, key + "/scores", scores, TOL)

        def loss_fn(Wg):
            l = O.pretrain_forward(Wg, cfg, b, task)
            loss_fn.loss = l.detach()
            return l.mean()
        _, grads = O.grads_of(loss_fn, W)
        check_against_golden(g, key + "/loss", loss_fn.loss, TOL * 5)
        _check_grads(g, key, grads)
    # soft labels
    b = strip(synth.make_batch(1000, 8, 32, 36, task="vmlm-soft", seed=1, n_soft=len(O.VALID_XLMR_TOKEN_IDS)))
    key = "tiny8/vmlm-soft"
    check_against_golden(g, key + "/scores", O.pretrain_forward(W, cfg, b, "vmlm-soft", compute_loss=False), TOL)

    def loss_fn(Wg):
        l = O.pretrain_forward(Wg, cfg, b, "vmlm-soft")
        loss_fn.loss = l.detach()
        return 1000 * l.mean()
    _, grads = O.grads_of(loss_fn, W)
    check_against_golden(g, key + "/loss", loss_fn.loss, 1e-4)
    _check_grads(g, key, grads)
    W["roberta.img_embeddings.mask_embedding.weight"][0].zero_()      # model/model.py:354 zeroes it in place in the reference run
    # ITM + OT
    b = strip(synth.make_batch(1000, 8, 32, 36, task="itm", seed=1, variable_len=True, ot=True))
    for pos_only in (False, True):
        key = "tiny8var/itm-ot%s" % ("-pos" if pos_only else "")

        def loss_fn(Wg):
            l, ot = O.pretrain_forward(Wg, cfg, b, "itm", ot_pos_only=pos_only)
            loss_fn.out = (l.detach(), ot)
            otl = ot.mean() if pos_only else (ot[0].sum() - ot[1].sum()) / (ot[0].size(0) + ot[1].size(0))
            return l.mean() + 0.1 * otl
        _, grads = O.grads_of(loss_fn, W)
        l, ot = loss_fn.out
        check_against_golden(g, key + "/loss", l, TOL * 5)
        if pos_only:
            check_against_golden(g, key + "/ot", ot.detach(), 1e-4)
        else:
            check_against_golden(g, key + "/ot_pos", ot[0].detach(), 1e-4)
            check_against_golden(g, key + "/ot_neg", ot[1].detach(), 1e-4)
        _check_grads(g, key, grads, tol=5e-4)
    # text-only / image-only branches, all layers
    full = synth.make_batch(1000, 8, 32, 36, task="mrfr", seed=3, variable_len=True)
    b = strip(full)
    T, R = b["input_ids"].shape[1], b["img_feat"].shape[1]
    am_t = (torch.arange(T).unsqueeze(0) < torch.tensor(full["_txt_lens"]).unsqueeze(1)).long()
    am_i = (torch.arange(R).unsqueeze(0) < torch.tensor(full["_num_bbs"]).unsqueeze(1)).long()
    check_against_golden(g, "txtonly/seq", O.model_forward(W, cfg, b["input_ids"], None, None, None, am_t), TOL)
    check_against_golden(g, "imgonly/seq", O.model_forward(W, cfg, None, None, b["img_feat"], b["img_pos_feat"], am_i,
                                                           img_masks=b["img_masks"]), TOL)
    _, grads = O.grads_of(lambda Wg: (O.model_forward(Wg, cfg, b["input_ids"], None, None, None, am_t)
                                      * synth.det_normal((8, T, 128), 55)).sum(), W)
    _check_grads(g, "txtonly", grads, min_n=30)
    _, grads = O.grads_of(lambda Wg: (O.model_forward(Wg, cfg, None, None, b["img_feat"], b["img_pos_feat"], am_i,
                                                      img_masks=b["img_masks"]) * synth.det_normal((8, R, 128), 56)).sum(), W)
    _check_grads(g, "imgonly", grads, min_n=30)
    layers = O.model_forward(W, cfg, b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"], b["gather_index"],
                             img_masks=b["img_masks"], output_all_encoded_layers=True)
    check_against_golden(g, "alllayers/0", layers[0], TOL)
    check_against_golden(g, "alllayers/1", layers[1], TOL)


def test_multihead_attention_variants():
    """model/attention.py: packed self-attention with key padding (+ gradients) and the general form
    (separate key/value inputs, additive attn_mask), against the reference's outputs and gradients"""
    g = golden("more")
    g0 = golden("mha")
    E, nh, L, N, S = 128, 4, 10, 3, 7
    names = {"in_proj_weight": (3 * E, E), "in_proj_bias": (3 * E,), "out_proj.weight": (E, E), "out_proj.bias": (E,)}
    W = make_weights(names)
    q = synth.det_normal((L, N, E), 77)
    kpm = torch.zeros(N, L, dtype=torch.bool)
    kpm[1, 7:] = True
    kpm[2, 4:] = True
    o, w = O.multi_head_attention(q, q, q, W, nh, key_padding_mask=kpm)
    check_against_golden(g0, "mha/out", o, TOL)
    check_against_golden(g0, "mha/weights", w, TOL)
    qg = q.clone().requires_grad_(True)
    Wg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in W.items())
    o, _ = O.multi_head_attention(qg, qg, qg, Wg, nh, key_padding_mask=kpm)
    (o * synth.det_normal((L, N, E), 78)).sum().backward()
    check_against_golden(g, "mha/dq", qg.grad, 1e-4)
    for n in names:
        check_against_golden(g, "mha/grad/" + n, Wg[n].grad, 1e-4)
    q2, k2, v2 = [synth.det_normal(shp, sd).requires_grad_(True) for shp, sd in (((L, N, E), 80), ((S, N, E), 81), ((S, N, E), 82))]
    amask = synth.det_normal((L, S), 83)
    kpm2 = torch.zeros(N, S, dtype=torch.bool)
    kpm2[2, 5:] = True
    Wg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in W.items())
    o2, w2 = O.multi_head_attention(q2, k2, v2, Wg, nh, key_padding_mask=kpm2, attn_mask=amask)
    check_against_golden(g, "mha_cross/out", o2, TOL)
    check_against_golden(g, "mha_cross/weights", w2, TOL)
    (o2 * synth.det_normal((L, N, E), 84)).sum().backward()
    for key, t in (("dq", q2), ("dk", k2), ("dv", v2)):
        check_against_golden(g, "mha_cross/" + key, t.grad, 1e-4)
    for n in names:
        check_against_golden(g, "mha_cross/grad/" + n, Wg[n].grad, 1e-4)


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


@pytest.mark.slow
def test_pretrain_base_geometry_mix_tasks():
    """round 4: the MRM / VMLM / TLM tasks of the pretrain mix (BASELINE.json configs[2]) at the BASE geometry, B = 4, variable
    lengths, against the reference's own outputs (tests/golden/golden_base_tasks.npz, make_golden.py::case_base_tasks)"""
    g = golden("base_tasks")
    cfg = cfg_of(O.BASE)
    W = make_weights(specs.pretrain_shapes(cfg))
    for task in ("vmlm", "tlm", "mrfr", "mrc", "mrc-kl"):
        batch = synth.make_batch(cfg.vocab_size, 4, 60, 36, task=task, seed=1, variable_len=True)
        key = "base4var/%s" % task
        if task == "tlm":
            assert np.array_equal(batch["position_ids"].numpy(), g[key + "/position_ids"])
        b = strip(batch)
        with torch.no_grad():
            scores = O.pretrain_forward(W, cfg, b, task, compute_loss=False)

        def loss_fn(Wg):
            l = O.pretrain_forward(Wg, cfg, b, task, compute_loss=True)
            loss_fn.loss = l.detach()
            return l.mean()
        names = {"roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.11.output.dense.weight",
                 "roberta.img_embeddings.img_linear.weight", "roberta.embeddings.LayerNorm.weight"}
        _, grads = O.grads_of(loss_fn, W, names=names)
        check_against_golden(g, key + "/scores", scores, 1e-4)
        check_against_golden(g, key + "/loss", loss_fn.loss, 1e-4)
        if key + "/argmax" in g.files:
            assert np.array_equal(scores.argmax(-1).numpy(), g[key + "/argmax"])
        for name in sorted(names):
            check_against_golden(g, "%s/grad/%s" % (key, name), grads[name], 5e-4)
