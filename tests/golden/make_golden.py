#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference; no GPU).  It imports the
reference's own modules -- model/model.py, model/layer.py, model/itm.py,
optim/adamw.py, optim/misc.py, optim/sched.py, data/data.py -- behind the three
inert shims of SURVEY.md Appendix B (apex.FusedLayerNorm := torch.nn.LayerNorm, a
bare `model` package, a stub `model.const_variable`), fills them with the
closed-form weights of uc2_amd.utils.synth, runs them on synth batches with
dropout = 0 and stores *outputs only* (numbers; no reference source or pickled
reference objects) as .npz files.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py tiny base  # selected cases
"""
import importlib
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from uc2_amd.utils import synth  # noqa: E402

REF = "/root/reference"
VALID_IDS = list(range(5, 50))


def install_shims():
    apex = types.ModuleType("apex")
    norm = types.ModuleType("apex.normalization")
    fln = types.ModuleType("apex.normalization.fused_layer_norm")
    fln.FusedLayerNorm = torch.nn.LayerNorm
    apex.normalization = norm
    norm.fused_layer_norm = fln
    sys.modules.update({"apex": apex, "apex.normalization": norm,
                        "apex.normalization.fused_layer_norm": fln})
    m = types.ModuleType("model")
    m.__path__ = [REF + "/model"]
    sys.modules["model"] = m
    cv = types.ModuleType("model.const_variable")
    cv.XLMR_TOKER = None
    cv.LABEL2TOKEN_MATRIX = None
    cv.VALID_XLMR_TOKEN_IDS = VALID_IDS
    sys.modules["model.const_variable"] = cv
    for name in ("horovod", "horovod.torch", "lmdb", "lz4", "lz4.frame", "msgpack",
                 "msgpack_numpy", "toolz", "toolz.sandbox", "cytoolz"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["horovod"].torch = sys.modules["horovod.torch"]
    sys.modules["lz4"].frame = sys.modules["lz4.frame"]
    sys.modules["lz4.frame"].compress = sys.modules["lz4.frame"].decompress = lambda x: x
    sys.modules["msgpack_numpy"].patch = lambda: None
    sys.modules["toolz.sandbox"].unzip = lambda x: zip(*x)
    sys.modules["toolz"].sandbox = sys.modules["toolz.sandbox"]
    sys.modules["cytoolz"].concat = lambda x: [z for y in x for z in y]


def load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_config(mm, geom, drop=0.0):
    d = dict(vocab_size=geom["vocab_size"], hidden_size=geom["hidden_size"],
             num_hidden_layers=geom["num_hidden_layers"],
             num_attention_heads=geom["num_attention_heads"],
             intermediate_size=geom["intermediate_size"], hidden_act="gelu",
             hidden_dropout_prob=drop, attention_probs_dropout_prob=drop,
             max_position_embeddings=514, type_vocab_size=2, initializer_range=0.02,
             layer_norm_eps=1e-5, pad_token_id=1)
    return mm.VLXLMRConfig.from_dict(d)


TINY = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
            intermediate_size=512)
BASE = dict(vocab_size=250002, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
            intermediate_size=3072)

GRAD_FULL = ["roberta.encoder.layer.0.attention.self.query.weight",
             "roberta.encoder.layer.0.attention.self.query.bias",
             "roberta.encoder.layer.0.attention.output.LayerNorm.weight",
             "roberta.encoder.layer.0.attention.output.LayerNorm.bias",
             "roberta.encoder.layer.1.output.LayerNorm.weight",
             "roberta.encoder.layer.1.output.LayerNorm.bias",
             "roberta.embeddings.LayerNorm.weight", "roberta.img_embeddings.pos_linear.weight",
             "roberta.pooler.dense.bias", "itm_output.weight", "rank_output.weight"]


def summarize(t):
    t = t.detach().double().flatten()
    idx = synth.slice_idx(t.numel())
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()]), \
        t[idx].float().numpy()


def put(out, key, t, full=False):
    s, sl = summarize(t)             # slice indices: synth.slice_idx(n)
    out[key + "/sum3"] = s
    out[key + "/slice"] = sl
    if full:
        out[key + "/full"] = t.detach().float().clone().numpy()   # clone: params are updated in place later


def grads_to(out, prefix, model, full_names=GRAD_FULL):
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        put(out, "%s/grad/%s" % (prefix, n), p.grad, full=(n in full_names and p.numel() <= 20000))


def strip(batch):
    return {k: v for k, v in batch.items() if not k.startswith("_")}


def run_pretrain_case(mm, geom, B, T, R, tasks, tag, out, full_hidden, variable_len=False):
    cfg = ref_config(mm, geom)
    model = mm.VLXLMRForPretraining(cfg, img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.train()                                   # dropout p = 0 -> deterministic
    for task in tasks:
        batch = synth.make_batch(geom["vocab_size"], B, T, R, task=task, seed=1, variable_len=variable_len)
        b = strip(batch)
        key = "%s/%s" % (tag, task)
        # encoder output itself
        seq = model.roberta(b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"],
                            b["gather_index"], img_masks=b.get("img_masks"),
                            output_all_encoded_layers=False)
        put(out, key + "/seq", seq, full=full_hidden)
        model.zero_grad()
        if task == "itm":
            loss, _ = model(b, task, compute_loss=True)
            scores, _ = model(b, task, compute_loss=False)
            put(out, key + "/scores", scores, full=True)
            out[key + "/argmax"] = scores.argmax(-1).numpy()
            put(out, key + "/pooled", model.roberta.pooler(seq), full=full_hidden)
        else:
            loss = model(b, task, compute_loss=True)
            scores = model(b, task, compute_loss=False)
            if task in ("mlm", "vmlm"):
                out[key + "/argmax"] = scores.argmax(-1).numpy()
                put(out, key + "/scores", scores)
            else:
                put(out, key + "/scores", scores)
        put(out, key + "/loss", loss, full=loss.numel() <= 70000 and task not in ("mrfr", "mrc-kl"))
        loss.mean().backward()
        grads_to(out, key, model)
        print("  %s: loss.mean=%.6f" % (key, loss.mean().item()))
    return model


def case_tiny(mm, mi, out):
    # full tensors at B=8, fixed and variable length
    run_pretrain_case(mm, TINY, 8, 32, 36, ["itm", "mlm", "mrfr", "mrc", "mrc-kl", "vmlm"], "tiny8", out, True)
    run_pretrain_case(mm, TINY, 8, 32, 36, ["itm", "mlm"], "tiny8var", out, True, variable_len=True)
    # BASELINE.json configs[0]: 64 pairs, 36 regions, 32 tokens, MLM+ITM
    run_pretrain_case(mm, TINY, 64, 32, 36, ["itm", "mlm"], "tiny64", out, False)
    # retrieval finetune model (model/itm.py:12-55), 4 triplets
    cfg = ref_config(mm, TINY)
    model = mi.VLXLMRForImageTextRetrieval(cfg, img_dim=2048, margin=0.2)
    synth.det_init_(model)
    model.train()
    batch = strip(synth.make_batch(1000, 12, 32, 36, task="rank", seed=2, variable_len=True, sample_size=3))
    loss = model(batch, compute_loss=True)
    scores = model(batch, compute_loss=False)
    put(out, "rank/loss", loss, full=True)
    put(out, "rank/scores", scores, full=True)
    loss.mean().backward()
    grads_to(out, "rank", model)
    print("  rank: loss.mean=%.6f" % loss.mean().item())


def case_gather(out):
    """pin the padding / gather_index contract against the reference's own helper."""
    dd = load_file("ref_data_data", REF + "/data/data.py")
    for (tls, nbs) in [([5, 3], [4, 2]), ([8, 8, 8, 6], [6, 4, 4, 3])]:
        bs = len(tls)
        out_size = max(a + b for a, b in zip(tls, nbs))
        gi = dd.get_gather_index(tls, nbs, bs, max(tls), out_size)
        mine = synth._gather_index(tls, nbs, bs, max(tls), out_size)
        assert torch.equal(gi, mine)
        out["gather/%s_%s" % ("-".join(map(str, tls)), "-".join(map(str, nbs)))] = gi.numpy()
    t = [torch.arange(6.).view(3, 2), torch.arange(10.).view(5, 2)]
    out["pad_tensors/out"] = dd.pad_tensors(t, [3, 5]).numpy()


def case_adamw(mm, out):
    """AdamW.step + build_optimizer grouping + schedule + clip (optim/*.py)."""
    adamw = load_file("ref_adamw", REF + "/optim/adamw.py")
    sched = load_file("ref_sched", REF + "/optim/sched.py")
    cfg = ref_config(mm, TINY)
    model = mm.VLXLMRForPretraining(cfg, img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.train()
    named = list(model.named_parameters())
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']      # optim/misc.py:11
    g0 = [(n, p) for n, p in named if not any(nd in n for nd in no_decay)]
    g1 = [(n, p) for n, p in named if any(nd in n for nd in no_decay)]
    out["adamw/no_decay_names"] = np.array([n for n, _ in g1])
    out["adamw/decay_names"] = np.array([n for n, _ in g0])
    opt = adamw.AdamW([{"params": [p for _, p in g0], "weight_decay": 0.01},
                       {"params": [p for _, p in g1], "weight_decay": 0.0}],
                      lr=4e-5, betas=(0.9, 0.98))
    # torch >= 2 dropped the (Number, Tensor) overloads adamw.py:77-78,89,101 use; they are
    # still accepted (deprecated) in 2.10 -- checked by running the step below.
    watch = ["roberta.encoder.layer.0.attention.self.query.weight",
             "roberta.encoder.layer.0.attention.self.query.bias",
             "roberta.img_embeddings.img_layer_norm.weight",        # decayed (Q6)
             "roberta.embeddings.LayerNorm.weight",                 # not decayed
             "cls.layer_norm.weight", "itm_output.weight"]
    pd = dict(named)
    for step in range(1, 4):
        lr = sched.warmup_linear(step, 2, 10) * 4e-5 + 1e-5
        for g in opt.param_groups:
            g["lr"] = lr
        for task in (["itm", "mlm", "mrfr"][step - 1],):
            # 3 accumulation micro-steps, summed (pretrain.py:553-559)
            model.zero_grad()
            for micro in range(3):
                b = strip(synth.make_batch(1000, 4, 32, 36, task=task, seed=10 * step + micro))
                loss = model(b, task, compute_loss=True)
                loss = loss[0] if isinstance(loss, tuple) else loss
                loss.mean().backward()
        gn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0 if step > 1 else 0.05)
        out["adamw/step%d/grad_norm" % step] = np.array([float(gn)])
        opt.step()
        for n in watch:
            put(out, "adamw/step%d/%s" % (step, n), pd[n].data, full=pd[n].numel() <= 20000)
        tot = torch.sqrt(sum((p.data.double() ** 2).sum() for p in model.parameters()))
        out["adamw/step%d/param_norm" % step] = np.array([tot.item()])
        print("  adamw step %d: grad_norm %.6f param_norm %.9f" % (step, float(gn), tot.item()))

    class O:
        pass
    o = O()
    o.learning_rate, o.warmup_steps, o.num_train_steps = 4e-5, 10000, 200000
    steps = [0, 1, 9999, 10000, 100000, 200000, 200001]
    for decay in ("linear", "invsqrt", "constant"):
        o.decay = decay
        out["sched/%s" % decay] = np.array([sched.get_lr_sched(s, o) for s in steps])
    out["sched/steps"] = np.array(steps)


def case_mha(out):
    """model/attention.py MultiheadAttention (NLVR2 API surface)."""
    att = importlib.import_module("model.attention")
    E, nh, L, N = 128, 4, 10, 3
    m = att.MultiheadAttention(E, nh, dropout=0.0)
    synth.det_init_(m)
    q = synth.det_normal((L, N, E), 77)
    kpm = torch.zeros(N, L, dtype=torch.bool)
    kpm[1, 7:] = True
    kpm[2, 4:] = True
    o, w = m(q, q, q, key_padding_mask=kpm)
    put(out, "mha/out", o, full=True)
    put(out, "mha/weights", w, full=True)


def case_base(mm, out):
    run_pretrain_case(mm, BASE, 4, 60, 36, ["itm", "mlm"], "base4", out, False)


def case_base_tasks(mm, out):
    """round 4: the MRM / VMLM / TLM tasks of the pretrain mix (BASELINE.json configs[2]; config/uc2_pretrain.json:72-102) at the
    BASE geometry (12L / 768H, vocabulary 250 002), B = 4, variable lengths -- the round-1..3 fixtures pin them at the tiny
    config only.  Outputs as in run_pretrain_case; `tlm` takes position_ids from the batch (model/model.py:498-499)."""
    cfg = ref_config(mm, BASE)
    model = mm.VLXLMRForPretraining(cfg, img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.train()
    for task in ("vmlm", "tlm", "mrfr", "mrc", "mrc-kl"):
        b = strip(synth.make_batch(BASE["vocab_size"], 4, 60, 36, task=task, seed=1, variable_len=True))
        key = "base4var/%s" % task
        if task == "tlm":
            out[key + "/position_ids"] = b["position_ids"].numpy()
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        scores = model(b, task, compute_loss=False)
        if task in ("vmlm", "tlm"):
            out[key + "/argmax"] = scores.argmax(-1).numpy()
        put(out, key + "/scores", scores)
        put(out, key + "/loss", loss, full=loss.numel() <= 70000 and task not in ("mrfr", "mrc-kl"))
        loss.mean().backward()
        grads_to(out, key, model)
        print("  %s: loss.mean=%.6f" % (key, loss.mean().item()))


LARGE = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
             intermediate_size=4096)


def case_large(mm, out):
    """BASELINE.json configs[4] geometry (no reference config file exists: VLXLMRConfig.from_dict with the
    XLM-R-large sizes): 24L/1024H/16 heads/4096 FFN, 80 tokens + 50 regions (L = 130), B = 2"""
    run_pretrain_case(mm, LARGE, 2, 80, 50, ["itm", "mlm"], "large2", out, False)


def install_uint8_mask_shim():
    """model/ot.py:27-29 indexes with a uint8 eye mask; torch >= 2 only accepts bool masks there.  Cast uint8 masks to
    bool inside masked_select (the torch 1.x meaning of a uint8 mask), nothing else changes."""
    orig = torch.Tensor.masked_select

    def masked_select(self, mask):
        return orig(self, mask.bool() if mask.dtype == torch.uint8 else mask)
    torch.Tensor.masked_select = masked_select


def case_more(mm, out):
    """round-2 additions: the task / branch holes of the first fixture set"""
    install_uint8_mask_shim()
    cfg = ref_config(mm, TINY)
    model = mm.VLXLMRForPretraining(cfg, img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.train()
    # ---- tlm (batch position_ids, model/model.py:498-499), tlm-ni (text only, :513-518), variable length
    for task in ("tlm", "tlm-ni"):
        b = strip(synth.make_batch(1000, 8, 32, 36, task=task, seed=1, variable_len=True))
        key = "tiny8var/%s" % task
        out[key + "/position_ids"] = b["position_ids"].numpy()
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        scores = model(b, task, compute_loss=False)
        out[key + "/argmax"] = scores.argmax(-1).numpy()
        put(out, key + "/scores", scores)
        put(out, key + "/loss", loss, full=True)
        loss.mean().backward()
        grads_to(out, key, model)
        print("  %s: loss.mean=%.6f" % (key, loss.mean().item()))
    # ---- vmlm-soft (model/model.py:627-651): KL over the VALID_XLMR_TOKEN_IDS columns of the MLM head
    b = strip(synth.make_batch(1000, 8, 32, 36, task="vmlm-soft", seed=1, n_soft=len(VALID_IDS)))
    key = "tiny8/vmlm-soft"
    model.zero_grad()
    loss = model(b, "vmlm-soft", compute_loss=True)
    scores = model(b, "vmlm-soft", compute_loss=False)
    put(out, key + "/scores", scores, full=True)
    put(out, key + "/loss", loss, full=True)
    (1000 * loss.mean()).backward()                  # pretrain.py:549-550
    grads_to(out, key, model)
    print("  %s: loss.mean=%.6f" % (key, loss.mean().item()))
    # ---- ITM with the OT regulariser (model/model.py:701-729, model/ot.py), variable length, both ot_pos_only settings
    b = strip(synth.make_batch(1000, 8, 32, 36, task="itm", seed=1, variable_len=True, ot=True))
    for pos_only in (False, True):
        key = "tiny8var/itm-ot%s" % ("-pos" if pos_only else "")
        model.ot_pos_only = pos_only
        model.zero_grad()
        itm_loss, ot_loss = model(b, "itm", compute_loss=True)
        put(out, key + "/loss", itm_loss, full=True)
        if pos_only:
            put(out, key + "/ot", ot_loss, full=True)
            ot = ot_loss.mean()
        else:
            put(out, key + "/ot_pos", ot_loss[0], full=True)
            put(out, key + "/ot_neg", ot_loss[1], full=True)
            ot = (ot_loss[0].sum() - ot_loss[1].sum()) / (ot_loss[0].size(0) + ot_loss[1].size(0))   # pretrain.py:531-533
        (itm_loss.mean() + 0.1 * ot).backward()      # itm_ot_lambda = 0.1
        grads_to(out, key, model)
        print("  %s: itm %.6f ot %.6f" % (key, itm_loss.mean().item(), ot.item()))
    model.ot_pos_only = False
    # ---- VLXLMRModel.forward text-only and image-only branches (model/model.py:439-446), variable length
    b = strip(synth.make_batch(1000, 8, 32, 36, task="mrfr", seed=3, variable_len=True))
    R = model.roberta
    T = b["input_ids"].shape[1]
    tl, nb = synth.make_batch(1000, 8, 32, 36, task="mrfr", seed=3, variable_len=True)["_txt_lens"], None
    am_t = (torch.arange(T).unsqueeze(0) < torch.tensor(tl).unsqueeze(1)).long()
    nbs = synth.make_batch(1000, 8, 32, 36, task="mrfr", seed=3, variable_len=True)["_num_bbs"]
    am_i = (torch.arange(b["img_feat"].shape[1]).unsqueeze(0) < torch.tensor(nbs).unsqueeze(1)).long()
    model.zero_grad()
    seq_t = R(b["input_ids"], None, None, None, am_t, output_all_encoded_layers=False)
    put(out, "txtonly/seq", seq_t, full=True)
    (seq_t * synth.det_normal(tuple(seq_t.shape), 55)).sum().backward()
    grads_to(out, "txtonly", model)
    model.zero_grad()
    seq_i = R(None, None, b["img_feat"], b["img_pos_feat"], am_i, img_masks=b["img_masks"], output_all_encoded_layers=False)
    put(out, "imgonly/seq", seq_i, full=True)
    (seq_i * synth.det_normal(tuple(seq_i.shape), 56)).sum().backward()
    grads_to(out, "imgonly", model)
    # all layers + pooled (the stored pooled vector was not compared before)
    layers = R(b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"], b["gather_index"],
               img_masks=b["img_masks"], output_all_encoded_layers=True)
    assert len(layers) == 2
    put(out, "alllayers/0", layers[0], full=True)
    put(out, "alllayers/1", layers[1], full=True)
    # ---- MultiheadAttention input / weight gradients (model/attention.py:267-401)
    att = importlib.import_module("model.attention")
    E, nh, L, N = 128, 4, 10, 3
    m = att.MultiheadAttention(E, nh, dropout=0.0)
    synth.det_init_(m)
    q = synth.det_normal((L, N, E), 77).requires_grad_(True)
    kpm = torch.zeros(N, L, dtype=torch.bool)
    kpm[1, 7:] = True
    kpm[2, 4:] = True
    o, w = m(q, q, q, key_padding_mask=kpm)
    (o * synth.det_normal((L, N, E), 78)).sum().backward()
    put(out, "mha/dq", q.grad, full=True)
    for n, p in m.named_parameters():
        put(out, "mha/grad/" + n, p.grad, full=p.numel() <= 70000)
    # general variants: cross-attention with an additive attn_mask and separate key/value inputs
    m2 = att.MultiheadAttention(E, nh, dropout=0.0)
    synth.det_init_(m2)
    S = 7
    q2 = synth.det_normal((L, N, E), 80).requires_grad_(True)
    k2 = synth.det_normal((S, N, E), 81).requires_grad_(True)
    v2 = synth.det_normal((S, N, E), 82).requires_grad_(True)
    amask = synth.det_normal((L, S), 83)
    kpm2 = torch.zeros(N, S, dtype=torch.bool)
    kpm2[2, 5:] = True
    o2, w2 = m2(q2, k2, v2, key_padding_mask=kpm2, attn_mask=amask)
    put(out, "mha_cross/out", o2, full=True)
    put(out, "mha_cross/weights", w2, full=True)
    (o2 * synth.det_normal((L, N, E), 84)).sum().backward()
    put(out, "mha_cross/dq", q2.grad, full=True)
    put(out, "mha_cross/dk", k2.grad, full=True)
    put(out, "mha_cross/dv", v2.grad, full=True)
    for n, p in m2.named_parameters():
        put(out, "mha_cross/grad/" + n, p.grad, full=p.numel() <= 70000)


def case_retrieval(mi, out):
    """retrieval inference (itm.py:516-538 + eval/itm.py:6-53): a 12-text x 11-image score matrix from the reference's
    VLXLMRForImageTextRetrieval in eval mode, mini-batches of 4 images, and the reference's own itm_eval on it"""
    ev = load_file("ref_eval_itm", REF + "/eval/itm.py")
    cfg = ref_config(importlib.import_module("model.model"), TINY)
    model = mi.VLXLMRForImageTextRetrieval(cfg, img_dim=2048, margin=0.2)
    synth.det_init_(model)
    model.eval()
    n_txt, n_img = 12, 11
    pool = synth.retrieval_pool(1000, n_txt, n_img, 32, 36)
    sm = torch.zeros(n_txt, n_img)
    with torch.no_grad():
        for i in range(n_txt):
            for j0 in range(0, n_img, 4):
                j1 = min(n_img, j0 + 4)
                b = synth.retrieval_batch(pool, i, j0, j1)
                sm[i, j0:j1] = model(b, compute_loss=False).squeeze(1)
    out["retrieval/scores"] = sm.numpy()
    txt_ids = ["t%d" % i for i in range(n_txt)]
    img_ids = ["i%d" % j for j in range(n_img)]
    txt2img = {t: img_ids[(3 * k) % n_img] for k, t in enumerate(txt_ids)}
    img2txts = {j: [t for t in txt_ids if txt2img[t] == j] for j in img_ids}
    # the recall numbers are pinned on a tie-free fp16 matrix: torch.topk leaves the order of equal scores
    # unspecified, and the tiny model's raw scores collide in fp16
    ev_in = ((sm - sm.mean()) / sm.std() + synth.det_uniform((n_txt, n_img), 99, -0.05, 0.05)).half()
    for r in range(n_txt):
        assert len(set(ev_in[r].tolist())) == n_img
    for c in range(n_img):
        assert len(set(ev_in[:, c].tolist())) == n_txt
    out["retrieval/eval_input"] = ev_in.float().numpy()
    log = ev.itm_eval(ev_in.float(), txt_ids, img_ids, txt2img, img2txts)
    for k, v in log.items():
        out["retrieval/eval/" + k] = np.array([v])
    print("  retrieval:", {k: round(v, 4) for k, v in log.items()})


def case_collate(out):
    """the reference's own collate functions (data/itm.py:205-232, data/mrm.py:73-119,253-288) on ragged synthetic
    samples: pins the device-side batch assembly of uc2_amd/data/loader.py"""
    sys.modules["cytoolz"].partition_all = lambda n, seq: [seq[i:i + n] for i in range(0, len(seq), n)]
    sys.modules["cytoolz"].curry = lambda f: f
    sys.modules["toolz"].curry = lambda f: f
    d = types.ModuleType("data")
    d.__path__ = [REF + "/data"]
    sys.modules["data"] = d
    di = importlib.import_module("data.itm")
    dm = importlib.import_module("data.mrm")
    for task, fn in (("itm", di.xlmr_itm_collate), ("mrfr", dm.xlmr_mrfr_collate), ("mrc", dm.xlmr_mrc_collate)):
        b = fn(synth.sample_tuples(task, 6))
        for k, v in b.items():
            out["collate/%s/%s" % (task, k)] = v.numpy()
        print("  collate", task, {k: tuple(v.shape) for k, v in b.items()})


def main():
    which = sys.argv[1:] or ["tiny", "gather", "adamw", "mha", "base", "base_tasks", "more", "large", "retrieval", "collate"]
    install_shims()
    sys.path.insert(0, REF)
    mm = importlib.import_module("model.model")
    mi = importlib.import_module("model.itm")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for w in which:
        out = OrderedDict()
        print("case", w)
        if w == "tiny":
            case_tiny(mm, mi, out)
        elif w == "gather":
            case_gather(out)
        elif w == "adamw":
            case_adamw(mm, out)
        elif w == "mha":
            case_mha(out)
        elif w == "base":
            case_base(mm, out)
        elif w == "base_tasks":
            case_base_tasks(mm, out)
        elif w == "large":
            case_large(mm, out)
        elif w == "more":
            case_more(mm, out)
        elif w == "retrieval":
            case_retrieval(mi, out)
        elif w == "collate":
            case_collate(out)
        else:
            raise SystemExit("unknown case " + w)
        path = os.path.join(HERE, "golden_%s.npz" % w)
        np.savez_compressed(path, **out)
        print("  wrote %s (%d arrays, %.1f KB)" % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
