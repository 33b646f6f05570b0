"""Model-level parity on the GPU: the HIP path (through the reference-shaped modules) against
  (1) the committed golden vectors produced by the reference itself, and
  (2) the CPU oracle run on the box on the same seeded inputs.
fp32 mode is held to the north-star tolerance (1e-3 relative, argmax bit-exact); bf16 mode is
checked against the same vectors at bf16 resolution (SURVEY.md §7 'Tolerance vs bf16')."""
from collections import OrderedDict

import numpy as np
import ctypes

import pytest
import torch

import uc2_amd
from oracle import specs
from oracle import uc2_oracle as O
from uc2_amd import ops
from uc2_amd.config import cfg as knobs, state
from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
from uc2_amd.model.layer import BertLayer
from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
from uc2_amd.optim.adamw import AdamW, clip_grad_norm_
from uc2_amd.optim.misc import param_groups
from uc2_amd.store import set_compute_dtype
from uc2_amd.utils import synth
from util import check_against_golden, golden, max_rel, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL32 = 1e-3          # north_star: within 1e-3 relative fp32


def make_cfg(geom, drop=0.0):
    d = dict(hidden_act="gelu", hidden_dropout_prob=drop, attention_probs_dropout_prob=drop,
             max_position_embeddings=514, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5,
             pad_token_id=1)
    d.update(geom)
    return VLXLMRConfig.from_dict(d)


def strip(b):
    return {k: v for k, v in b.items() if not k.startswith("_")}


def to_dev(b):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in strip(b).items()}


def build_pretrain(geom, dtype):
    model = VLXLMRForPretraining(make_cfg(geom), img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.to(DEV).train()
    set_compute_dtype(model, dtype)
    return model


# ------------------------------------------------------------------------------------------ one layer
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom,B,L", [(O.TINY, 5, 68), (O.BASE, 2, 96)])
def test_bert_layer_vs_oracle(dtype, geom, B, L):
    cfg = make_cfg(geom)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    W = OrderedDict((n, p.detach().clone()) for n, p in layer.named_parameters())
    layer.to(DEV).train()
    set_compute_dtype(layer, dtype)
    H = cfg.hidden_size
    x = synth.det_normal((B, L, H), 11)
    am = torch.ones(B, L, dtype=torch.long)
    am[0, L - 7:] = 0
    ext = O.extended_mask(am)
    dy = synth.det_normal((B, L, H), 12)
    # oracle (CPU, fp32 autograd over plain ops)
    xo = x.clone().requires_grad_(True)
    Wg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in W.items())
    yo = O.bert_layer(xo, ext, Wg, "", cfg.num_attention_heads)
    yo.backward(dy)
    # HIP path
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    yd = layer(xd, ext.to(DEV))
    yd.backward(dy.to(DEV).to(dtype))
    t_out, t_grad = (TOL32, 2e-3) if dtype == torch.float32 else (3e-2, 4e-2)
    assert max_rel(yd.float().cpu(), yo.detach()) < t_out
    assert rel_err(xd.grad.float().cpu(), xo.grad) < t_grad
    qb = Wg["attention.self.query.bias"].grad.norm().item()
    for n, p in layer.named_parameters():
        ref = Wg[n].grad
        if n.endswith("key.bias"):          # mathematically zero (softmax is shift invariant): noise only
            assert p.grad.norm().item() < 1e-2 * qb and ref.norm().item() < 1e-3 * qb
            continue
        e = rel_err(p.grad.cpu(), ref)
        assert e < t_grad, "%s grad rel err %.3e" % (n, e)


def test_bert_layer_dropout_runs_and_is_reproducible():
    cfg = make_cfg(O.TINY, drop=0.1)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    layer.to(DEV).train()
    x = synth.det_normal((4, 68, 128), 3).to(DEV)
    ext = torch.zeros(4, 1, 1, 68, device=DEV)
    ops.rng.manual_seed(77)
    y1 = layer(x, ext)
    ops.rng.manual_seed(77)
    y2 = layer(x, ext)
    y3 = layer(x, ext)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    layer.eval()
    assert torch.equal(layer(x, ext), layer(x, ext))


def test_bert_layer_backward_paths_at_bench_token_counts():
    """One base-geometry BertLayer at 176 x 96 = 16 896 tokens (>= knobs.wgrad_side_min_rows): the backward the bench step runs --
    input gradients on the k-contiguous weight copies W^T (ParamStore.compute_t), weight gradients on the side stream -- against the
    same layer with W read k-strided and everything on one stream, and against the grouped weight-gradient launch forced on at
    this size.  Same dropout masks (same seed), bf16: dx and every parameter gradient agree to bf16 rounding of the GEMM outputs."""
    cfg = make_cfg(O.BASE, drop=0.1)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    layer.to(DEV).train()
    set_compute_dtype(layer, torch.bfloat16)
    B, L, H = 176, 96, 768
    x0 = (synth.det_normal((B, L, H), 3) * 0.5).to(DEV).to(torch.bfloat16)
    ext = torch.zeros(B, 1, 1, L, device=DEV)
    dy = (synth.det_normal((B, L, H), 4) * 0.1).to(DEV).to(torch.bfloat16)

    def run(wt, side, group_rows):
        saved = (knobs.dgrad_transposed_w, knobs.wgrad_side_stream, knobs.wgrad_side_min_rows)
        knobs.dgrad_transposed_w, knobs.wgrad_side_stream = wt, side
        if group_rows is not None:
            knobs.wgrad_side_min_rows = group_rows          # (the grouped launch and W^T are both gated by this threshold)
        try:
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            ops.rng.manual_seed(5)
            y = layer(x, ext)
            y.backward(dy)
            ops.join_side_streams()
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), OrderedDict((n, p.grad.detach().clone()) for n, p in layer.named_parameters())
        finally:
            knobs.dgrad_transposed_w, knobs.wgrad_side_stream, knobs.wgrad_side_min_rows = saved
    y_ref, dx_ref, g_ref = run(False, False, None)                  # W k-strided, one stream, one GEMM per weight gradient
    for (wt, side, rows) in ((True, True, None), (False, False, 1 << 30)):          # bench path; grouped weight gradients
        y, dx, g = run(wt, side, rows)
        assert torch.equal(y, y_ref)
        assert rel_err(dx.float(), dx_ref.float()) < 4e-3
        for n in g_ref:
            if n.endswith("key.bias"):
                continue                                            # (mathematically zero: rounding noise on both sides)
            assert rel_err(g[n], g_ref[n]) < 4e-3, n


def test_bert_layer_fused_dropout_residual_tails_vs_unfused():
    """knobs.ln_fuse (from knobs.ln_fuse_min_rows tokens): the Wo and FFN2 GEMMs write dropout(dense) + residual, the LayerNorm kernels
    read that one tensor.  Same dropout masks as the unfused tails (same seed and sites); the sum is rounded to bf16 once where the
    unfused form rounds the dense output: layer output, dx and every parameter gradient agree to bf16 rounding
    (base geometry, 176 x 96 tokens, dropout on; forward-only route too)"""
    cfg = make_cfg(O.BASE, drop=0.1)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    layer.to(DEV).train()
    set_compute_dtype(layer, torch.bfloat16)
    B, L, H = 176, 96, 768
    assert B * L >= knobs.ln_fuse_min_rows
    x0 = (synth.det_normal((B, L, H), 3) * 0.5).to(DEV).to(torch.bfloat16)
    ext = torch.zeros(B, 1, 1, L, device=DEV)
    ext[::5, :, :, L - 11:] = -10000.0
    dy = (synth.det_normal((B, L, H), 4) * 0.1).to(DEV).to(torch.bfloat16)

    def run(on):
        was, knobs.ln_fuse = knobs.ln_fuse, (3 if on else 0)
        try:
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            ops.rng.manual_seed(5)
            y = layer(x, ext)
            y.backward(dy)
            ops.join_side_streams()
            torch.cuda.synchronize()
            with torch.no_grad():
                ops.rng.manual_seed(5)
                yf = layer(x0, ext)
            return y.detach().clone(), x.grad.clone(), OrderedDict((n, p.grad.detach().clone()) for n, p in layer.named_parameters()), yf
        finally:
            knobs.ln_fuse = was
    y0, dx0, g0, yf0 = run(False)
    y1, dx1, g1, yf1 = run(True)
    assert not torch.equal(y1, y0)                                   # (the fused route really ran: another rounding point)
    assert torch.equal(yf1, y1) and torch.equal(yf0, y0)             # forward-only route == training forward, in both forms
    assert rel_err(y1.float(), y0.float()) < 6e-3
    assert rel_err(dx1.float(), dx0.float()) < 1e-2
    for n in g0:
        if n.endswith("key.bias"):
            continue
        assert rel_err(g1[n], g0[n]) < 1e-2, n


@pytest.mark.parametrize("dtype,geom,B,L", [(torch.bfloat16, O.BASE, 104, 96), (torch.bfloat16, O.TINY, 5, 68), (torch.float32, O.TINY, 5, 68)])
def test_bert_layer_native_entry_is_bit_identical(dtype, geom, B, L):
    """uc2_bert_layer_fwd / uc2_bert_layer_bwd (knobs.native_layer: one C call per layer and direction at the reference's micro-batch
    sizes, config/uc2_pretrain.json:17-19) against the per-kernel calls of BertLayerFn: the same kernels with the same arguments in
    the same order -- layer output, dx and every weight gradient bit-identical (dropout on, key mask, forward-only route unchanged)"""
    cfg = make_cfg(geom, drop=0.1)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    layer.to(DEV).train()
    set_compute_dtype(layer, dtype)
    H = cfg.hidden_size
    x0 = (synth.det_normal((B, L, H), 3) * 0.5).to(DEV).to(dtype)
    ext = torch.zeros(B, 1, 1, L, device=DEV)
    ext[::3, :, :, L - 9:] = -10000.0
    dy = (synth.det_normal((B, L, H), 4) * 0.1).to(DEV).to(dtype)
    from uc2_amd import _lib
    calls = []
    lib = _lib.load()

    def run(on):
        was, knobs.native_layer = knobs.native_layer, on
        try:
            outs = []
            for need_dx in (True, False):
                layer.zero_grad()
                x = x0.clone().requires_grad_(need_dx)
                ops.rng.manual_seed(5)
                y = layer(x, ext)
                y.backward(dy)
                ops.join_side_streams()
                torch.cuda.synchronize()
                outs.append((y.detach().clone(), x.grad.clone() if need_dx else None,
                             OrderedDict((n, p.grad.detach().clone()) for n, p in layer.named_parameters())))
            return outs
        finally:
            knobs.native_layer = was
    assert ops._native_layer_ok(dtype, B * L, False, None) == knobs.native_layer
    ref = run(False)
    nat = run(True)
    for (y0, dx0, g0), (y1, dx1, g1) in zip(ref, nat):
        assert torch.equal(y1, y0)
        assert (dx0 is None and dx1 is None) or torch.equal(dx1, dx0)
        for n in g0:
            if n.endswith("bias") or "LayerNorm" in n:
                assert rel_err(g1[n], g0[n]) < 1e-5 or g0[n].norm() < 1e-6, n      # (column sums through fp32 atomics: order varies run to run)
            else:
                assert torch.equal(g1[n], g0[n]), n
    # argument errors are reported, not executed
    bad = ops._BertLayerC()
    assert lib.uc2_bert_layer_fwd(ctypes.byref(bad), None) != 0 and b"layer.hip" in lib.uc2_last_error()


def test_bert_layer_interleaved_qkv_route_is_bit_identical():
    """knobs.qkv_interleaved (from 16 384 tokens): the QKV GEMM on the row-permuted weight copy, attention on head-interleaved q|k|v,
    dWqkv un-permuted by its split-K reduction, dX through W'^T -- same dot products in the same order as the plain layout: the
    layer output and every parameter gradient bit-identical, dx up to the summation order of its contraction over the interleaved
    index (base geometry, 176 x 96 tokens, dropout on, side stream on)"""
    cfg = make_cfg(O.BASE, drop=0.1)
    layer = BertLayer(cfg)
    synth.det_init_(layer)
    layer.to(DEV).train()
    set_compute_dtype(layer, torch.bfloat16)
    B, L, H = 176, 96, 768
    x0 = (synth.det_normal((B, L, H), 3) * 0.5).to(DEV).to(torch.bfloat16)
    ext = torch.zeros(B, 1, 1, L, device=DEV)
    ext[::5, :, :, L - 11:] = -10000.0
    dy = (synth.det_normal((B, L, H), 4) * 0.1).to(DEV).to(torch.bfloat16)

    def run(on):
        was, knobs.qkv_interleaved = knobs.qkv_interleaved, on
        try:
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            ops.rng.manual_seed(5)
            y = layer(x, ext)
            y.backward(dy)
            ops.join_side_streams()
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), OrderedDict((n, p.grad.detach().clone()) for n, p in layer.named_parameters())
        finally:
            knobs.qkv_interleaved = was
    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    assert torch.equal(y1, y0)
    # (dx = dqkv W contracts over the interleaved index: same products, another summation order -- bf16 rounding of the result)
    assert rel_err(dx1.float(), dx0.float()) < 3e-3
    for n in g0:
        if n.endswith("bias") or "LayerNorm" in n:
            assert rel_err(g1[n], g0[n]) < 1e-5 or g0[n].norm() < 1e-6, n      # (column sums added up with fp32 atomics: order varies run to run)
        else:
            assert torch.equal(g1[n], g0[n]), n                                # weight gradients: bit-identical
    with torch.no_grad():                                                       # forward-only route (scoring / validation)
        layer.eval()
        a = layer(x0, ext)
        was, knobs.qkv_interleaved = knobs.qkv_interleaved, False
        try:
            b = layer(x0, ext)
        finally:
            knobs.qkv_interleaved = was
        assert torch.equal(a, b)


def test_embedding_dropout_sits_after_layernorm():
    """training mode, p > 0: the reference computes dropout(LayerNorm(emb)) (model/model.py:331-333,361-363), so every
    embedding output element is 0 or eval_output / (1 - p)"""
    p = 0.25
    model = VLXLMRForPretraining(make_cfg(O.TINY, drop=p), img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.to(DEV)
    b = to_dev(synth.make_batch(1000, 6, 32, 36, task="mrfr", seed=4))
    R = model.roberta
    model.eval()
    with torch.no_grad():
        t0 = R._compute_txt_embeddings(b["input_ids"], None)
        i0 = R._compute_img_embeddings(b["img_feat"], b["img_pos_feat"], b["img_masks"])
    model.train()
    with torch.no_grad():
        t1 = R._compute_txt_embeddings(b["input_ids"], None)
        i1 = R._compute_img_embeddings(b["img_feat"], b["img_pos_feat"], b["img_masks"])
    for e0, e1 in ((t0, t1), (i0, i1)):
        keep = e1 != 0
        assert abs((1 - keep.float().mean().item()) - p) < 0.02
        assert torch.allclose(e1[keep], e0[keep] / (1 - p), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ whole model vs golden
def run_task(model, batch, task):
    b = to_dev(batch)
    model.zero_grad()
    seq = model.roberta(b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"], b["gather_index"],
                        img_masks=b.get("img_masks"), output_all_encoded_layers=False)
    scores = model(b, task, compute_loss=False)
    scores = scores[0] if isinstance(scores, tuple) else scores
    model.zero_grad()
    loss = model(b, task, compute_loss=True)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.mean().backward()
    return seq, scores, loss


@pytest.mark.parametrize("tag,B,var,tasks", [("tiny8", 8, False, ["itm", "mlm", "mrfr", "mrc", "mrc-kl", "vmlm"]),
                                             ("tiny8var", 8, True, ["itm", "mlm"]),
                                             ("tiny64", 64, False, ["itm", "mlm"])])
def test_pretrain_fp32_vs_golden(tag, B, var, tasks):
    g = golden("tiny")
    model = build_pretrain(O.TINY, torch.float32)
    for task in tasks:
        batch = synth.make_batch(1000, B, 32, 36, task=task, seed=1, variable_len=var)
        seq, scores, loss = run_task(model, batch, task)
        key = "%s/%s" % (tag, task)
        check_against_golden(g, key + "/seq", seq, TOL32)
        check_against_golden(g, key + "/scores", scores, TOL32)
        check_against_golden(g, key + "/loss", loss, TOL32)
        if key + "/argmax" in g.files:
            assert np.array_equal(scores.argmax(-1).cpu().numpy(), g[key + "/argmax"])    # bit-exact labels
        n = 0
        for name, p in model.named_parameters():
            k = "%s/grad/%s" % (key, name)
            if k + "/sum3" not in g.files:
                continue
            if float(g[k + "/sum3"][2]) < 1e-7:
                assert p.grad is None or p.grad.norm().item() < 1e-5
            else:
                assert p.grad is not None, name
                check_against_golden(g, k, p.grad, 3e-3, what=task)
            n += 1
        assert n > 40


def test_itm_rank_fp32_vs_golden():
    g = golden("tiny")
    model = VLXLMRForImageTextRetrieval(make_cfg(O.TINY), img_dim=2048, margin=0.2)
    synth.det_init_(model)
    model.to(DEV).train()
    b = to_dev(synth.make_batch(1000, 12, 32, 36, task="rank", seed=2, variable_len=True, sample_size=3))
    check_against_golden(g, "rank/scores", model(b, compute_loss=False), TOL32)
    loss = model(b, compute_loss=True)
    check_against_golden(g, "rank/loss", loss, TOL32)
    loss.mean().backward()
    for name, p in model.named_parameters():
        k = "rank/grad/%s" % name
        if k + "/sum3" in g.files and float(g[k + "/sum3"][2]) > 1e-7:
            check_against_golden(g, k, p.grad, 1e-2)


@pytest.mark.parametrize("task", ["itm", "mlm"])
def test_pretrain_bf16_vs_golden(task):
    """throughput mode (bf16 MFMA GEMMs, fp32 statistics) against the fp32 reference at bf16 resolution"""
    g = golden("tiny")
    model = build_pretrain(O.TINY, torch.bfloat16)
    batch = synth.make_batch(1000, 8, 32, 36, task=task, seed=1)
    seq, scores, loss = run_task(model, batch, task)
    key = "tiny8/%s" % task
    check_against_golden(g, key + "/seq", seq, 6e-2)
    check_against_golden(g, key + "/loss", loss, 3e-2)
    ref_lm = float(g[key + "/loss/sum3"][0]) / loss.numel()
    assert abs(loss.mean().item() - ref_lm) < 5e-3 * abs(ref_lm)
    for name in ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.1.output.dense.weight",
                 "roberta.img_embeddings.img_linear.weight"):
        p = dict(model.named_parameters())[name]
        # the ITM gradient at init is tiny and cancels heavily: bf16 noise is relatively larger there
        check_against_golden(g, "%s/grad/%s" % (key, name), p.grad, 6e-2 if task == "mlm" else 0.25, metric="l2")


@pytest.mark.parametrize("task", ["itm", "mlm"])
def test_pretrain_base_geometry_fp32_vs_golden(task):
    """BASELINE.json configs[1] geometry (12L/768H, vocab 250002, 60 tokens + 36 regions), B=4"""
    g = golden("base")
    model = build_pretrain(O.BASE, torch.float32)
    batch = synth.make_batch(250002, 4, 60, 36, task=task, seed=1)
    seq, scores, loss = run_task(model, batch, task)
    key = "base4/%s" % task
    check_against_golden(g, key + "/seq", seq, TOL32)
    check_against_golden(g, key + "/loss", loss, TOL32)
    assert np.array_equal(scores.argmax(-1).cpu().numpy(), g[key + "/argmax"])
    P = dict(model.named_parameters())
    for name in ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.11.output.dense.weight",
                 "roberta.img_embeddings.img_linear.weight", "roberta.embeddings.LayerNorm.weight"):
        check_against_golden(g, "%s/grad/%s" % (key, name), P[name].grad, 3e-3)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pretrain_base_geometry_mix_tasks_vs_golden(dtype):
    """round 4 fixtures: vmlm / tlm / mrfr / mrc / mrc-kl -- the MRM, VMLM and TLM tasks of BASELINE.json configs[2]'s pretrain mix
    (config/uc2_pretrain.json:72-102) -- at the BASE geometry (12L / 768H, vocabulary 250 002), B = 4, variable lengths, against
    the reference's own outputs.  fp32: the north-star tolerance (scores and losses 1e-3, argmax bit-exact, gradients 3e-3);
    bf16 (the measured mode): mean loss, argmax agreement and gradient L2 errors at bf16 resolution, printed.
    Reference: model/model.py:600-688,738-775."""
    g = golden("base_tasks")
    model = build_pretrain(O.BASE, dtype)
    f32 = dtype == torch.float32
    names = ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.11.output.dense.weight",
             "roberta.img_embeddings.img_linear.weight", "roberta.embeddings.LayerNorm.weight")
    for task in ("vmlm", "tlm", "mrfr", "mrc", "mrc-kl"):
        batch = synth.make_batch(250002, 4, 60, 36, task=task, seed=1, variable_len=True)
        b = to_dev(batch)
        key = "base4var/%s" % task
        model.zero_grad()
        scores = model(b, task, compute_loss=False)
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        loss.mean().backward()
        ops.join_side_streams()
        torch.cuda.synchronize()
        P = dict(model.named_parameters())
        ref_mean = float(g[key + "/loss/sum3"][0]) / max(loss.numel(), 1)
        mean_rel = abs(loss.mean().item() - ref_mean) / abs(ref_mean)
        if f32:
            check_against_golden(g, key + "/scores", scores, TOL32)
            check_against_golden(g, key + "/loss", loss, TOL32)
            if key + "/argmax" in g.files:
                assert np.array_equal(scores.argmax(-1).cpu().numpy(), g[key + "/argmax"])     # bit-exact predictions
            for name in names:
                check_against_golden(g, "%s/grad/%s" % (key, name), P[name].grad, 3e-3, what=task)
        else:
            rep = {"mean-loss rel": mean_rel}
            assert mean_rel < (2e-2 if task == "mrc-kl" else 5e-3), (task, mean_rel)      # (mrc-kl: a loss of 9e-4, differences of small numbers)
            if key + "/argmax" in g.files:
                agree = float((scores.argmax(-1).cpu().numpy() == g[key + "/argmax"]).mean())
                rep["argmax agreement"] = agree
                assert agree >= 0.8           # ~35 masked positions at B = 4: one flip is 3 %; near-ties of a random-init head (measured 0.88-1.0)
            for name in names:
                rep["grad L2 " + name.split("roberta.")[1]] = check_against_golden(
                    g, "%s/grad/%s" % (key, name), P[name].grad, 0.04, what=task, metric="l2")          # measured 0.8-1.7 %
            for k, v in rep.items():
                print("bf16 base4var %-7s %-55s %.4g" % (task, k, v))
    del model
    torch.cuda.empty_cache()


def _grad_checks(g, key, model, tol, min_n=30, what=""):
    n = 0
    for name, p in model.named_parameters():
        k = "%s/grad/%s" % (key, name)
        if k + "/sum3" not in g.files:
            continue
        if name.endswith("key.bias"):                  # mathematically zero: rounding noise only
            n += 1
            continue
        if float(g[k + "/sum3"][2]) < 1e-7:
            assert p.grad is None or p.grad.norm().item() < 1e-5, name
        else:
            assert p.grad is not None, name
            check_against_golden(g, k, p.grad, tol, what=what)
        n += 1
    assert n > min_n, n


def test_more_tasks_fp32_vs_golden():
    """round-2 fixtures from the reference: tlm (batch position_ids, model/model.py:498-499), tlm-ni (text-only
    branch, :513-518), vmlm-soft (:627-651); variable length; losses 1e-3, argmax bit-exact, gradients 3e-3"""
    g = golden("more")
    model = build_pretrain(O.TINY, torch.float32)
    for task in ("tlm", "tlm-ni"):
        b = to_dev(synth.make_batch(1000, 8, 32, 36, task=task, seed=1, variable_len=True))
        key = "tiny8var/%s" % task
        model.zero_grad()
        scores = model(b, task, compute_loss=False)
        check_against_golden(g, key + "/scores", scores, TOL32)
        assert np.array_equal(scores.argmax(-1).cpu().numpy(), g[key + "/argmax"])
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        check_against_golden(g, key + "/loss", loss, TOL32)
        loss.mean().backward()
        _grad_checks(g, key, model, 3e-3, what=task)
    b = to_dev(synth.make_batch(1000, 8, 32, 36, task="vmlm-soft", seed=1, n_soft=45))
    key = "tiny8/vmlm-soft"
    model.zero_grad()
    check_against_golden(g, key + "/scores", model(b, "vmlm-soft", compute_loss=False), TOL32)
    model.zero_grad()
    loss = model(b, "vmlm-soft", compute_loss=True)
    check_against_golden(g, key + "/loss", loss, TOL32)
    (1000 * loss.mean()).backward()                     # pretrain.py:549-550
    _grad_checks(g, key, model, 3e-3, what="vmlm-soft")


def test_itm_with_ot_regulariser_fp32_vs_golden():
    """f4: ITM + optimal-transport regulariser (model/model.py:701-729, model/ot.py:32-82), variable length, both
    ot_pos_only settings: ITM losses, OT distances and every gradient of itm.mean() + 0.1 * ot against the reference"""
    g = golden("more")
    model = build_pretrain(O.TINY, torch.float32)
    b = to_dev(synth.make_batch(1000, 8, 32, 36, task="itm", seed=1, variable_len=True, ot=True))
    b["ot_inputs"] = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b["ot_inputs"].items()}
    for pos_only in (False, True):
        key = "tiny8var/itm-ot%s" % ("-pos" if pos_only else "")
        model.ot_pos_only = pos_only
        model.zero_grad()
        itm_loss, ot = model(b, "itm", compute_loss=True)
        check_against_golden(g, key + "/loss", itm_loss, TOL32)
        if pos_only:
            check_against_golden(g, key + "/ot", ot, TOL32)
            otl = ot.mean()
        else:
            check_against_golden(g, key + "/ot_pos", ot[0], TOL32)
            check_against_golden(g, key + "/ot_neg", ot[1], TOL32)
            otl = (ot[0].sum() - ot[1].sum()) / (ot[0].size(0) + ot[1].size(0))
        (itm_loss.mean() + 0.1 * otl).backward()
        _grad_checks(g, key, model, 3e-3, what=key)
    model.ot_pos_only = False


def test_text_only_image_only_all_layers_fp32_vs_golden():
    """VLXLMRModel.forward's text-only / image-only branches (model/model.py:439-446) with gradients (padding rows
    of the embedding tables get none: nn.Embedding padding_idx), and output_all_encoded_layers=True"""
    g = golden("more")
    model = build_pretrain(O.TINY, torch.float32)
    full = synth.make_batch(1000, 8, 32, 36, task="mrfr", seed=3, variable_len=True)
    b = to_dev(full)
    R = model.roberta
    T, NR = b["input_ids"].shape[1], b["img_feat"].shape[1]
    am_t = (torch.arange(T).unsqueeze(0) < torch.tensor(full["_txt_lens"]).unsqueeze(1)).long().to(DEV)
    am_i = (torch.arange(NR).unsqueeze(0) < torch.tensor(full["_num_bbs"]).unsqueeze(1)).long().to(DEV)
    model.zero_grad()
    seq_t = R(b["input_ids"], None, None, None, am_t, output_all_encoded_layers=False)
    check_against_golden(g, "txtonly/seq", seq_t, TOL32)
    (seq_t * synth.det_normal(tuple(seq_t.shape), 55).to(DEV)).sum().backward()
    _grad_checks(g, "txtonly", model, 3e-3, min_n=25)
    model.zero_grad()
    seq_i = R(None, None, b["img_feat"], b["img_pos_feat"], am_i, img_masks=b["img_masks"], output_all_encoded_layers=False)
    check_against_golden(g, "imgonly/seq", seq_i, TOL32)
    (seq_i * synth.det_normal(tuple(seq_i.shape), 56).to(DEV)).sum().backward()
    _grad_checks(g, "imgonly", model, 3e-3, min_n=25)
    layers = R(b["input_ids"], None, b["img_feat"], b["img_pos_feat"], b["attn_masks"], b["gather_index"],
               img_masks=b["img_masks"], output_all_encoded_layers=True)
    assert len(layers) == 2
    check_against_golden(g, "alllayers/0", layers[0], TOL32)
    check_against_golden(g, "alllayers/1", layers[1], TOL32)
    # pooled output of the ITM golden (stored by round 1, compared here for the first time)
    g0 = golden("tiny")
    bi = to_dev(synth.make_batch(1000, 8, 32, 36, task="itm", seed=1))
    seq = R(bi["input_ids"], None, bi["img_feat"], bi["img_pos_feat"], bi["attn_masks"], bi["gather_index"],
            output_all_encoded_layers=False)
    check_against_golden(g0, "tiny8/itm/pooled", R.pooler(seq), TOL32)


def test_bf16_base_12_layers_vs_golden():
    """THE MEASURED MODE against the reference: bf16 MFMA GEMMs / MFMA attention / polynomial GELU, 12 layers,
    768 hidden, vocabulary 250 002 (BASELINE.json configs[1] geometry, B = 4), against the reference's fp32 vectors
    (tests/golden/golden_base.npz).  bf16 carries 8 significant bits, so element-wise 1e-3 is out of reach by
    construction; the tolerances below are the tightest that hold, the measured errors are printed.
    ITM labels must agree exactly; MLM argmax agreement over the 250 002-way head is reported and held >= 90 %."""
    g = golden("base")
    model = build_pretrain(O.BASE, torch.bfloat16)
    report = {}
    for task in ("itm", "mlm"):
        batch = synth.make_batch(250002, 4, 60, 36, task=task, seed=1)
        seq, scores, loss = run_task(model, batch, task)
        key = "base4/%s" % task
        report[task + " seq slice rel (max)"] = check_against_golden(g, key + "/seq", seq, 4e-2)        # measured 1.8e-2 / 2.1e-2
        report[task + " loss slice rel (max)"] = check_against_golden(g, key + "/loss", loss, 1e-2)     # measured 3.7e-3 / 1.0e-3
        ref_mean = float(g[key + "/loss/sum3"][0]) / loss.numel()
        report[task + " mean-loss rel"] = abs(loss.mean().item() - ref_mean) / abs(ref_mean)
        assert report[task + " mean-loss rel"] < 5e-3
        am = scores.argmax(-1).cpu().numpy()
        agree = float((am == g[key + "/argmax"]).mean())
        report[task + " argmax agreement"] = agree
        if task == "itm":
            assert agree == 1.0                           # ITM labels bit-exact
        else:
            assert agree >= 0.9
        P = dict(model.named_parameters())
        for name in ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.11.output.dense.weight",
                     "roberta.img_embeddings.img_linear.weight", "roberta.embeddings.LayerNorm.weight"):
            report["%s grad L2 rel %s" % (task, name.split("roberta.")[1])] = check_against_golden(
                g, "%s/grad/%s" % (key, name), P[name].grad, 0.03 if task == "mlm" else 0.07, metric="l2")   # measured 1.3e-2 / 3.4e-2
    for k, v in report.items():
        print("bf16-12L %-70s %.4g" % (k, v))
    del model
    torch.cuda.empty_cache()


def test_bf16_measured_kernels_vs_oracle_at_128_pairs():
    """THE MEASURED MODE through THE MEASURED KERNELS: at 128 pairs (M = 12 288 tokens) the committed plans put every encoder GEMM on
    the persistent ping-pong family (variants 8 / 9 / 12: 256- or 192-row tiles, 32x32x16 or 16x16x32 MFMA, fused GELU / gelu' / residual epilogues) -- the
    kernels bench.py times -- where the golden-vector tests (B = 4) run on the library's default kernels.  12 layers, vocabulary
    250 002, against the CPU oracle on the same weights and batch, run here on the box: ITM label agreement over 128 labels (flips are
    listed with the oracle's logit margin and must be near-ties), MLM argmax agreement over ~1 100 masked tokens, mean losses."""
    for (ta, tb, n, k) in [(False, False, 2304, 768), (False, False, 768, 768), (False, False, 3072, 768), (False, False, 768, 3072),
                           (False, True, 3072, 768), (False, True, 768, 3072), (False, True, 768, 2304), (False, True, 768, 768)]:
        assert ops.gemm_plan(torch.bfloat16, ta, tb, 128 * 96, n, k)[0] in (8, 9, 12), (ta, tb, n, k)
    model = build_pretrain(O.BASE, torch.bfloat16)
    W = _oracle_weights(model)
    cfg = _oracle_cfg(O.BASE)
    B = 128
    for task in ("itm", "mlm"):
        batch = synth.make_batch(250002, B, 60, 36, task=task, seed=11)
        _, scores, loss = run_task(model, batch, task)
        with torch.no_grad():
            ref_scores = O.pretrain_forward(W, cfg, strip(batch), task, compute_loss=False)
            ref_scores = ref_scores[0] if isinstance(ref_scores, tuple) else ref_scores
            tgt = batch["targets"] if task == "itm" else batch["txt_labels"][batch["txt_labels"] != -1]
            ref_loss = torch.nn.functional.cross_entropy(ref_scores, tgt, reduction="none")       # what pretrain_forward returns with compute_loss
        got = scores.float().cpu()
        am, ram = got.argmax(-1), ref_scores.argmax(-1)
        flips = (am != ram).nonzero().flatten().tolist()
        mean_rel = abs(loss.mean().item() - float(ref_loss.mean())) / abs(float(ref_loss.mean()))
        if task == "itm":
            margins = [abs(float(ref_scores[i, 1] - ref_scores[i, 0])) for i in flips]
            print("bf16 128 pairs ITM: %d labels, %d flips (oracle logit margins %s), median margin of all labels %.4f, mean-loss rel %.2e"
                  % (B, len(flips), ["%.2e" % m for m in margins], float((ref_scores[:, 1] - ref_scores[:, 0]).abs().median()), mean_rel))
            assert len(flips) <= B // 32 and all(m < 2e-2 for m in margins)        # a flip is only acceptable on a near-tie
        else:
            n_tok = ram.numel()
            top2 = ref_scores.topk(2, -1).values
            margins = [float(top2[i, 0] - top2[i, 1]) for i in flips]
            agree = 1.0 - len(flips) / n_tok
            print("bf16 128 pairs MLM: %d masked tokens, argmax agreement %.4f, largest oracle top-2 margin among the %d flips %.3e, mean-loss rel %.2e"
                  % (n_tok, agree, len(flips), max(margins) if margins else 0.0, mean_rel))
            assert agree >= 0.9
        assert mean_rel < 5e-3
    # batch independence through the persistent kernels at the bench size: the first 128 pairs of a 1024-pair evaluation forward
    # against the same 128 pairs run alone.  Not bit for bit -- the plans pick other kernels at M = 98 304 than at 12 288 for some
    # shapes and the ping-pong kernel starts its fp32 sums at the bias -- but within a few bf16 ulps of the logits (measured
    # 6.8e-3 on values of ~0.25) and with identical labels.
    model.eval()
    big = synth.make_batch(250002, 1024, 60, 36, task="itm", seed=12)
    small = {k: (v[:128].clone() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1024 else v) for k, v in strip(big).items()}
    with torch.no_grad():
        s_big = model(to_dev(big), "itm", compute_loss=False)
        s_small = model(to_dev(small), "itm", compute_loss=False)
    s_big = s_big[0] if isinstance(s_big, tuple) else s_big
    s_small = s_small[0] if isinstance(s_small, tuple) else s_small
    assert torch.isfinite(s_big).all()
    dmax = float((s_big[:128].float() - s_small.float()).abs().max())
    print("bf16 1024-pair forward, rows 0..127 vs the 128 pairs alone: max |logit diff| %.3e" % dmax)
    assert dmax < 2e-2 and torch.equal(s_big[:128].argmax(-1), s_small.argmax(-1))
    del model
    torch.cuda.empty_cache()


def test_bf16_measured_kernels_gradients_vs_oracle_at_176_pairs():
    """GRADIENTS through the measured kernels against the oracle's backward (not against another path of this repo): 176 pairs =
    16 896 tokens >= knobs.wgrad_side_min_rows, so the backward is the one bench.py times -- input gradients on the k-contiguous
    W^T copies (variant 12 with the gelu'-multiply / residual epilogues), weight gradients as split-K launches of the ping-pong
    kernel on the side stream, attention backward with the fused q|k|v bias gradient, LayerNorm backward with the dense-bias column
    sums.  12 layers, vocabulary 250 002, dropout 0, same weights and batch on both sides; the oracle's autograd runs on the box's
    host cores.  Bounds = the bf16 bounds of the B = 4 golden test (test_bf16_base_12_layers_vs_golden): L2 error of a gradient
    tensor <= 7 % (ITM) / 3 % (MLM); measured values are printed.  Reference: model/layer.py:159-170, model/model.py:571-598,690-735."""
    B = 176
    M = B * 96
    assert M >= knobs.wgrad_side_min_rows and knobs.wgrad_side_stream and knobs.dgrad_transposed_w
    model = build_pretrain(O.BASE, torch.bfloat16)
    W = _oracle_weights(model)
    cfg = _oracle_cfg(O.BASE)
    names = ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.0.attention.self.value.bias",
             "roberta.encoder.layer.5.intermediate.dense.weight", "roberta.encoder.layer.5.intermediate.dense.bias",
             "roberta.encoder.layer.11.output.dense.weight", "roberta.encoder.layer.11.attention.output.dense.weight",
             "roberta.encoder.layer.11.output.LayerNorm.weight", "roberta.encoder.layer.0.attention.output.LayerNorm.weight",
             "roberta.img_embeddings.img_linear.weight", "roberta.embeddings.LayerNorm.weight",
             "roberta.embeddings.position_embeddings.weight")      # the last two: dx of layer 0 as it reaches the embeddings
    for task, bound in (("itm", 0.03), ("mlm", 0.03)):       # measured 0.7-1.5 % (ITM) and 0.9-1.5 % (MLM)
        batch = synth.make_batch(250002, B, 60, 36, task=task, seed=21)
        _, _, loss = run_task(model, batch, task)
        ops.join_side_streams()
        torch.cuda.synchronize()
        # the backward really took the measured route: every encoder GEMM shape of this token count is planned on the ping-pong family
        for (ta, tb, n, k) in [(False, False, 2304, 768), (False, False, 768, 768), (False, False, 3072, 768), (False, False, 768, 3072)]:
            assert ops.gemm_plan(torch.bfloat16, ta, tb, M, n, k)[0] in (8, 9, 12), (ta, tb, n, k)

        def loss_fn(Wg):
            l = O.pretrain_forward(Wg, cfg, strip(batch), task)
            l = l[0] if isinstance(l, tuple) else l
            return l.mean()
        ref_loss, ref_grads = O.grads_of(loss_fn, W, names=set(names))
        assert abs(loss.mean().item() - float(ref_loss)) / abs(float(ref_loss)) < 5e-3
        P = dict(model.named_parameters())
        for n in names:
            g = P[n].grad
            assert g is not None and torch.isfinite(g).all(), n
            e = rel_err(g.float().cpu(), ref_grads[n])
            print("bf16 %d pairs %s grad L2 rel %-62s %.4g" % (B, task, n.split("roberta.")[1], e))
            assert e < bound, "%s %s: %.3e" % (task, n, e)
        del ref_grads
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_large_geometry_vs_golden(dtype):
    """BASELINE.json configs[4] geometry: 24L / 1024H / 16 heads / 4096 FFN, 80 tokens + 50 regions (L = 130), B = 2,
    against vectors the reference produced through VLXLMRConfig.from_dict (tests/golden/golden_large.npz).
    fp32: north-star tolerance, argmax bit-exact; bf16: reported, ITM labels exact"""
    g = golden("large")
    model = build_pretrain(O.LARGE, dtype)
    f32 = dtype == torch.float32
    for task in ("itm", "mlm"):
        batch = synth.make_batch(250002, 2, 80, 50, task=task, seed=1)
        seq, scores, loss = run_task(model, batch, task)
        key = "large2/%s" % task
        e1 = check_against_golden(g, key + "/seq", seq, TOL32 if f32 else 0.1)
        e2 = check_against_golden(g, key + "/loss", loss, TOL32 if f32 else 6e-2)
        agree = float((scores.argmax(-1).cpu().numpy() == g[key + "/argmax"]).mean())
        print("large %s %s: seq %.3g loss %.3g argmax agreement %.3f" % (str(dtype)[6:], task, e1, e2, agree))
        if f32 or task == "itm":
            assert agree == 1.0
        else:
            assert agree >= 0.85
        P = dict(model.named_parameters())
        for name in ("roberta.encoder.layer.0.attention.self.query.weight", "roberta.encoder.layer.23.output.dense.weight",
                     "roberta.img_embeddings.img_linear.weight"):
            if f32:
                check_against_golden(g, "%s/grad/%s" % (key, name), P[name].grad, 3e-3)
            else:
                check_against_golden(g, "%s/grad/%s" % (key, name), P[name].grad, 0.15 if task == "mlm" else 0.5, metric="l2")
    del model
    torch.cuda.empty_cache()


def test_large_geometry_fp8_vs_golden():
    """BASELINE.json configs[4]: 24L / 1024H, L = 130, fp8 (e4m3) forward and input-gradient GEMMs with bf16 weight
    gradients, against the reference's fp32 vectors at fp8 resolution (3 mantissa bits per operand, fp32 accumulation):
    the measured errors are printed; the loss must stay within a few per cent and the ITM labels must not flip"""
    g = golden("large")
    model = build_pretrain(O.LARGE, torch.bfloat16)
    uc2_amd.set_fp8(model, True)
    for task in ("itm", "mlm"):
        batch = synth.make_batch(250002, 2, 80, 50, task=task, seed=1)
        seq, scores, loss = run_task(model, batch, task)
        key = "large2/%s" % task
        e1 = check_against_golden(g, key + "/seq", seq, 0.15, metric="l2")    # 96 GEMMs of e4m3 operands deep; measured 0.10-0.11
        ref_mean = float(g[key + "/loss/sum3"][0]) / loss.numel()
        e2 = abs(loss.mean().item() - ref_mean) / abs(ref_mean)
        agree = float((scores.argmax(-1).cpu().numpy() == g[key + "/argmax"]).mean())
        P = dict(model.named_parameters())
        e3 = check_against_golden(g, "%s/grad/roberta.encoder.layer.23.output.dense.weight" % key,
                                  P["roberta.encoder.layer.23.output.dense.weight"].grad, 0.35 if task == "itm" else 0.2, metric="l2")
        # (measured 0.29 for ITM -- at initialisation the ITM gradient is a difference of large cancelling terms, the bf16 run of the
        #  tiny config needs 0.25 for it too -- and 0.13 for MLM)
        print("large fp8 %s: seq slice L2 rel %.3g, mean-loss rel %.3g, argmax agreement %.3f, last-layer grad L2 rel %.3g" % (task, e1, e2, agree, e3))
        assert e2 < 3e-2
        if task == "itm":
            assert agree == 1.0
        else:
            assert agree >= 0.9                                             # measured 0.917
    del model
    torch.cuda.empty_cache()


def _fp8_routes(reset=False):
    lib = uc2_amd._lib.load()
    return int(lib.uc2_gemm_fp8_route_count(0, int(reset))), int(lib.uc2_gemm_fp8_route_count(1, int(reset)))      # (ring, ping-pong)


def test_large_geometry_fp8_pingpong_kernels_and_delayed_scaling_vs_oracle():
    """configs[4] through the e4m3 PING-PONG kernel (gemm_pp8.hip) against the CPU ORACLE, not against this repo's bf16 path
    (VERDICT r5 weak #2): 24L / 1024H / 16 heads / 4096, 80 tokens + 48 regions (L = 128), 32 pairs = 4 096 tokens = 16 whole
    256-row tiles, so every e4m3 GEMM of the layer runs on gemm_pp8 (asserted through uc2_gemm_fp8_route_count; the B = 2 golden
    test above runs 260 tokens on the ring kernel).  Step 1 with just-in-time scales, steps 2-3 with delayed scaling (one-pass
    quantisation with half the scale of the previous use's maximum).  Same weights and batch on both sides; the oracle's forward and
    autograd run on the box's host cores.  The bf16 run of the same step is measured against the oracle too, so the fp8 bound on
    the last-layer ITM gradient can be read against what bf16 itself needs there: at initialisation the ITM logits are ~0 and the
    gradient is a difference of large cancelling terms -- bf16 alone shows a several-per-cent error on that tensor, e4m3 (3 mantissa
    bits per operand, 96 GEMMs deep) a few times that; the MLM gradient, which is not a cancellation, is held to 0.2."""
    geom = dict(O.LARGE)
    T, R, B = 80, 48, 32
    assert (B * (T + R)) % 256 == 0
    model = build_pretrain(geom, torch.bfloat16)
    W = _oracle_weights(model)
    cfg = _oracle_cfg(geom)
    name = "roberta.encoder.layer.23.output.dense.weight"
    P = dict(model.named_parameters())
    for task, gbound in (("itm", 0.25), ("mlm", 0.18)):         # measured 0.19-0.20 / 0.148 (bf16 on the same tensors: 0.021 / 0.013)
        batch = synth.make_batch(250002, B, T, R, task=task, seed=3)

        def loss_fn(Wg):
            l = O.pretrain_forward(Wg, cfg, strip(batch), task)
            return (l[0] if isinstance(l, tuple) else l).mean()
        ref_loss, ref_grads = O.grads_of(loss_fn, W, names={name})
        with torch.no_grad():
            ref_scores = O.pretrain_forward(W, cfg, strip(batch), task, compute_loss=False)
            ref_scores = ref_scores[0] if isinstance(ref_scores, tuple) else ref_scores
        ram = ref_scores.argmax(-1)
        errs = {}
        for mode in ("bf16", "fp8", "fp8", "fp8"):
            uc2_amd.set_fp8(model, mode == "fp8")
            if mode == "fp8" and "fp8" not in errs:
                _fp8_routes(reset=True)
            _, scores, loss = run_task(model, batch, task)
            ops.join_side_streams()
            torch.cuda.synchronize()
            e_l = abs(loss.mean().item() - float(ref_loss)) / abs(float(ref_loss))
            e_g = rel_err(P[name].grad.float().cpu(), ref_grads[name])
            got = scores.float().cpu()
            am = got.argmax(-1)
            agree = float((am == ram).float().mean())
            # how far below the oracle's best logit the kernel's choice sits IN THE ORACLE'S OWN logits (0 where the labels agree)
            deficit = (ref_scores.max(-1).values - ref_scores.gather(-1, am.unsqueeze(-1)).squeeze(-1))
            errs.setdefault(mode, []).append((e_l, agree, e_g, float(deficit.max()), float(deficit.mean())))
            print("large %s %s at %d pairs vs ORACLE: mean-loss rel %.3g, argmax agreement %.3f (oracle-logit deficit of the chosen label: max %.3g, "
                  "mean %.3g; oracle logit std %.3g), last-layer grad L2 rel %.3g"
                  % (mode, task, B, e_l, agree, float(deficit.max()), float(deficit.mean()), float(ref_scores.std()), e_g))
        ring, pp = _fp8_routes()
        # 3 fp8 passes x (forward-only scoring + training forward + backward) x 24 layers: every e4m3 GEMM on the ping-pong kernel
        assert ring == 0 and pp >= 3 * 24 * (4 + 4 + 4), (ring, pp)
        sd = float(ref_scores.std())
        for (e_l, agree, e_g, dmax, dmean) in errs["fp8"]:
            assert e_l < 1e-2 and e_g < gbound                    # measured: mean loss 4e-3 (ITM) / 5e-4 (MLM)
            if task == "itm":
                assert agree == 1.0                               # 32 / 32 labels
            else:
                # MLM at initialisation: 250 002 nearly flat logits (std ~0.6, top-2 gaps ~0.1) behind a 24-layer e4m3 encoder whose
                # hidden states carry ~10 % L2 error -- the arg-max is decided by differences smaller than that noise.  Measured
                # agreement 0.65-0.71 over ~300 masked tokens (bf16: 0.93).  What is asserted: every flip is a near-tie IN THE ORACLE'S
                # OWN LOGITS (the chosen label's oracle logit is within one logit std of the oracle's best, a tenth of one on average);
                # a wrong kernel would pick labels several std below.
                assert agree >= 0.6 and dmax < 1.0 * sd and dmean < 0.1 * sd, (agree, dmax, dmean, sd)
        del ref_grads
    uc2_amd.set_fp8(model, False)
    del model
    torch.cuda.empty_cache()


def test_fp8_mode_trains_like_the_bf16_mode():
    """fp8 mode over a training RUN (VERDICT r5 #5a), in the form of test_bf16_bench_path_trains_like_the_fp32_parity_mode: a 2-layer
    model of uc2-large width (1024 / 16 heads / 4096, vocabulary 2000) at 128 pairs x 130 tokens = 16 640 tokens (whole 256-row
    tiles: every e4m3 GEMM on gemm_pp8.hip), trained for 24 optimizer steps -- ITM and MLM alternating, i.e. delayed scaling with
    one amax history per task and role, four batches cycling, clip 5.0, AdamW re-quantising the e4m3 weight copies every step,
    dropout off, lr 1e-5 -- from the same initial weights in bf16 and with fp8 GEMMs: the MLM curve falls in both modes, and the two
    runs stay together step by step (MLM steps within 1 %, ITM steps within 5 %; measured values printed, profiles/r06_experiments.md)."""
    geom = dict(O.LARGE, num_hidden_layers=2, vocab_size=2000)
    B, T, R = 128, 80, 50
    assert (B * (T + R)) % 256 == 0
    batches = [(t, to_dev(synth.make_batch(2000, B, T, R, task=t, seed=170 + i))) for i, t in enumerate(("itm", "mlm", "itm", "mlm"))]
    curves = {}
    for mode in ("bf16", "fp8"):
        model = build_pretrain(geom, torch.bfloat16)
        uc2_amd.set_fp8(model, mode == "fp8")
        opt = AdamW(param_groups(model, 0.01), lr=1e-5, betas=(0.9, 0.98))
        _fp8_routes(reset=True)
        losses = []
        for step in range(24):
            task, b = batches[step % 4]
            opt.zero_grad()
            loss = model(b, task, compute_loss=True)
            loss = (loss[0] if isinstance(loss, tuple) else loss).mean()
            loss.backward()
            clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0)
            opt.step()
            losses.append(float(loss.detach()))
        ring, pp = _fp8_routes()
        assert (ring, pp > 0) == ((0, True) if mode == "fp8" else (0, False)), (mode, ring, pp)
        curves[mode] = losses
        del model, opt
    b16, f8 = curves["bf16"], curves["fp8"]
    assert all(l == l for l in b16 + f8)
    print("bf16", [round(v, 4) for v in b16])
    print("fp8 ", [round(v, 4) for v in f8])
    print("max rel gap %.4f" % max(abs(a - c) / abs(a) for a, c in zip(b16, f8)))
    assert b16[21] < b16[1] and f8[21] < f8[1]                 # the MLM loss of the first batch pair falls in both modes
    # (the ITM targets of a synthetic batch are coin flips: its loss starts at ln 2 and only moves once the batch is memorised --
    #  it is compared step by step, not asked to fall)
    for s_, (a, c) in enumerate(zip(b16, f8)):
        bound = 0.01 if s_ % 2 else 0.05                       # MLM steps (odd): 1 %; ITM steps: 5 % (first run of this test, lr 5e-5,
        assert abs(a - c) <= bound * abs(a) + 0.005, (s_, a, c)  # where ITM overshoots to 2.5: MLM 0.06 %, ITM 7.9 %)


def test_submodule_forwards_compose_to_the_fused_layer():
    """BertSelfAttention / BertSelfOutput / BertAttention / BertIntermediate / BertOutput have working forward()s
    (model/layer.py:75-156); composed the reference's way they reproduce the fused BertLayerFn node bit for bit in
    fp32 (same kernels, same order) -- outputs and every gradient"""
    for dtype in (torch.float32, torch.bfloat16):
        cfg = make_cfg(O.TINY)
        layer = BertLayer(cfg)
        synth.det_init_(layer)
        layer.to(DEV).train()
        set_compute_dtype(layer, dtype)
        x = synth.det_normal((3, 68, 128), 21).to(DEV).to(dtype)
        am = torch.ones(3, 68, dtype=torch.long)
        am[1, 50:] = 0
        ext = O.extended_mask(am).to(DEV)
        dy = synth.det_normal((3, 68, 128), 22).to(DEV).to(dtype)
        x1 = x.clone().requires_grad_(True)
        layer.zero_grad()
        y1 = layer(x1, ext)
        y1.backward(dy)
        g1 = {n: p.grad.clone() for n, p in layer.named_parameters()}
        for p in layer.parameters():
            p.grad = None
        uc2_amd.store.store_of(layer).zero_grad()
        x2 = x.clone().requires_grad_(True)
        y2 = layer.forward_unfused(x2, ext)
        y2.backward(dy)
        assert torch.equal(y1, y2)
        assert rel_err(x2.grad.float(), x1.grad.float()) < (1e-6 if dtype == torch.float32 else 1e-2)
        for n, p in layer.named_parameters():
            if n.endswith("key.bias"):
                continue
            assert rel_err(p.grad, g1[n]) < (1e-5 if dtype == torch.float32 else 2e-2), n


def test_gelu_module_and_sequential_heads():
    """GELU is a real activation (model/layer.py:31-50): calling the head Sequentials the plain way equals the
    fused-epilogue path the model uses"""
    from uc2_amd.model.layer import GELU
    model = build_pretrain(O.TINY, torch.float32)
    x = synth.det_normal((50, 128), 31).to(DEV)
    ref = torch.nn.functional.gelu(x)
    assert max_rel(GELU()(x), ref) < 1e-5
    rc = model.region_classifier
    assert max_rel(rc.net(x), rc(x)) < 1e-5
    fr = model.feat_regress
    h = fr.net(x)
    assert max_rel(h, fr.net[2](fr.net[0](x, act=ops.EPI_GELU))) < 1e-5
    xg = x.clone().requires_grad_(True)
    GELU()(xg).sum().backward()
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.gelu(xr).sum().backward()
    assert max_rel(xg.grad, xr.grad) < 1e-4


def test_checkpoint_load_reproduces_golden_and_resume_matches(tmp_path):
    """f3: a reference-layout checkpoint dict (legacy gamma/beta names) through from_pretrained reproduces the
    reference's own outputs; ModelSaver + train_state files resume a run onto the same trajectory"""
    from uc2_amd.utils.save import ModelSaver
    g = golden("tiny")
    cfgf = tmp_path / "cfg.json"
    cfgf.write_text(make_cfg(O.TINY).to_json_string())
    src = VLXLMRForPretraining(make_cfg(O.TINY), img_dim=2048, img_label_dim=1601)
    synth.det_init_(src)
    from collections import OrderedDict as OD
    sd = OD((k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"), v.clone())
            for k, v in src.state_dict().items())
    model = VLXLMRForPretraining.from_pretrained(str(cfgf), sd, img_dim=2048, img_label_dim=1601)
    model.to(DEV).train()
    b = to_dev(synth.make_batch(1000, 8, 32, 36, task="mlm", seed=1))
    loss = model(b, "mlm", compute_loss=True)
    check_against_golden(g, "tiny8/mlm/loss", loss, TOL32)
    # ---- resume: 2 steps, save, 1 more step  ==  load + 1 step
    def make_opt(m):
        return AdamW(param_groups(m, 0.01), lr=1e-3, betas=(0.9, 0.98))

    def one_step(m, opt, seed):
        bb = to_dev(synth.make_batch(1000, 4, 32, 36, task="itm", seed=seed))
        opt.zero_grad()
        l, _ = m(bb, "itm")
        l.mean().backward()
        opt.step()
    opt = make_opt(model)
    one_step(model, opt, 31)
    one_step(model, opt, 32)
    ModelSaver(str(tmp_path)).save(model, 2, opt)
    one_step(model, opt, 33)
    m2 = VLXLMRForPretraining.from_pretrained(str(cfgf), torch.load(tmp_path / "model_step_2.pt"), img_dim=2048, img_label_dim=1601)
    m2.to(DEV).train()
    opt2 = make_opt(m2)
    opt2.load_state_dict(torch.load(tmp_path / "train_state_2.pt")["optimizer"])
    one_step(m2, opt2, 33)
    for (n, p), (_, q) in zip(model.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p, q, rtol=0, atol=2e-6), n          # (fp32 atomics in the embedding backward: not bit-exact)
    w = "roberta.encoder.layer.0.output.dense.weight"
    assert not torch.equal(dict(model.named_parameters())[w], src.state_dict()[w].to(DEV))


def test_retrieval_inference_vs_golden():
    """f2: forward-only scoring into the fp16 score matrix (itm.py:516-538) and device-side recall (eval/itm.py:6-53)
    against the reference's score matrix and its own itm_eval numbers; the forward-only layer equals the training
    forward"""
    from uc2_amd.eval.itm import inference, itm_eval, validate
    g = golden("retrieval")
    model = VLXLMRForImageTextRetrieval(make_cfg(O.TINY), img_dim=2048, margin=0.2)
    synth.det_init_(model)
    model.to(DEV)
    n_txt, n_img = 12, 11
    pool = synth.retrieval_pool(1000, n_txt, n_img, 32, 36)
    loader = [[to_dev(synth.retrieval_batch(pool, i, j0, min(n_img, j0 + 4))) for j0 in range(0, n_img, 4)] for i in range(n_txt)]
    sm = inference(model, loader, n_txt, n_img)
    assert sm.dtype == torch.float16 and tuple(sm.shape) == (n_txt, n_img)
    ref = torch.from_numpy(g["retrieval/scores"])
    assert max_rel(sm.float().cpu(), ref.half().float()) < 2e-3          # fp16 storage of the scores
    txt_ids = ["t%d" % i for i in range(n_txt)]
    img_ids = ["i%d" % j for j in range(n_img)]
    txt2img = {t: img_ids[(3 * k) % n_img] for k, t in enumerate(txt_ids)}
    img2txts = {j: [t for t in txt_ids if txt2img[t] == j] for j in img_ids}
    ev_in = torch.from_numpy(g["retrieval/eval_input"]).half().to(DEV)     # tie-free (topk's order of equal scores is unspecified)
    log = itm_eval(ev_in, txt_ids, img_ids, txt2img, img2txts)
    for k, v in log.items():
        assert abs(v - float(g["retrieval/eval/" + k][0])) < 1e-6, (k, v, float(g["retrieval/eval/" + k][0]))
    plain = itm_eval(ev_in, txt_ids, img_ids, txt2img, img2txts, reference_row_term=False)
    assert plain["img_r10"] <= 1.0
    # forward-only path == training-graph path in eval mode (same kernels minus the saved streams)
    b = loader[3][1]
    model.eval()
    with torch.no_grad():
        s_inf = model(b, compute_loss=False)
    s_trn = model(b, compute_loss=False)
    assert torch.equal(s_inf, s_trn.detach())
    v = validate(model, [loader[0][0], loader[1][0]])
    assert set(v) == {"valid/recall_1", "valid/recall_5", "valid/recall_10"}


def test_hard_negative_forward_matches_manual_selection():
    """configs[3] pattern (model/itm.py:105-186 restated for the VL-XLM-R model): no-grad eval scoring -> top-k ->
    training forward on the selected sub-batch, against doing the same three steps by hand"""
    from uc2_amd.model.itm import VLXLMRForImageTextRetrievalHardNeg
    torch.manual_seed(0)
    model = VLXLMRForImageTextRetrievalHardNeg(make_cfg(O.TINY), img_dim=2048, margin=0.2, hard_size=3)
    synth.det_init_(model)
    model.to(DEV).train()
    pool = synth.retrieval_pool(1000, 2, 9, 32, 36)
    b = to_dev(synth.retrieval_batch(pool, 0, 0, 9))
    b["input_ids"] = b["input_ids"][:1]                      # one text, expanded by the model
    loss = model(dict(b), sample_from='t', compute_loss=True)
    assert tuple(loss.shape) == (1, 3)
    loss.mean().backward()
    assert model.rank_output.weight.grad is not None
    plain = VLXLMRForImageTextRetrieval(make_cfg(O.TINY), img_dim=2048, margin=0.2)
    synth.det_init_(plain)
    plain.to(DEV).eval()
    full = dict(b, input_ids=b["input_ids"].expand(9, -1))
    with torch.no_grad():
        sc = plain(full, compute_loss=False).reshape(-1)
    hard = sc[1:].topk(3, sorted=False)[1] + 1
    idx = torch.cat([torch.zeros(1, dtype=torch.long, device=DEV), hard])
    attn = b["attn_masks"].index_select(0, idx)
    ml = int(attn.sum(1).max().item())
    mi = ml - b["input_ids"].size(1)
    hb = dict(sample_size=4, input_ids=full["input_ids"][:4], img_feat=b["img_feat"].index_select(0, idx)[:, :mi],
              img_pos_feat=b["img_pos_feat"].index_select(0, idx)[:, :mi], attn_masks=attn[:, :ml],
              gather_index=b["gather_index"].index_select(0, idx)[:, :ml])
    plain.train()
    want = plain(hb, compute_loss=True)
    assert torch.allclose(loss, want, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------ optimizer
def test_adamw_clip_vs_golden():
    """3 optimizer steps x 3 summed micro-batches, both param groups, clipping, per-parameter step
    counts (heads untouched by a task keep grad None and are skipped) -- against the reference's AdamW."""
    g = golden("adamw")
    model = build_pretrain(O.TINY, torch.float32)
    groups = param_groups(model, 0.01)
    names = dict((id(p), n) for n, p in model.named_parameters())
    assert [names[id(p)] for p in groups[1]["params"]] == list(g["adamw/no_decay_names"])
    opt = AdamW(groups, lr=4e-5, betas=(0.9, 0.98))
    P = dict(model.named_parameters())
    for step in range(1, 4):
        lr = O.warmup_linear(step, 2, 10) * 4e-5 + 1e-5
        for grp in opt.param_groups:
            grp["lr"] = lr
        task = ["itm", "mlm", "mrfr"][step - 1]
        opt.zero_grad()
        for micro in range(3):
            b = to_dev(synth.make_batch(1000, 4, 32, 36, task=task, seed=10 * step + micro))
            loss = model(b, task, compute_loss=True)
            loss = loss[0] if isinstance(loss, tuple) else loss
            loss.mean().backward()
        gn = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0 if step > 1 else 0.05)
        ref_gn = float(g["adamw/step%d/grad_norm" % step][0])
        assert abs(gn.item() - ref_gn) < 2e-3 * ref_gn
        opt.step()
        pre = "adamw/step%d/" % step
        for key in g.files:
            if key.startswith(pre) and key.endswith("/sum3") and "_norm" not in key:
                n = key[len(pre):-len("/sum3")]
                check_against_golden(g, pre + n, P[n].data, 2e-5)
        tot = torch.sqrt(sum((p.data.double() ** 2).sum() for p in model.parameters())).item()
        assert abs(tot - float(g[pre + "param_norm"][0])) < 2e-6 * tot
    # state_dict surface of the reference optimizer
    sd = opt.state_dict()
    some = next(iter(sd["state"].values()))
    assert set(some.keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_bf16_bench_path_trains_like_the_fp32_parity_mode():
    """The arithmetic mode bench.py times against the parity mode over a training RUN, not one step: a 2-layer base-width model
    (768 / 12 heads / 3072, vocabulary 2000) at 176 pairs x 96 tokens -- every route of the bench step is on at that size: fused
    dropout-residual tails, k-contiguous W^T input gradients, head-interleaved q|k|v, weight gradients on the side stream -- trained
    for 24 optimizer steps (ITM and MLM alternating, four batches cycling, clip 5.0, AdamW, dropout off) from the same initial
    weights in fp32 and in bf16: the two loss curves stay together and both fall."""
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    B, T, R = 176, 60, 36
    assert B * (T + R) >= max(knobs.ln_fuse_min_rows, knobs.wgrad_side_min_rows)
    batches = [(t, to_dev(synth.make_batch(2000, B, T, R, task=t, seed=70 + i))) for i, t in enumerate(("itm", "mlm", "itm", "mlm"))]
    curves = {}
    for dtype in (torch.float32, torch.bfloat16):
        model = build_pretrain(geom, dtype)
        opt = AdamW(param_groups(model, 0.01), lr=5e-5, betas=(0.9, 0.98))
        losses = []
        for step in range(24):
            task, b = batches[step % 4]
            opt.zero_grad()
            loss = model(b, task, compute_loss=True)
            loss = loss[0] if isinstance(loss, tuple) else loss
            loss = loss.mean()
            loss.backward()
            clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0)
            opt.step()
            losses.append(float(loss.detach()))
        curves[dtype] = losses
        del model, opt
    f32, b16 = curves[torch.float32], curves[torch.bfloat16]
    assert all(l == l for l in f32 + b16)
    print("fp32", [round(v, 4) for v in f32])
    print("bf16", [round(v, 4) for v in b16])
    assert f32[21] < f32[1] and b16[21] < b16[1]               # the MLM loss of the first batch pair falls in both modes
    for s_, (a, c) in enumerate(zip(f32, b16)):
        assert abs(a - c) <= 0.01 * abs(a) + 0.005, (s_, a, c)          # (measured: at most 0.3 % apart, step 3)


@pytest.mark.parametrize("task,pairs", [("itm", 24), ("mlm", 24), ("itm", 176)])
def test_accumulation_overlap_equals_the_sequential_loop(task, pairs):
    """ops.accum_pass (round 6): the reference's accumulation loop AS WRITTEN (pretrain.py:514-566: forward, backward, forward,
    backward, ...) with each training forward on one of the two overlap streams, so that forward i+1 runs beside backward i,
    against the same loop on one stream (knobs.accum_overlap off).  Same kernels, same accumulation order: identical losses, gradients
    equal to the fp32 order of the atomics (1e-6 fp32, 2e-5 bf16 -- what two runs on ONE stream differ by), fp32 and bf16; the passes really ran on the overlap streams; consumers of gradients (clip, optimizer) are ordered
    behind both streams; with dropout on, the three forwards draw distinct masks and equal the one-stream loop's."""
    from uc2_amd.store import store_of
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    if pairs == 24:
        batches = [to_dev(synth.make_batch(2000, 24, 40, 20, task=task, seed=90 + i, variable_len=(i == 1))) for i in range(3)]
    else:
        # 176 x 96 = 16 896 tokens: above the side-stream / fused-tail / W^T thresholds and below the overlap's (49 152) -- the passes
        # then also share the weight-gradient side stream (the retrieval finetune's 18 k-token windows run this combination)
        batches = [to_dev(synth.make_batch(2000, pairs, 60, 36, task=task, seed=90 + i)) for i in range(3)]
        assert knobs.wgrad_side_min_rows <= pairs * 96 < knobs.accum_overlap_max_rows

    def loop(model, bs):
        losses = []
        for b in bs:                                              # the reference's loop: nothing but forward + backward
            l = model(b, task, compute_loss=True)
            l = (l[0] if isinstance(l, tuple) else l).mean()
            l.backward()
            losses.append(l.detach())
        norm = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1e9)      # a consumer: joins the streams
        grads = OrderedDict((n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.grad is not None)
        torch.cuda.synchronize()
        return [float(l) for l in losses], grads, float(norm)
    was = knobs.accum_overlap
    try:
        for dtype, drop in ((torch.float32, 0.0), (torch.bfloat16, 0.0), (torch.bfloat16, 0.1)):
            # ONE model for both runs: the dropout sites are keyed by the layers' process-wide ids, two instances draw different masks
            model = VLXLMRForPretraining(make_cfg(geom, drop=drop), img_dim=2048, img_label_dim=1601)
            synth.det_init_(model)
            model.to(DEV).train()
            set_compute_dtype(model, dtype)
            st = store_of(model)
            if dtype == torch.bfloat16:
                st.sync_shadow()
                st.auto_sync = False                       # what AdamW.step does after the first optimizer step
            res = {}
            for overlap in (False, True):
                knobs.accum_overlap = overlap
                ops.rng.manual_seed(4242, DEV)
                model.zero_grad()
                before = sum(s_.passes for s_ in ops._accum.values())
                res[overlap] = loop(model, batches if drop == 0.0 else [batches[0]] * 3)
                ran = sum(s_.passes for s_ in ops._accum.values()) - before
                assert ran == (3 if overlap else 0), ran
            del model
            (l0, g0, n0), (l1, g1, n1) = res[False], res[True]
            assert l0 == l1 and abs(n0 - n1) <= 1e-5 * abs(n0) and set(g0) == set(g1)
            for n in g0:
                # equal to the fp32 order of the reductions, not bit for bit -- on one stream two runs differ the same way: the fp32
                # parity kernels' split-K weight gradients, the bias / column sums and the embedding rows are float atomics
                assert rel_err(g0[n], g1[n]) < (1e-5 if dtype == torch.float32 else 2e-5) or g0[n].norm() < 1e-7, n
            if drop:
                assert len(set(l1)) == 3, l1                      # three forwards of one batch: three different masks
    finally:
        knobs.accum_overlap = was


def test_accumulation_overlap_with_zero_grad_between_forward_and_backward():
    """the common loop `loss = model(b); optimizer.zero_grad(); loss.backward()` with the overlap on: the arena is zeroed on the
    CALLER's stream after the pass's forward was enqueued on an overlap stream -- the pass's backward must be ordered behind that
    (ops._AccumMarker waits for the caller's stream).  The arena holds another batch's gradients before; afterwards it must hold
    exactly this batch's (not the sum, not a partly zeroed mix): compared with the same sequence with the overlap off, three rounds."""
    from uc2_amd.store import store_of
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    b0 = to_dev(synth.make_batch(2000, 24, 40, 20, task="itm", seed=60))
    b1 = to_dev(synth.make_batch(2000, 24, 40, 20, task="itm", seed=61))
    model = build_pretrain(geom, torch.bfloat16)
    st = store_of(model)
    st.sync_shadow()
    st.auto_sync = False

    def fwd(b):
        l = model(b, "itm", compute_loss=True)
        return (l[0] if isinstance(l, tuple) else l).mean()
    was = knobs.accum_overlap
    res = {}
    try:
        for overlap in (False, True, True, True):
            knobs.accum_overlap = overlap
            model.zero_grad()
            fwd(b0).backward()                                # the arena now holds b0's gradients (on an overlap stream when on)
            before = sum(s_.passes for s_ in ops._accum.values())
            l = fwd(b1)
            model.zero_grad()                                 # AFTER the forward, on the caller's stream
            l.backward()
            assert (sum(s_.passes for s_ in ops._accum.values()) - before) == (1 if overlap else 0)
            g = OrderedDict((n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.grad is not None)    # (.grad: waits for the pass)
            torch.cuda.synchronize()
            if not overlap:
                res = g
                continue
            assert set(g) == set(res)
            for n in res:
                assert rel_err(g[n], res[n]) < 2e-5 or res[n].norm() < 1e-7, n
    finally:
        knobs.accum_overlap = was


def test_accumulation_overlap_stays_off_where_it_must():
    """ops.accum_pass is a plain pass on the caller's stream for eval / no-grad forwards, fp8 stores, stores that re-cast their bf16
    copies at every forward (store.auto_sync) and micro-batches of ACCUM_OVERLAP_MAX_ROWS tokens or more; utils.pipeline.accumulate
    (the round-5 opt-in API) is the plain loop on top of it."""
    from uc2_amd.store import set_fp8, store_of
    from uc2_amd.utils.pipeline import accumulate
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    model = build_pretrain(geom, torch.bfloat16)
    st = store_of(model)
    b = to_dev(synth.make_batch(2000, 8, 40, 20, task="itm", seed=5))

    def passes():
        return sum(s_.passes for s_ in ops._accum.values())

    def fwd():
        l = model(b, "itm", compute_loss=True)
        return (l[0] if isinstance(l, tuple) else l).mean()
    n0 = passes()
    assert st.auto_sync
    fwd().backward()                                            # auto_sync: the bf16 copies are re-cast by every forward
    assert passes() == n0
    st.sync_shadow()
    st.auto_sync = False
    fwd().backward()
    assert passes() == n0 + 1
    with torch.no_grad():
        fwd()
    model.eval()
    model(b, "itm", compute_loss=False)
    model.train()
    assert passes() == n0 + 1
    set_fp8(model, True)
    model.__dict__.pop("_uc2_fp8_probe", None)
    fwd().backward()
    set_fp8(model, False)
    assert passes() == n0 + 1
    was = knobs.accum_overlap_max_rows
    knobs.accum_overlap_max_rows = 8 * 60                         # this batch is 8 x 60 tokens: at the threshold -> off
    try:
        fwd().backward()
    finally:
        knobs.accum_overlap_max_rows = was
    assert passes() == n0 + 1
    order = []
    out = accumulate([fwd, fwd], before_backward=order.append)
    assert order == [0, 1] and len(out) == 2 and passes() == n0 + 3
    ops.join_side_streams()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_accumulation_overlap_adapts_to_loops_without_accumulation():
    """one forward per optimizer step (retrieval loops, plain fine-tuning): nothing to run beside, the stream hops only cost
    (profiles/r06_experiments.md section 1) -- after ONE such window the passes stay on the caller's stream; the second forward of a
    later window switches the overlap back on, and from the window after that every pass is overlapped again.  Gradients are the
    same either way (compared with the overlap off)."""
    from uc2_amd.store import store_of
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    model = build_pretrain(geom, torch.bfloat16)
    st = store_of(model)
    st.sync_shadow()
    st.auto_sync = False
    bs = [to_dev(synth.make_batch(2000, 16, 40, 20, task="itm", seed=70 + i)) for i in range(3)]

    def passes():
        return sum(s_.passes for s_ in ops._accum.values())

    def window(n):
        model.zero_grad()
        before = passes()
        for b in bs[:n]:
            l = model(b, "itm", compute_loss=True)
            (l[0] if isinstance(l, tuple) else l).mean().backward()
        norm = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1e9)       # the consumer that ends the window
        return passes() - before, float(norm)
    was = knobs.accum_overlap
    try:
        knobs.accum_overlap = False
        want = {n: window(n)[1] for n in (1, 3)}
        knobs.accum_overlap = True
        ops.forget_accum_history()
        got = [window(n) for n in (1, 1, 1, 3, 3, 1, 1)]
        assert [g[0] for g in got] == [1, 0, 0, 2, 3, 1, 0], got
        for n, (_, norm) in zip((1, 1, 1, 3, 3, 1, 1), got):
            assert abs(norm - want[n]) <= 2e-5 * want[n], (n, norm, want[n])
    finally:
        knobs.accum_overlap = was


def test_adamw_bf16_shadow_and_fused_clip():
    model = build_pretrain(O.TINY, torch.bfloat16)
    opt = AdamW(param_groups(model, 0.01), lr=1e-3, betas=(0.9, 0.98))
    b = to_dev(synth.make_batch(1000, 4, 32, 36, task="itm", seed=3))
    loss, _ = model(b, "itm")
    loss.mean().backward()
    st = uc2_amd.store.store_of(model)
    norm, coef = clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1e-3, fused=True)
    before = st.data.clone()
    opt.step(grad_scale=coef, zero_grad=True)
    assert not torch.equal(before, st.data)
    assert st.shadow_version == st.version
    w = model.roberta.encoder.layer[0].output.dense.weight
    assert torch.equal(st.view(st.shadow, w), w.data.to(torch.bfloat16))
    assert all(p.grad is None for p in model.parameters())
    assert st.grad.abs().max().item() == 0.0


# ------------------------------------------------------------------------------------------ full-size properties
def test_base_geometry_bf16_properties():
    """size-independent properties at the bench geometry: batch rows are independent (permutation
    equivariance), padded keys do not influence valid rows, eval is deterministic"""
    model = build_pretrain(O.BASE, torch.bfloat16).eval()
    B = 16
    batch = to_dev(synth.make_batch(250002, B, 60, 36, task="itm", seed=5))
    with torch.no_grad():
        s1, _ = model(batch, "itm", compute_loss=False)
        perm = torch.randperm(B, device=DEV)
        pb = {k: (v[perm] if torch.is_tensor(v) and v.shape[0] == B else v) for k, v in batch.items()}
        s2, _ = model(pb, "itm", compute_loss=False)
        assert torch.equal(s1[perm], s2)
        s3, _ = model(batch, "itm", compute_loss=False)
        assert torch.equal(s1, s3)
        # mask the last 6 regions of every pair and perturb them: valid-row outputs must not move
        am = batch["attn_masks"].clone()
        am[:, -6:] = 0
        b2 = dict(batch, attn_masks=am)
        h1 = model.roberta(b2["input_ids"], None, b2["img_feat"], b2["img_pos_feat"], am, b2["gather_index"],
                           output_all_encoded_layers=False)
        f2 = b2["img_feat"].clone()
        f2[:, -6:] += 3.0
        h2 = model.roberta(b2["input_ids"], None, f2, b2["img_pos_feat"], am, b2["gather_index"],
                           output_all_encoded_layers=False)
        assert torch.equal(h1[:, :-6], h2[:, :-6])
    del model
    torch.cuda.empty_cache()


def test_multihead_attention_vs_golden():
    """reference model/attention.py MultiheadAttention (NLVR2 API surface): output and averaged weights"""
    from uc2_amd.model.attention import MultiheadAttention
    g = golden("mha")
    E, nh, L, N = 128, 4, 10, 3
    m = MultiheadAttention(E, nh, dropout=0.0)
    synth.det_init_(m)
    m.to(DEV).train()
    q = synth.det_normal((L, N, E), 77).to(DEV)
    kpm = torch.zeros(N, L, dtype=torch.bool, device=DEV)
    kpm[1, 7:] = True
    kpm[2, 4:] = True
    qg = q.clone().requires_grad_(True)
    o, w = m(qg, qg, qg, key_padding_mask=kpm)
    check_against_golden(g, "mha/out", o, TOL32)
    check_against_golden(g, "mha/weights", w, TOL32)
    g2 = golden("more")
    for p in m.parameters():
        p.grad = None
    qg = q.clone().requires_grad_(True)
    o, _ = m(qg, qg, qg, key_padding_mask=kpm)
    (o * synth.det_normal((L, N, E), 78).to(DEV)).sum().backward()
    check_against_golden(g2, "mha/dq", qg.grad, 3e-3)                      # backward parity (reference gradients)
    for n, p in m.named_parameters():
        check_against_golden(g2, "mha/grad/" + n, p.grad, 3e-3)


def test_multihead_attention_cross_and_attn_mask_vs_golden():
    """general MultiheadAttention (model/attention.py:12-264): separate key / value inputs (S != L), additive attn_mask,
    key padding; output, head-averaged weights and every gradient against the reference"""
    from uc2_amd.model.attention import MultiheadAttention
    g = golden("more")
    E, nh, L, N, S = 128, 4, 10, 3, 7
    m = MultiheadAttention(E, nh, dropout=0.0)
    synth.det_init_(m)
    m.to(DEV).train()
    q2, k2, v2 = [synth.det_normal(shp, sd).to(DEV).requires_grad_(True) for shp, sd in (((L, N, E), 80), ((S, N, E), 81), ((S, N, E), 82))]
    amask = synth.det_normal((L, S), 83).to(DEV)
    kpm2 = torch.zeros(N, S, dtype=torch.bool, device=DEV)
    kpm2[2, 5:] = True
    o2, w2 = m(q2, k2, v2, key_padding_mask=kpm2, attn_mask=amask)
    check_against_golden(g, "mha_cross/out", o2, TOL32)
    check_against_golden(g, "mha_cross/weights", w2, TOL32)
    (o2 * synth.det_normal((L, N, E), 84).to(DEV)).sum().backward()
    for key, t in (("dq", q2), ("dk", k2), ("dv", v2)):
        check_against_golden(g, "mha_cross/" + key, t.grad, 3e-3)
    for n, p in m.named_parameters():
        check_against_golden(g, "mha_cross/grad/" + n, p.grad, 3e-3)


def test_multihead_attention_cross_attention_with_dropout():
    """the form the reference actually calls (model/nlvr2.py:120-125,163-166): cross-attention (key = value != query) with a key
    padding mask AND attention dropout in training mode.  Masks are counter-based, so with the seed reset the forward repeats
    bit for bit: (1) the returned weights are the eval-mode weights, each either dropped or scaled by 1/(1-p), at a keep rate of
    1-p; (2) the output equals out_proj(dropped weights x projected values); (3) every input gradient matches central differences
    of the same masked function; (4) p = 0 in training mode equals eval mode."""
    from uc2_amd.model.attention import MultiheadAttention
    E, nh, L, N, S, pdrop = 64, 1, 40, 6, 33, 0.25
    m = MultiheadAttention(E, nh, dropout=pdrop)
    synth.det_init_(m)
    m.to(DEV)
    q, k = synth.det_normal((L, N, E), 90).to(DEV), synth.det_normal((S, N, E), 91).to(DEV)
    kpm = torch.zeros(N, S, dtype=torch.bool, device=DEV)
    kpm[1, 20:] = True
    m.eval()
    o_eval, w_eval = m(q, k, k, key_padding_mask=kpm)

    def train_fwd(qq, kk):
        ops.rng.manual_seed(4242, DEV)
        return m(qq, kk, kk, key_padding_mask=kpm)
    m.train()
    o1, w1 = train_fwd(q, k)
    o2, w2 = train_fwd(q, k)
    assert torch.equal(o1, o2) and torch.equal(w1, w2)
    kept = w1 > 0
    live = w_eval > 1e-12
    assert torch.allclose(w1[kept], w_eval[kept] / (1 - pdrop), rtol=1e-5, atol=1e-9)
    rate = float((kept & live).sum()) / float(live.sum())
    assert abs(rate - (1 - pdrop)) < 0.03, rate
    # (2) output from the dropped weights: out = out_proj(W_dropped @ (k W_v^T + b_v))
    Wv, bv = m.in_proj_weight[2 * E:], m.in_proj_bias[2 * E:]
    vproj = torch.einsum("sne,fe->snf", k, Wv) + bv                              # (S, N, E)
    ctx = torch.einsum("nls,snf->lnf", w1, vproj)
    want = torch.einsum("lnf,ef->lne", ctx, m.out_proj.weight) + m.out_proj.bias
    assert max_rel(o1.detach().cpu(), want.detach().cpu()) < 1e-4
    # (3) gradients against central differences with the masks held fixed
    wgt = synth.det_normal((L, N, E), 92).to(DEV)
    qg, kg = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    (train_fwd(qg, kg)[0] * wgt).sum().backward()
    f = lambda qq, kk: float((train_fwd(qq, kk)[0].detach().double() * wgt.double()).sum())
    eps = 1e-2
    for (t, g, idx) in ((q, qg.grad, (3, 1, 7)), (q, qg.grad, (39, 5, 60)), (k, kg.grad, (2, 0, 11)), (k, kg.grad, (30, 4, 5))):
        tp, tm = t.clone(), t.clone()
        tp[idx] += eps
        tm[idx] -= eps
        num = ((f(tp, k) - f(tm, k)) if t is q else (f(q, tp) - f(q, tm))) / (2 * eps)
        assert abs(num - float(g[idx])) < 2e-2 * max(1.0, abs(num)), (idx, num, float(g[idx]))
    # (4) p = 0 in training mode is the eval function
    m.dropout = 0.0
    o0, w0 = m(q, k, k, key_padding_mask=kpm)
    assert torch.equal(o0, o_eval) and torch.equal(w0, w_eval)


@pytest.mark.parametrize("task", ["itm", "mrfr", "mrc"])
def test_device_collate_vs_reference_collates(task):
    """f1: flat pinned buffers -> two kernels -> the padded batch, against the reference's own xlmr_*_collate outputs
    (tests/golden/golden_collate.npz), value for value; then through the prefetcher (side stream) and in bf16"""
    from uc2_amd.data.loader import DevicePrefetcher, assemble, ragged_collate
    g = golden("collate")
    samples = synth.sample_tuples(task, 6)
    rb = ragged_collate(task)(samples)
    b = assemble(rb, torch.device(DEV))
    torch.cuda.synchronize()
    keys = [k[len("collate/%s/" % task):] for k in g.files if k.startswith("collate/%s/" % task)]
    # (the reference's keys, plus the host-side masked-row count the device loader adds for the sync-free head)
    assert set(keys) == set(b.keys()) - {"n_txt_labels", "n_img_mask_tgt"}, (sorted(keys), sorted(b.keys()))
    for k in keys:
        ref = torch.from_numpy(g["collate/%s/%s" % (task, k)])
        got = b[k].cpu()
        assert tuple(got.shape) == tuple(ref.shape), (k, got.shape, ref.shape)
        assert torch.equal(got.to(ref.dtype) if got.dtype != ref.dtype else got, ref), k
    pf = DevicePrefetcher([ragged_collate(task)(samples), ragged_collate(task)(samples)], DEV, feat_dtype=torch.bfloat16)
    outs = list(pf)
    assert len(outs) == 2 and outs[0]["img_feat"].dtype == torch.bfloat16
    assert torch.equal(outs[1]["img_feat"].float().cpu(), torch.from_numpy(g["collate/%s/img_feat" % task]).to(torch.bfloat16).float())
    assert torch.equal(outs[0]["gather_index"], b["gather_index"])


def test_device_collate_mlm_feeds_the_model():
    """the MLM collate (data/mlm.py:761-801 is not importable offline: tokenizer download at import) against the same
    layout built on the host, then straight into the model"""
    from uc2_amd.data.loader import assemble, ragged_collate
    samples = synth.sample_tuples("mlm", 5, T=32, R=36, vocab_size=1000, img_dim=2048)
    b = assemble(ragged_collate("mlm")(samples), torch.device(DEV))
    tls = [s[0].numel() for s in samples]
    nbs = [s[1].shape[0] for s in samples]
    assert torch.equal(b["gather_index"].cpu(), O.get_gather_index(tls, nbs, 5, max(tls), max(a + c for a, c in zip(tls, nbs))))
    lab = torch.nn.utils.rnn.pad_sequence([s[4] for s in samples], batch_first=True, padding_value=-1)
    assert torch.equal(b["txt_labels"].cpu(), lab)
    model = build_pretrain(O.TINY, torch.float32)
    loss = model(b, "mlm", compute_loss=True)
    assert loss.numel() == int((lab != -1).sum()) and torch.isfinite(loss).all()
    # the loader hands the model the masked-token count it already has on the host (`n_txt_labels`): the model then compacts
    # the masked rows without a device -> host sync; same rows, same losses as the boolean-indexing route without the hint
    assert b["n_txt_labels"] == int((lab != -1).sum())
    unhinted = {k: v for k, v in b.items() if k != "n_txt_labels"}
    assert torch.equal(model(unhinted, "mlm", compute_loss=True), loss)
    samples = synth.sample_tuples("mrfr", 5, T=32, R=36, vocab_size=1000, img_dim=2048)
    b = assemble(ragged_collate("mrfr")(samples), torch.device(DEV))
    assert b["n_img_mask_tgt"] == int(b["img_mask_tgt"].sum())
    l1 = model(b, "mrfr", compute_loss=True)
    l2 = model({k: v for k, v in b.items() if k != "n_img_mask_tgt"}, "mrfr", compute_loss=True)
    assert torch.equal(l1, l2) and l1.shape[0] == b["n_img_mask_tgt"]


# ------------------------------------------------------------------------------------------ edge cases vs the oracle
def _oracle_weights(model):
    return OrderedDict((n, p.detach().float().cpu().clone()) for n, p in model.state_dict().items())


def _oracle_cfg(geom):
    return O.Config.make(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **geom)


def test_wrong_count_hint_is_memory_safe_and_counted():
    """`n_txt_labels` is the caller's promise about labels the host never reads back (sync-free row compaction).  A hint that is
    too LARGE must not index out of bounds: the surplus entries are -1, gathered as zero rows, skipped by the scatter in the
    backward, and their labels are ignore_index -- the real tokens' losses and every gradient equal the exact-hint run.  A hint
    that is too SMALL drops tokens (nothing the device can repair without a sync).  Both are counted in
    VLXLMRForPretraining.hint_mismatches() for whoever syncs anyway."""
    model = build_pretrain(O.TINY, torch.float32)
    batch = to_dev(synth.make_batch(1000, 6, 32, 36, task="mlm", seed=9))
    n = int((batch["txt_labels"] != -1).sum())
    base = type(model).hint_mismatches()

    def run(hint):
        b = dict(batch)
        b["n_txt_labels"] = hint
        model.zero_grad()
        loss = model(b, "mlm", compute_loss=True)
        loss.sum().backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), OrderedDict((k, p.grad.detach().clone()) for k, p in model.named_parameters() if p.grad is not None)
    l0, g0 = run(n)
    assert type(model).hint_mismatches() == base
    l1, g1 = run(n + 5)
    assert l1.shape[0] == n + 5 and torch.equal(l1[:n], l0) and float(l1[n:].abs().sum()) == 0.0
    for k in g0:
        # (fp32 atomics in the embedding / column-sum accumulations: the summation order varies from run to run)
        if k.endswith("key.bias"):
            continue                                    # mathematically zero (soft-max is shift invariant): rounding noise on both sides
        assert rel_err(g1[k], g0[k]) < 1e-5 or float(g0[k].norm()) < 1e-9, k
    assert type(model).hint_mismatches() == base + 1
    l2, _ = run(n - 2)
    assert l2.shape[0] == n - 2 and torch.equal(l2, l0[:n - 2])
    assert type(model).hint_mismatches() == base + 2


@pytest.mark.parametrize("task", ["mlm", "mrfr", "mrc"])
def test_nothing_masked_gives_empty_losses_like_the_reference(task):
    """a batch in which no token / region was selected for masking (possible with the reference's 15 % sampling on short
    inputs when its `at least one` guard is bypassed, and the shape every head must survive): the reference returns an
    empty loss tensor (boolean-mask indexing, model/model.py:583-596,600-625); so does the HIP path, and backward runs"""
    model = build_pretrain(O.TINY, torch.float32)
    batch = synth.make_batch(1000, 4, 16, 9, task=task, seed=3)
    if task == "mlm":
        batch["txt_labels"] = torch.full_like(batch["txt_labels"], -1)
    else:
        batch["img_mask_tgt"] = torch.zeros_like(batch["img_mask_tgt"])
        batch["img_masks"] = torch.zeros_like(batch["img_masks"])
        if task == "mrfr":
            batch["feat_targets"] = batch["feat_targets"][:0]
        else:
            batch["label_targets"] = batch["label_targets"][:0]
    ref = O.pretrain_forward(_oracle_weights(model), _oracle_cfg(O.TINY), strip(batch), task)
    b = to_dev(batch)
    model.zero_grad()
    loss = model(b, task, compute_loss=True)
    assert tuple(loss.shape) == tuple(ref.shape) and loss.numel() == 0
    (loss.sum() + 0.0 * sum(p.sum() for p in model.parameters() if p.requires_grad)).backward()
    for p in model.parameters():
        assert p.grad is None or float(p.grad.abs().sum()) == 0.0
    scores = model(b, task, compute_loss=False)
    assert scores.shape[0] == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_single_pair_batch_vs_oracle(dtype):
    """B = 1 (the last batch of an epoch): ITM scores, loss and a gradient against the oracle on the same weights"""
    model = build_pretrain(O.TINY, dtype)
    W = _oracle_weights(model)
    batch = synth.make_batch(1000, 1, 32, 36, task="itm", seed=5)
    _, scores, loss = run_task(model, batch, "itm")

    def loss_fn(Wg):
        return O.pretrain_forward(Wg, _oracle_cfg(O.TINY), strip(batch), "itm")[0].mean()
    ref_loss, grads = O.grads_of(loss_fn, W)
    ref_scores = O.pretrain_forward(W, _oracle_cfg(O.TINY), strip(batch), "itm", compute_loss=False)[0]
    f32 = dtype == torch.float32
    assert max_rel(scores.float().cpu(), ref_scores) < (TOL32 if f32 else 5e-2)
    assert abs(loss.mean().item() - float(ref_loss)) < (1e-5 if f32 else 2e-2)
    assert int(scores.argmax(-1)) == int(ref_scores.argmax(-1))
    name = "roberta.encoder.layer.1.output.dense.weight"
    assert rel_err(dict(model.named_parameters())[name].grad.float().cpu(), grads[name]) < (3e-3 if f32 else 0.3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sequence_longer_than_the_mfma_attention_tile_vs_oracle(dtype):
    """L = 100 tokens + 120 regions = 220 positions (max_bb 100 + max_txt_len 60 is the reference's normal ceiling; the
    fused MFMA attention kernels cover L <= 160): the general attention kernels take over, same results"""
    model = build_pretrain(O.TINY, dtype)
    W = _oracle_weights(model)
    batch = synth.make_batch(1000, 3, 100, 120, task="mlm", seed=6, variable_len=True)
    assert batch["attn_masks"].shape[1] > 160
    _, scores, loss = run_task(model, batch, "mlm")

    def loss_fn(Wg):
        return O.pretrain_forward(Wg, _oracle_cfg(O.TINY), strip(batch), "mlm").mean()
    ref_loss, grads = O.grads_of(loss_fn, W)
    ref_scores = O.pretrain_forward(W, _oracle_cfg(O.TINY), strip(batch), "mlm", compute_loss=False)
    f32 = dtype == torch.float32
    assert max_rel(scores.float().cpu(), ref_scores) < (TOL32 if f32 else 8e-2)
    assert abs(loss.mean().item() - float(ref_loss)) < (1e-4 if f32 else 5e-2) * abs(float(ref_loss))
    agree = float((scores.argmax(-1).cpu() == ref_scores.argmax(-1)).float().mean())
    assert agree == 1.0 if f32 else agree >= 0.9
    name = "roberta.encoder.layer.0.attention.self.value.weight"
    assert rel_err(dict(model.named_parameters())[name].grad.float().cpu(), grads[name]) < (3e-3 if f32 else 0.2)
