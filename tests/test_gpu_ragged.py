"""Ragged token counts (round 6): the reference's token-bucket batches have a new B x L every step (data/sampler.py:11-59).  In bf16
the encoder runs such a batch on B L rounded up to whole 256-row GEMM tiles (ops.padded_rows / PadRowsFn, VLXLMREncoder.forward) so
that every GEMM stays on its planned ping-pong kernel.  The padding must be invisible: same outputs, same gradients."""
import pytest
import torch

from oracle import uc2_oracle as O
from uc2_amd import ops
from uc2_amd.config import cfg as knobs
from uc2_amd.model.itm import VLXLMRForImageTextRetrieval
from uc2_amd.model.model import VLXLMRConfig, VLXLMRForPretraining
from uc2_amd.store import set_compute_dtype, store_of
from uc2_amd.utils import synth
from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(geom, drop):
    d = dict(hidden_act="gelu", hidden_dropout_prob=drop, attention_probs_dropout_prob=drop, max_position_embeddings=514,
             type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1)
    d.update(geom)
    return VLXLMRConfig.from_dict(d)


def _dev(b):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items() if not k.startswith("_")}


@pytest.mark.parametrize("task,drop", [("itm", 0.0), ("mlm", 0.0), ("itm", 0.1)])
def test_padded_rows_are_invisible(task, drop):
    """A variable-length batch whose B x L is NOT a multiple of 256 (56 pairs x 37..60 positions), 3 layers of base width, bf16:
    with the rows padded (default) against UC2_PAD_ROWS=0 -- the SAME dropout masks (a function of seed, site, row, column), scores,
    losses and every gradient equal to bf16 kernel-choice noise (the padded run takes the ping-pong kernels, the ragged one the
    ring / generic kernels: other summation orders), nothing non-finite; with all_encoded_layers the per-layer outputs come back
    as [B, L, H]; and the padded run leaves no GEMM off its plan."""
    geom = dict(O.BASE, num_hidden_layers=3, vocab_size=2000)
    batch = _dev(synth.make_batch(2000, 56, 40, 20, task=task, seed=31, variable_len=True))
    B, L = batch["attn_masks"].shape
    M = B * L
    assert M >= 1024 and M % 256 != 0, (B, L)
    res = {}
    was = knobs.pad_rows
    # ONE model for both runs (dropout sites are keyed by the layers' process-wide ids: two instances draw different masks)
    model = VLXLMRForPretraining(_cfg(geom, drop), img_dim=2048, img_label_dim=1601)
    synth.det_init_(model)
    model.to(DEV)
    set_compute_dtype(model, torch.bfloat16)
    try:
        for pad in (False, True):
            knobs.pad_rows = pad
            assert ops.padded_rows(M, torch.bfloat16) == ((M + 255) // 256 * 256 if pad else M)
            model.train()
            ops.rng.manual_seed(99, DEV)
            model.zero_grad()
            ops.gemm_fallbacks(reset=True)
            loss = model(batch, task, compute_loss=True)
            loss = loss[0] if isinstance(loss, tuple) else loss
            loss.mean().backward()
            ops.join_side_streams()
            torch.cuda.synchronize()
            fb = ops.gemm_fallbacks()
            grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
            with torch.no_grad():
                model.eval()
                seqs = model.roberta(batch["input_ids"], None, batch["img_feat"], batch["img_pos_feat"], batch["attn_masks"],
                                     batch["gather_index"], output_all_encoded_layers=True)
            assert len(seqs) == 3 and all(tuple(s_.shape) == (B, L, 768) for s_ in seqs)
            res[pad] = (loss.detach().float().clone(), grads, seqs[-1].float().clone(), fb)
    finally:
        knobs.pad_rows = was
    del model
    (l0, g0, s0, _), (l1, g1, s1, fb1) = res[False], res[True]
    assert fb1 == 0, fb1
    assert torch.isfinite(l1).all() and torch.isfinite(s1).all()
    assert rel_err(l1, l0) < 2e-3, rel_err(l1, l0)
    assert rel_err(s1, s0) < 1e-2, rel_err(s1, s0)
    assert set(g0) == set(g1)
    worst = ("", 0.0)
    for n in g0:
        assert torch.isfinite(g1[n]).all(), n
        if g0[n].norm() < 1e-7:
            assert g1[n].norm() < 1e-6, n
            continue
        e = rel_err(g1[n], g0[n])
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < (3e-2 if drop == 0.0 else 5e-2), (n, e)
    print("padded vs ragged rows, %s drop %.1f: %d x %d tokens, loss rel %.2e, last hidden rel %.2e, worst gradient %s %.2e"
          % (task, drop, B, L, rel_err(l1, l0), rel_err(s1, s0), worst[0], worst[1]))


def test_padded_rows_fp32_mode_is_untouched_and_retrieval_model_pads():
    """fp32 parity mode never pads (its kernels take any shape; the goldens stay bit-stable); the retrieval model
    (VLXLMRForImageTextRetrieval, the itm.py finetune shape: triplets, ragged num_bb) pads like the pretraining model and its
    triplet losses equal the unpadded run's."""
    assert ops.padded_rows(5000, torch.float32) == 5000 and ops.padded_rows(5000, torch.bfloat16) == 5120
    assert ops.padded_rows(512, torch.bfloat16) == 512 and ops.padded_rows(5120, torch.bfloat16) == 5120
    geom = dict(O.BASE, num_hidden_layers=2, vocab_size=2000)
    b = _dev(synth.make_batch(2000, 30, 40, 50, task="itm", seed=5, variable_len=True))
    b.pop("targets", None)
    b["sample_size"] = 3
    assert (b["attn_masks"].numel()) % 256 != 0
    out = {}
    was = knobs.pad_rows
    model = VLXLMRForImageTextRetrieval(_cfg(geom, 0.0), img_dim=2048)
    synth.det_init_(model)
    model.to(DEV).train()
    set_compute_dtype(model, torch.bfloat16)
    try:
        for pad in (False, True):
            knobs.pad_rows = pad
            model.zero_grad()
            loss = model(b, compute_loss=True)
            loss.mean().backward()
            ops.join_side_streams()
            torch.cuda.synchronize()
            g = dict(model.named_parameters())["roberta.encoder.layer.0.intermediate.dense.weight"].grad.float().clone()
            out[pad] = (loss.detach().float().clone(), g)
    finally:
        knobs.pad_rows = was
    del model
    assert (out[True][0] - out[False][0]).abs().max().item() < 5e-3
    # (the triplet gradient at initialisation is a difference of large cancelling terms; the padded run takes other GEMM kernels --
    #  other summation orders, other bf16 roundings: measured 8.4 %, the bf16 golden tests allow the ITM gradient 7 % against fp32)
    assert rel_err(out[True][1], out[False][1]) < 0.15 or out[False][1].norm() < 1e-7
