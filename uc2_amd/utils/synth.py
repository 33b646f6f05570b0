"""Deterministic synthetic weights and CC-shaped batches.

Everything here is integer-hash based (splitmix64 -> 24-bit mantissa), so the same
call gives bit-identical tensors in every process and on every machine; that is
what lets the parity tests, the golden-vector script and bench.py agree on inputs
without shipping any data.  Batch layout follows the reference's collates
(data/itm.py:205-232, data/mlm.py:761-801, data/mrm.py:73-119, data/data.py:360-384).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(idx, seed):
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def name_seed(name, seed=0):
    return (zlib.crc32(name.encode()) ^ (seed * 0x85EBCA6B)) & 0xFFFFFFFF


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 uniform in [lo, hi), exact 24-bit grid."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix(np.arange(n, dtype=np.uint64), seed)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def det_normal(shape, seed):
    """approximately N(0,1): sum of four uniforms (Irwin-Hall), exact arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        z = _splitmix(idx, seed * 4 + k + 0x51ED)
        acc += (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy(((acc - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape))


def det_randint(shape, seed, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix(np.arange(n, dtype=np.uint64), seed)
    return torch.from_numpy((lo + (z >> np.uint64(11)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape))


def det_bernoulli(shape, seed, p):
    return det_uniform(shape, seed, 0.0, 1.0) < p


def slice_idx(n, k=64):
    """the k (or n) evenly spread flat indices the golden fixtures sample (exact integer math)."""
    k = min(k, n)
    if k <= 1:
        return torch.zeros(k, dtype=torch.long)
    return (torch.arange(k, dtype=torch.long) * (n - 1)) // (k - 1)


def det_fill_(name, tensor, seed=0, std=0.02):
    """Fill one parameter in place from its state_dict name.
    weights ~ U(+-std*sqrt3) (variance std^2, model/model.py:159-172 uses N(0, 0.02));
    LayerNorm gains 1 +- 0.1, every bias +- 0.02 so that no term is trivially zero."""
    s = name_seed(name, seed)
    shape = tuple(tensor.shape)
    low = name.lower()
    is_ln = ("layernorm" in low) or ("layer_norm" in low) or low.endswith("net.2.weight") or low.endswith("net.2.bias")
    if is_ln and name.endswith("weight"):
        v = 1.0 + det_uniform(shape, s, -0.1, 0.1)
    elif name.endswith("bias"):
        v = det_uniform(shape, s, -0.02, 0.02)
    else:
        a = std * (3.0 ** 0.5)
        v = det_uniform(shape, s, -a, a)
    with torch.no_grad():
        tensor.copy_(v.to(tensor.dtype))
    return tensor


def det_init_(module_or_state, seed=0):
    """Fill every floating-point entry of a module's named_parameters (or a dict)."""
    items = module_or_state.named_parameters() if hasattr(module_or_state, "named_parameters") \
        else module_or_state.items()
    for n, p in items:
        if p.is_floating_point():
            det_fill_(n, p.data if hasattr(p, "data") else p, seed)
    return module_or_state


# --------------------------------------------------------------------------- #
# batches
# --------------------------------------------------------------------------- #
def _gather_index(txt_lens, num_bbs, bs, max_len, out_size):
    gi = torch.arange(0, out_size, dtype=torch.long).unsqueeze(0).repeat(bs, 1)
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        gi[i, tl:tl + nbb] = torch.arange(max_len, max_len + nbb, dtype=torch.long)
    return gi


def tlm_position_ids(row_ids):
    """data/mlm.py:420-429: positions count up from 3 and restart at 2 on every <s> (id 0), so the second
    caption of a TLM pair gets its own position range"""
    out, pos = [], 2
    for t in row_ids:
        pos = 2 if t == 0 else pos + 1
        out.append(pos)
    return out


def ot_inputs_for(txt_lens, num_bbs, max_tl, max_bb, joint_len):
    """data/itm.py:264-278,303-309: scatter index back to the padded [txt | img] layout and the pad masks
    (bool here; the reference builds uint8 masks, which torch >= 1.2 treats as bool)"""
    sc = torch.arange(0, joint_len, dtype=torch.long).unsqueeze(0).repeat(len(txt_lens), 1)
    for i, tl in enumerate(txt_lens):
        sc[i, tl:] = torch.arange(max_tl, max_tl + (joint_len - tl), dtype=torch.long)
    tp = torch.zeros(len(txt_lens), max_tl, dtype=torch.bool)
    ip = torch.zeros(len(num_bbs), max_bb, dtype=torch.bool)
    for i, (tl, nb) in enumerate(zip(txt_lens, num_bbs)):
        tp[i, tl:] = True
        ip[i, nb:] = True
    return {"ot_scatter": sc, "scatter_max": int(sc.max().item()), "txt_pad": tp, "img_pad": ip}


def make_batch(vocab_size, B, T, R, task="itm", seed=1, img_dim=2048, img_label_dim=1601,
               variable_len=False, sample_size=None, n_soft=45, ot=False):
    """One synthetic batch with the reference's dict keys.

    Fixed length (default): every pair has T tokens and R regions, attn mask all
    ones, gather_index = arange(T+R).  variable_len: per-pair txt_len in [T//2, T],
    num_bb in [R//2, R]; padding, masks and gather_index built exactly as
    xlmr_itm_collate + get_gather_index do.
    """
    s = seed * 1000003
    if variable_len:
        tls = det_randint((B,), s + 1, max(3, T // 2), T + 1).tolist()
        nbs = det_randint((B,), s + 2, max(1, R // 2), R + 1).tolist()
        tls[0], nbs[0] = T, R                       # keep the padded extents
    else:
        tls, nbs = [T] * B, [R] * B
    max_tl, max_bb = max(tls), max(nbs)
    ids = det_randint((B, max_tl), s + 3, 5, vocab_size)
    input_ids = torch.ones(B, max_tl, dtype=torch.long)              # pad id = 1
    for i, tl in enumerate(tls):
        input_ids[i, :tl] = ids[i, :tl]
        input_ids[i, 0] = 0                                          # <s>
        input_ids[i, tl - 1] = 2                                     # </s>
    position_ids = torch.arange(0, max_tl, dtype=torch.long).unsqueeze(0)
    feat = det_normal((B, max_bb, img_dim), s + 4)
    pos = det_uniform((B, max_bb, 7), s + 5, 0.0, 1.0)
    pos[..., 6] = pos[..., 4] * pos[..., 5]                          # data/data.py:339
    for i, nb in enumerate(nbs):
        feat[i, nb:] = 0
        pos[i, nb:] = 0
    out_size = max(tl + nb for tl, nb in zip(tls, nbs))
    attn = torch.zeros(B, out_size, dtype=torch.long)
    for i, (tl, nb) in enumerate(zip(tls, nbs)):
        attn[i, :tl + nb] = 1
    batch = OrderedDict(input_ids=input_ids, position_ids=position_ids, img_feat=feat,
                        img_pos_feat=pos, attn_masks=attn,
                        gather_index=_gather_index(tls, nbs, B, max_tl, out_size))
    batch["_txt_lens"], batch["_num_bbs"] = tls, nbs

    if task in ("tlm", "tlm-ni"):
        # two captions per sample: a second <s> in the middle (data/mlm.py TLM datasets), batch position_ids
        for i, tl in enumerate(tls):
            input_ids[i, max(2, tl // 2)] = 0
        pid = torch.ones(B, max_tl, dtype=torch.long)                # pad_sequence(..., padding_value=1), data/mlm.py:822
        for i, tl in enumerate(tls):
            pid[i, :tl] = torch.tensor(tlm_position_ids(input_ids[i, :tl].tolist()))
        batch["position_ids"] = pid
    if task == "tlm-ni":                                             # data/mlm.py:803-843: text only, no gather
        a = torch.zeros(B, max_tl, dtype=torch.long)
        for i, tl in enumerate(tls):
            a[i, :tl] = 1
        batch["attn_masks"] = a
        batch["gather_index"] = None
    if task in ("mlm", "tlm", "tlm-ni", "vmlm", "mmxlm"):
        input_ids = batch["input_ids"]
        lab = torch.full((B, max_tl), -1, dtype=torch.long)
        pick = det_bernoulli((B, max_tl), s + 6, 0.15)
        for i, tl in enumerate(tls):
            pick[i, 0] = False
            pick[i, tl - 1:] = False
            if not pick[i].any():
                pick[i, 1 + (i % max(1, tl - 2))] = True              # at least one (data/mlm.py:59-62)
        lab[pick] = input_ids[pick]
        masked_ids = input_ids.clone()
        masked_ids[pick] = 250001 if vocab_size > 250001 else vocab_size - 1   # <mask>
        batch["input_ids"] = masked_ids
        if task in ("vmlm", "mmxlm"):
            # labels over the whole (txt+img) sequence; masked regions carry a token id
            full = torch.full((B, out_size), -1, dtype=torch.long)
            full[:, :max_tl] = lab
            im = _img_masks(B, max_bb, nbs, s + 7)
            tok = det_randint((B, max_bb), s + 8, 5, vocab_size)
            for i, (tl, nb) in enumerate(zip(tls, nbs)):
                row = torch.full((max_bb,), -1, dtype=torch.long)
                row[im[i]] = tok[i][im[i]]
                full[i, tl:tl + nb] = row[:nb]
            batch["txt_labels"] = full
            batch["img_masks"] = im
            batch["img_feat"] = feat.masked_fill(im.unsqueeze(-1), 0)      # data/mrm.py:36-39
        else:
            batch["txt_labels"] = lab
    if task == "itm":
        batch["targets"] = det_bernoulli((B,), s + 9, 0.5).long()
        if ot:
            batch["targets"][0], batch["targets"][-1] = 1, 0        # both OT branches non-empty
            batch["ot_inputs"] = ot_inputs_for(tls, nbs, max_tl, max_bb, out_size)
    if task in ("vmlm-soft", "mmxlm-soft"):
        # model/model.py:627-651: masked regions predict a soft distribution over the VALID token subset
        im = _img_masks(B, max_bb, nbs, s + 7)
        batch["img_masks"] = im
        tgt = torch.zeros(B, out_size, dtype=torch.bool)
        for i, (tl, nb) in enumerate(zip(tls, nbs)):
            tgt[i, tl:tl + nb] = im[i, :nb]
        batch["tgt_masks"] = tgt
        soft = det_uniform((B, max_bb, n_soft), s + 11, 0.0, 1.0) ** 4
        soft = soft / soft.sum(-1, keepdim=True)
        batch["label_targets"] = soft[im].contiguous()
        batch["img_feat"] = feat.masked_fill(im.unsqueeze(-1), 0)
    if task in ("mrfr",) or task.startswith("mrc"):
        im = _img_masks(B, max_bb, nbs, s + 7)
        batch["img_masks"] = im
        tgt = torch.zeros(B, out_size, dtype=torch.bool)
        for i, (tl, nb) in enumerate(zip(tls, nbs)):
            tgt[i, tl:tl + nb] = im[i, :nb]
        batch["img_mask_tgt"] = tgt
        if task == "mrfr":
            batch["feat_targets"] = feat[im].contiguous()
        else:
            soft = det_uniform((B, max_bb, img_label_dim), s + 10, 0.0, 1.0) ** 8
            soft = soft / soft.sum(-1, keepdim=True)
            batch["label_targets"] = soft[im].contiguous()
        batch["img_feat"] = feat.masked_fill(im.unsqueeze(-1), 0)
    if sample_size is not None:
        batch["sample_size"] = sample_size
    return batch


def _img_masks(B, max_bb, nbs, seed):
    im = det_bernoulli((B, max_bb), seed, 0.15)
    for i, nb in enumerate(nbs):
        im[i, nb:] = False
        if not im[i].any():
            im[i, i % nb] = True                                       # data/mrm.py:13-19
    return im


def batch_to(batch, device, float_dtype=None):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            v = v.to(device)
            if float_dtype is not None and v.is_floating_point():
                v = v.to(float_dtype)
        out[k] = v
    return out


def retrieval_pool(vocab_size, n_txt, n_img, T, R, seed=7, img_dim=2048):
    """texts and images of a synthetic retrieval set (variable lengths): ids [n_txt, T] (pad 1), txt_lens,
    img_feat [n_img, R, img_dim], img_pos_feat [n_img, R, 7], num_bbs"""
    tb = make_batch(vocab_size, n_txt, T, R, task="itm", seed=seed, variable_len=True, img_dim=img_dim)
    ib = make_batch(vocab_size, n_img, T, R, task="itm", seed=seed + 1, variable_len=True, img_dim=img_dim)
    return dict(input_ids=tb["input_ids"], txt_lens=tb["_txt_lens"], img_feat=ib["img_feat"], img_pos_feat=ib["img_pos_feat"],
                num_bbs=ib["_num_bbs"])


def retrieval_batch(pool, i, j0, j1):
    """text i against images j0..j1-1, built like itm_eval_collate (data/itm.py:905-930): the text repeated, images
    padded to the longest of the mini-batch, attention mask and gather index from the true lengths"""
    n = j1 - j0
    tl = pool["txt_lens"][i]
    nbs = pool["num_bbs"][j0:j1]
    max_bb = max(nbs)
    ids = pool["input_ids"][i:i + 1, :tl].repeat(n, 1)
    feat = pool["img_feat"][j0:j1, :max_bb].clone()
    pos = pool["img_pos_feat"][j0:j1, :max_bb].clone()
    out_size = tl + max_bb
    attn = torch.zeros(n, out_size, dtype=torch.long)
    for r, nb in enumerate(nbs):
        attn[r, :tl + nb] = 1
    return OrderedDict(input_ids=ids, position_ids=torch.arange(0, tl, dtype=torch.long).unsqueeze(0), img_feat=feat,
                       img_pos_feat=pos, attn_masks=attn, gather_index=_gather_index([tl] * n, nbs, n, tl, out_size))


def sample_tuples(task, B, T=12, R=9, vocab_size=1000, img_dim=64, img_label_dim=16, seed=5):
    """per-sample tuples shaped like the reference datasets' __getitem__ output (ragged: text length in [T/2, T],
    regions in [R/2, R]), for the collate tests (data/itm.py:186-202, data/mrm.py:54-71,233-250, data/mlm.py:375-394)"""
    s = seed * 7919
    tls = det_randint((B,), s + 1, max(3, T // 2), T + 1).tolist()
    nbs = det_randint((B,), s + 2, max(2, R // 2), R + 1).tolist()
    out = []
    for i, (tl, nb) in enumerate(zip(tls, nbs)):
        ids = det_randint((tl,), s + 10 + i, 5, vocab_size)
        ids[0], ids[-1] = 0, 2
        feat = det_normal((nb, img_dim), s + 100 + i)
        pos = det_uniform((nb, 7), s + 200 + i, 0.0, 1.0)
        attn = torch.ones(tl + nb, dtype=torch.long)
        if task == "itm":
            out.append((ids, feat, pos, attn, torch.tensor([i % 2], dtype=torch.long)))
        elif task == "mlm":
            lab = torch.full((tl,), -1, dtype=torch.long)
            lab[1 + i % max(1, tl - 2)] = ids[1 + i % max(1, tl - 2)]
            out.append((ids, feat, pos, attn, lab))
        else:
            m = det_bernoulli((nb,), s + 300 + i, 0.3)
            if not m.any():
                m[i % nb] = True
            tgt = torch.cat([torch.zeros(tl, dtype=torch.bool), m])
            if task == "mrfr":
                out.append((ids, feat, pos, attn, m, tgt))
            else:
                soft = det_uniform((nb, img_label_dim), s + 400 + i, 0.0, 1.0)
                out.append((ids, feat, pos, soft / soft.sum(-1, keepdim=True), attn, m, tgt))
    return out
