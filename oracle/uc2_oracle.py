"""CPU restatement (plain torch fp32 ops) of the UC2 encoder hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  This is the checker the HIP
path is compared with; it is never the thing shipped or measured (except as the
``cpu_baseline`` leg of bench.py, kind="port").

Pinned against the reference itself: ``tests/golden/make_golden.py`` imports the
reference's own modules from /root/reference (three inert shims, SURVEY.md
Appendix B), runs them on closed-form inputs and commits the outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors.  The reference has no tests or golden vectors of its own
(SURVEY.md F2).  Third-party arithmetic that is not vendored in the reference
(apex FusedLayerNorm, Horovod's averaging all-reduce) is restated from its
published behaviour (= torch.nn.LayerNorm, arithmetic mean): parity for those
two boundaries is *unpinned* (SURVEY.md §8c).

All functions are pure: weights come in a dict keyed by the reference's
``state_dict`` names.  Every function cites the reference lines it follows
(paths relative to /root/reference).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# configuration
# --------------------------------------------------------------------------- #
class Config(dict):
    """Same field names as the reference's VLXLMRConfig (model/model.py:45-141)."""
    __getattr__ = dict.__getitem__

    @staticmethod
    def make(vocab_size=250002, hidden_size=768, num_hidden_layers=12,
             num_attention_heads=12, intermediate_size=3072,
             hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
             max_position_embeddings=514, type_vocab_size=2,
             initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1,
             hidden_act="gelu"):
        return Config(locals())


BASE = dict(vocab_size=250002, hidden_size=768, num_hidden_layers=12,
            num_attention_heads=12, intermediate_size=3072)       # config/uc2-base.json
LARGE = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=24,
             num_attention_heads=16, intermediate_size=4096)      # BASELINE.json configs[4] (XLM-R-large sizes; no reference config file)
TINY = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2,
            num_attention_heads=4, intermediate_size=512)          # BASELINE.json configs[0]


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def gelu_erf(x):
    """model/layer.py:31-37 -- exact erf form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, gamma, beta, eps):
    """apex FusedLayerNorm == torch.nn.LayerNorm semantics (model/layer.py:25):
    biased variance over the last dim, affine."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def dropout(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0.0) else x


# --------------------------------------------------------------------------- #
# encoder
# --------------------------------------------------------------------------- #
def self_attention(x, ext_mask, W, p, nheads, drop_p=0.0, training=False):
    """BertSelfAttention.forward, model/layer.py:75-101."""
    B, L, H = x.shape
    d = H // nheads
    q = linear(x, W[p + "query.weight"], W[p + "query.bias"])
    k = linear(x, W[p + "key.weight"], W[p + "key.bias"])
    v = linear(x, W[p + "value.weight"], W[p + "value.bias"])

    def heads(t):
        return t.view(B, L, nheads, d).permute(0, 2, 1, 3)
    q, k, v = heads(q), heads(k), heads(v)
    s = q @ k.transpose(-1, -2)
    s = s / math.sqrt(d)            # scale after the product (:86)
    s = s + ext_mask                # additive [B,1,1,L] (:88)
    pr = torch.softmax(s, dim=-1)
    pr = dropout(pr, drop_p, training)
    c = pr @ v
    return c.permute(0, 2, 1, 3).contiguous().view(B, L, H)


def bert_layer(x, ext_mask, W, p, nheads, cfg=None, training=False):
    """BertLayer.forward, model/layer.py:159-170 (post-LN block, eps 1e-12)."""
    hp = cfg.hidden_dropout_prob if cfg is not None else 0.0
    ap = cfg.attention_probs_dropout_prob if cfg is not None else 0.0
    ctx = self_attention(x, ext_mask, W, p + "attention.self.", nheads, ap, training)
    o1 = linear(ctx, W[p + "attention.output.dense.weight"], W[p + "attention.output.dense.bias"])
    o1 = dropout(o1, hp, training)
    a = layer_norm(o1 + x, W[p + "attention.output.LayerNorm.weight"],
                   W[p + "attention.output.LayerNorm.bias"], 1e-12)      # :111-115
    u = gelu_erf(linear(a, W[p + "intermediate.dense.weight"], W[p + "intermediate.dense.bias"]))  # :139-142
    o2 = linear(u, W[p + "output.dense.weight"], W[p + "output.dense.bias"])
    o2 = dropout(o2, hp, training)
    return layer_norm(o2 + a, W[p + "output.LayerNorm.weight"],
                      W[p + "output.LayerNorm.bias"], 1e-12)             # :152-156


def encoder(x, ext_mask, W, cfg, prefix="roberta.encoder.", training=False):
    """VLXLMREncoder.forward, model/model.py:373-383 -> list of every layer's output."""
    outs = []
    for i in range(cfg.num_hidden_layers):
        x = bert_layer(x, ext_mask, W, "%slayer.%d." % (prefix, i),
                       cfg.num_attention_heads, cfg, training)
        outs.append(x)
    return outs


def extended_mask(attention_mask, dtype=torch.float32):
    """model/model.py:433-436: (1 - m) * -10000, shape [B,1,1,L]."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * -10000.0


# --------------------------------------------------------------------------- #
# embeddings
# --------------------------------------------------------------------------- #
def position_ids_from_input_ids(input_ids, padding_idx=1):
    """create_position_ids_from_input_ids, model/model.py:280-290."""
    mask = input_ids.ne(padding_idx).int()
    inc = torch.cumsum(mask, dim=1).type_as(mask) * mask
    return inc.long() + padding_idx


def text_embeddings(input_ids, position_ids, W, cfg, token_type_ids=None,
                    p="roberta.embeddings.", training=False):
    """VLXLMRTextEmbeddings.forward, model/model.py:304-335."""
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    if position_ids is None:
        position_ids = position_ids_from_input_ids(input_ids, cfg.pad_token_id)
    # nn.Embedding(..., padding_idx=pad_token_id) for words AND positions (model/model.py:296-298): the
    # padding row is read like any other but receives no gradient
    e = (F.embedding(input_ids, W[p + "word_embeddings.weight"], padding_idx=cfg.pad_token_id)
         + F.embedding(position_ids, W[p + "position_embeddings.weight"], padding_idx=cfg.pad_token_id)
         + W[p + "new_token_type_embeddings.weight"][token_type_ids])
    e = layer_norm(e, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], cfg.layer_norm_eps)
    return dropout(e, cfg.hidden_dropout_prob, training)


def image_embeddings(img_feat, img_pos_feat, W, cfg, img_masks=None, img_type_ids=None,
                     p="roberta.img_embeddings.", training=False):
    """VLXLMRModel._compute_img_embeddings (model/model.py:401-410) +
    VLXLMRImageEmbeddings.forward (:352-364)."""
    if img_type_ids is None:
        img_type_ids = torch.ones_like(img_feat[:, :, 0].long())
    type_emb = W["roberta.embeddings.new_token_type_embeddings.weight"][img_type_ids]
    if img_masks is not None:
        me = W[p + "mask_embedding.weight"]
        me.data[0, :].fill_(0)          # re-zeroed IN PLACE every call (:354): the parameter itself changes
        me = torch.cat([torch.zeros_like(me[:1]), me[1:]], 0)     # padding_idx=0: row 0 gets no gradient (:347)
        img_feat = img_feat + me[img_masks.long()]
    ti = layer_norm(linear(img_feat, W[p + "img_linear.weight"], W[p + "img_linear.bias"]),
                    W[p + "img_layer_norm.weight"], W[p + "img_layer_norm.bias"], cfg.layer_norm_eps)
    tp = layer_norm(linear(img_pos_feat, W[p + "pos_linear.weight"], W[p + "pos_linear.bias"]),
                    W[p + "pos_layer_norm.weight"], W[p + "pos_layer_norm.bias"], cfg.layer_norm_eps)
    e = layer_norm(ti + tp + type_emb, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"],
                   cfg.layer_norm_eps)
    return dropout(e, cfg.hidden_dropout_prob, training)


def model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                  gather_index=None, img_masks=None, output_all_encoded_layers=False,
                  txt_type_ids=None, img_type_ids=None, training=False):
    """VLXLMRModel.forward, model/model.py:427-458."""
    ext = extended_mask(attention_mask)
    if input_ids is None:
        emb = image_embeddings(img_feat, img_pos_feat, W, cfg, img_masks, img_type_ids, training=training)
    elif img_feat is None:
        emb = text_embeddings(input_ids, position_ids, W, cfg, txt_type_ids, training=training)
    else:
        te = text_embeddings(input_ids, position_ids, W, cfg, txt_type_ids, training=training)
        ie = image_embeddings(img_feat, img_pos_feat, W, cfg, img_masks, img_type_ids, training=training)
        gi = gather_index.unsqueeze(-1).expand(-1, -1, cfg.hidden_size)
        emb = torch.gather(torch.cat([te, ie], dim=1), dim=1, index=gi)     # :412-425
    outs = encoder(emb, ext, W, cfg, training=training)
    return outs if output_all_encoded_layers else outs[-1]


# --------------------------------------------------------------------------- #
# heads
# --------------------------------------------------------------------------- #
def pooler(seq, W, p="roberta.pooler."):
    """BertPooler.forward, model/layer.py:179-185."""
    return torch.tanh(linear(seq[:, 0], W[p + "dense.weight"], W[p + "dense.bias"]))


def lm_head(x, W, cfg, p="cls."):
    """RobertaLMHead.forward, model/layer.py:257-265 (decoder tied to the word embedding)."""
    z = layer_norm(gelu_erf(linear(x, W[p + "dense.weight"], W[p + "dense.bias"])),
                   W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], cfg.layer_norm_eps)
    return linear(z, W["roberta.embeddings.word_embeddings.weight"], W[p + "bias"])


def feat_regress(x, W, p="feat_regress."):
    """RegionFeatureRegression.forward, model/model.py:1143-1156 (weight tied to img_linear)."""
    h = layer_norm(gelu_erf(linear(x, W[p + "net.0.weight"], W[p + "net.0.bias"])),
                   W[p + "net.2.weight"], W[p + "net.2.bias"], 1e-12)
    return linear(h, W["roberta.img_embeddings.img_linear.weight"].t(), W[p + "bias"])


def region_classifier(x, W, p="region_classifier."):
    """RegionClassification.forward, model/model.py:1159-1169."""
    h = layer_norm(gelu_erf(linear(x, W[p + "net.0.weight"], W[p + "net.0.bias"])),
                   W[p + "net.2.weight"], W[p + "net.2.bias"], 1e-12)
    return linear(h, W[p + "net.3.weight"], W[p + "net.3.bias"])


def masked_hidden(hidden, mask):
    """_compute_masked_hidden, model/model.py:653-657."""
    return hidden[mask.unsqueeze(-1).expand_as(hidden)].contiguous().view(-1, hidden.size(-1))


# --------------------------------------------------------------------------- #
# optimal-transport regulariser of the ITM head (model/ot.py)
# --------------------------------------------------------------------------- #
def cost_matrix_cosine(x, y, eps=1e-5):
    """model/ot.py:8-19: 1 - cosine similarity of every (x_i, y_j) pair, [B,Lx,D],[B,Ly,D] -> [B,Lx,Ly]."""
    xn = F.normalize(x, p=2, dim=-1, eps=eps)
    yn = F.normalize(y, p=2, dim=-1, eps=eps)
    return 1 - xn.matmul(yn.transpose(1, 2))


@torch.no_grad()
def ipot(C, x_len, x_pad, y_len, y_pad, joint_pad, beta, iteration, k):
    """model/ot.py:32-63: inexact proximal point iterations for the transport plan T [B,N,M]."""
    b, m, n = C.size()
    sigma = torch.ones(b, m, dtype=C.dtype) / x_len.unsqueeze(1)
    T = torch.ones(b, n, m, dtype=C.dtype)
    A = torch.exp(-C.transpose(1, 2) / beta)
    sigma = sigma.masked_fill(x_pad, 0)
    jp = joint_pad.transpose(1, 2)
    T = T.masked_fill(jp, 0)
    A = A.masked_fill(jp, 0)
    x_len = x_len.unsqueeze(1).unsqueeze(2)
    y_len = y_len.unsqueeze(1).unsqueeze(2)
    x_mask = (x_pad.to(C.dtype) * 1e4).unsqueeze(1)
    y_mask = (y_pad.to(C.dtype) * 1e4).unsqueeze(1)
    for _ in range(iteration):
        Q = A * T
        sigma = sigma.view(b, m, 1)
        for _ in range(k):
            delta = 1 / (y_len * Q.matmul(sigma).view(b, 1, n) + y_mask)
            sigma = 1 / (x_len * delta.matmul(Q) + x_mask)
        T = delta.view(b, n, 1) * Q * sigma
    return T.masked_fill(jp, 0)


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1):
    """model/ot.py:66-82: trace(cost @ T) with T from IPOT on the detached cost."""
    cost = cost_matrix_cosine(txt_emb, img_emb)
    joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
    cost = cost.masked_fill(joint_pad, 0)
    txt_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(cost.dtype)
    img_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(cost.dtype)
    T = ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, beta, iteration, k)
    return torch.diagonal(cost.matmul(T.detach()), dim1=1, dim2=2).sum(-1)


def itm_ot_loss(seq, input_ids, img_feat, targets, ot_inputs, ot_pos_only=False):
    """model/model.py:701-729: scatter the compact sequence back to [txt | img], OT distance per pair,
    split by the ITM target."""
    b, _, H = seq.shape
    tl, il = input_ids.size(1), img_feat.size(1)
    max_l = max(ot_inputs["scatter_max"] + 1, tl + il)
    sc = ot_inputs["ot_scatter"].unsqueeze(-1).expand_as(seq)
    ctx = torch.zeros(b, max_l, H, dtype=seq.dtype).scatter(1, sc, seq)
    dist = optimal_transport_dist(ctx[:, :tl], ctx[:, tl:tl + il], ot_inputs["txt_pad"].bool(), ot_inputs["img_pad"].bool())
    if ot_pos_only:
        return dist.masked_select(targets == 1)
    return dist.masked_select(targets == 1), dist.masked_select(targets == 0)


def multi_head_attention(query, key, value, W, nheads, key_padding_mask=None, attn_mask=None):
    """MultiheadAttention.forward / multi_head_attention_forward (model/attention.py:12-264) for the packed
    in_proj case: (L,N,E) layout, q scaled by d^-1/2 BEFORE the product (:139), additive float attn_mask
    [L,S] (:186-190), boolean key_padding_mask -> -inf (:222-229), dropout 0.  Returns (out (L,N,E),
    head-averaged weights (N,L,S))."""
    L, N, E = query.shape
    S = key.shape[0]
    d = E // nheads
    Wi, bi = W["in_proj_weight"], W.get("in_proj_bias")
    b3 = (lambda i: None if bi is None else bi[i * E:(i + 1) * E])
    q = linear(query, Wi[:E], b3(0)) * (float(d) ** -0.5)
    k = linear(key, Wi[E:2 * E], b3(1))
    v = linear(value, Wi[2 * E:], b3(2))
    q = q.contiguous().view(L, N * nheads, d).transpose(0, 1)
    k = k.contiguous().view(S, N * nheads, d).transpose(0, 1)
    v = v.contiguous().view(S, N * nheads, d).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        w = w + attn_mask.unsqueeze(0)
    if key_padding_mask is not None:
        w = w.view(N, nheads, L, S).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(N * nheads, L, S)
    w = torch.softmax(w, dim=-1)
    o = torch.bmm(w, v).transpose(0, 1).contiguous().view(L, N, E)
    o = linear(o, W["out_proj.weight"], W.get("out_proj.bias"))
    return o, w.view(N, nheads, L, S).sum(dim=1) / nheads


VALID_XLMR_TOKEN_IDS = list(range(5, 50))     # stand-in for model/const_variable.py (tokenizer download), as in the fixtures


def pretrain_forward(W, cfg, batch, task, compute_loss=True, training=False, ot_pos_only=False):
    """VLXLMRForPretraining.forward, model/model.py:495-775.  Returns what the
    reference returns: unreduced losses, or raw scores when compute_loss=False."""
    g = lambda k: batch.get(k, None)
    input_ids = g("input_ids")
    position_ids = g("position_ids") if task == "tlm" else None
    img_feat, img_pos_feat = g("img_feat"), g("img_pos_feat")
    am, gi = g("attn_masks"), g("gather_index")
    if task in ("mlm", "tlm", "tlm-ni"):
        if task == "tlm-ni":
            img_feat = img_pos_feat = gi = None
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi, training=training)
        seq = seq[:, :input_ids.size(1), :]
        labels = g("txt_labels")
        scores = lm_head(masked_hidden(seq, labels != -1), W, cfg)
        if compute_loss:
            return F.cross_entropy(scores, labels[labels != -1], reduction="none")
        return scores
    if task in ("mmxlm", "vmlm"):
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi,
                            img_masks=g("img_masks"), training=training)
        labels = g("txt_labels")
        scores = lm_head(masked_hidden(seq, labels != -1), W, cfg)
        if compute_loss:
            return F.cross_entropy(scores, labels[labels != -1], reduction="none")
        return scores
    if task == "mrfr":
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi,
                            img_masks=g("img_masks"), training=training)
        pred = feat_regress(masked_hidden(seq, g("img_mask_tgt")), W)
        if compute_loss:
            return F.mse_loss(pred, g("feat_targets"), reduction="none")
        return pred
    if task in ("mmxlm-soft", "vmlm-soft"):
        # forward_mmxlm_soft, model/model.py:627-651
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi,
                            img_masks=g("img_masks"), training=training)
        scores = lm_head(masked_hidden(seq, g("tgt_masks")), W, cfg)[:, VALID_XLMR_TOKEN_IDS]
        if compute_loss:
            return F.kl_div(F.log_softmax(scores, dim=-1), g("label_targets"), reduction="none")
        return scores
    if task == "itm":
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi, training=training)
        scores = linear(pooler(seq, W), W["itm_output.weight"], W["itm_output.bias"])
        ot_loss = None
        if g("ot_inputs") is not None:
            ot_loss = itm_ot_loss(seq, input_ids, img_feat, g("targets"), g("ot_inputs"), ot_pos_only)
        if compute_loss:
            return F.cross_entropy(scores, g("targets"), reduction="none"), ot_loss
        return scores, ot_loss
    if task.startswith("mrc"):
        seq = model_forward(W, cfg, input_ids, position_ids, img_feat, img_pos_feat, am, gi,
                            img_masks=g("img_masks"), training=training)
        pred = region_classifier(masked_hidden(seq, g("img_mask_tgt")), W)
        if compute_loss:
            lt = g("label_targets")
            if "kl" in task:
                return F.kl_div(F.log_softmax(pred, dim=-1), lt, reduction="none")
            lt = torch.max(lt[:, 1:], dim=-1)[1] + 1
            return F.cross_entropy(pred, lt, ignore_index=0, reduction="none")
        return pred
    raise ValueError("invalid task")


def itm_rank_forward(W, cfg, batch, margin=0.2, compute_loss=True, training=False):
    """VLXLMRForImageTextRetrieval.forward, model/itm.py:28-55."""
    seq = model_forward(W, cfg, batch["input_ids"], None, batch["img_feat"], batch["img_pos_feat"],
                        batch["attn_masks"], batch["gather_index"], training=training)
    scores = linear(pooler(seq, W), W["rank_output.weight"], W["rank_output.bias"])
    if not compute_loss:
        return scores
    s = torch.sigmoid(scores).contiguous().view(-1, batch["sample_size"])
    pos, neg = s[:, :1], s[:, 1:]
    return torch.clamp(margin + neg - pos, 0)


# --------------------------------------------------------------------------- #
# optimizer / schedule / data-parallel averaging
# --------------------------------------------------------------------------- #
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")       # optim/misc.py:11


def is_no_decay(name):
    """optim/misc.py:12-19: case-sensitive substring match."""
    return any(nd in name for nd in NO_DECAY)


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, correct_bias=True):
    """One AdamW.step for one tensor, optim/adamw.py:50-101.  `step` is the
    1-based count *after* the increment at :74.  Updates p, m, v in place."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as called at pretrain.py:610 (L2, eps 1e-6).
    Scales `grads` in place; returns the total norm before clipping."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1.0:
        for g in grads:
            g.mul_(coef)
    return total


def allreduce_mean(per_rank_tensors, rescale_denom=1.0):
    """all_reduce_and_rescale_tensors, utils/distributed.py:15-42: Horovod's
    allreduce_ averages by default, then the explicit div by rescale_denom."""
    n = len(per_rank_tensors)
    return sum(per_rank_tensors) / n / rescale_denom


def warmup_linear(step, warmup_step, tot_step):
    """optim/sched.py:13-16."""
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def noam_schedule(step, warmup_step=4000):
    """optim/sched.py:7-10."""
    if step <= warmup_step:
        return step / warmup_step
    return (warmup_step ** 0.5) * (step ** -0.5)


def get_lr_sched(global_step, learning_rate, warmup_steps, num_train_steps, decay="linear"):
    """optim/sched.py:35-52 (linear / invsqrt / constant)."""
    if decay == "linear":
        lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    elif decay == "invsqrt":
        lr = learning_rate * noam_schedule(global_step, warmup_steps)
    elif decay == "constant":
        lr = learning_rate
    else:
        raise ValueError(decay)
    if lr <= 0:
        lr = 1e-8
    return lr


# --------------------------------------------------------------------------- #
# batch layout helpers (the contract the hot path consumes)
# --------------------------------------------------------------------------- #
def get_gather_index(txt_lens, num_bbs, batch_size, max_len, out_size):
    """data/data.py:376-384."""
    gi = torch.arange(0, out_size, dtype=torch.long).unsqueeze(0).repeat(batch_size, 1)
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        gi[i, tl:tl + nbb] = torch.arange(max_len, max_len + nbb, dtype=torch.long)
    return gi


# --------------------------------------------------------------------------- #
# convenience for tests: loss -> gradients via autograd over the plain ops above
# --------------------------------------------------------------------------- #
def grads_of(loss_fn, W, names=None):
    """Run loss_fn(Wg) with requires_grad copies of W; return (loss, {name: grad})."""
    Wg = OrderedDict((k, v.detach().clone().requires_grad_(v.is_floating_point())) for k, v in W.items())
    loss = loss_fn(Wg)
    loss.backward()
    out = {}
    for k, v in Wg.items():
        if v.is_floating_point() and v.grad is not None and (names is None or k in names):
            out[k] = v.grad
    return loss.detach(), out
