"""Optimizer construction with the reference's parameter grouping (optim/misc.py:9-32)."""
from .adamw import AdamW

NO_DECAY = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']   # case-sensitive substrings


def param_groups(model, weight_decay):
    named = list(model.named_parameters())
    return [
        {'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': weight_decay},
        {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0},
    ]


def build_optimizer(model, opts):
    if opts.optim != 'adamw':
        raise ValueError('uc2_amd ships the fused AdamW only (reference default); got %r' % (opts.optim,))
    return AdamW(param_groups(model, opts.weight_decay), lr=opts.learning_rate, betas=opts.betas)


def xlmr_pretrained_encoder_layer(n, load_layer):
    """is parameter `n` one that was loaded from the pretrained XLM-R (optim/misc.py:34-46): the text embeddings
    and the encoder layers 0..load_layer"""
    assert isinstance(load_layer, int)
    if 'roberta.encoder' in n:
        return int(n.split('.')[3]) <= load_layer
    return 'roberta.embeddings' in n


def xlmr_param_groups(model, weight_decay, learning_rate, xlmr_lr, load_layer=None):
    """the four groups of optim/misc.py:48-88: {pretrained XLM-R part at xlmr_lr, the rest at learning_rate} x
    {decayed, not decayed}; without load_layer the pretrained part is `roberta.embeddings` only"""
    named = list(model.named_parameters())
    if load_layer:
        assert isinstance(load_layer, int) and load_layer > 0
        pre = lambda n: xlmr_pretrained_encoder_layer(n, load_layer)
    else:
        pre = lambda n: 'roberta.embeddings' in n
    nd = lambda n: any(x in n for x in NO_DECAY)
    sel = lambda want_pre, want_nd: [p for n, p in named if pre(n) == want_pre and nd(n) == want_nd]
    return [{'params': sel(True, False), 'weight_decay': weight_decay, 'lr': xlmr_lr},
            {'params': sel(True, True), 'weight_decay': 0.0, 'lr': xlmr_lr},
            {'params': sel(False, False), 'weight_decay': weight_decay, 'lr': learning_rate},
            {'params': sel(False, True), 'weight_decay': 0.0, 'lr': learning_rate}]


def build_xlmr_optimizer(model, opts):
    """optim/misc.py:48-100 (AdamW only: the fused kernel is the one optimizer this package ships)"""
    if opts.optim != 'adamw':
        raise ValueError('uc2_amd ships the fused AdamW only (reference default); got %r' % (opts.optim,))
    groups = xlmr_param_groups(model, opts.weight_decay, opts.learning_rate, opts.xlmr_lr, getattr(opts, 'load_layer', None))
    return AdamW(groups, lr=opts.learning_rate, betas=opts.betas)
