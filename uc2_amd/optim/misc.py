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
