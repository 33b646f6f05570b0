from .adamw import AdamW, clip_grad_norm_  # noqa: F401
from .misc import build_optimizer, param_groups  # noqa: F401
from .sched import get_lr_sched, warmup_linear  # noqa: F401
