"""uc2_amd -- MI355X-native (gfx950) implementation of the UC2 encoder hot path.

Host side mirrors the reference's Python surface (model/layer.py, model/model.py,
model/itm.py, optim/adamw.py, utils/distributed.py); the compute is hand-written
HIP behind the C-ABI declared in include/uc2_hip.h (uc2_amd/csrc).
Importing this package never loads the shared library; the first compute call
does, and raises if it is missing -- there is no CPU fallback.
"""
from . import store  # noqa: F401
from .store import mark_all_dirty, set_compute_dtype, set_fp8  # noqa: F401

__version__ = "0.1.0"
