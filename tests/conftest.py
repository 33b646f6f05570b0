import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes about a minute on 8 host cores")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically when no device is present, so a bare `pytest tests`
    # works on both kinds of machine; `-m gpu` / `-m "not gpu"` select explicitly.
    try:
        import torch
        # device_count() does not initialise the GPU in this process (is_available() does): tests/test_gpu_dist.py
        # has to start its rank processes from a process that has not touched the GPU yet
        have = torch.cuda.device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_accumulation_window():
    """the accumulation overlap (ops/streams.py) skips the first forward of a window when the PREVIOUS window had a single forward:
    process-wide history, reset so that no test depends on what ran before it"""
    m = sys.modules.get("uc2_amd.ops.streams")
    if m is not None:
        m.forget_accum_history()
    yield
