"""Shared helpers for the parity tests: golden-vector access and comparisons."""
import os

import numpy as np
import torch

from uc2_amd.utils import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN_DIR, "golden_%s.npz" % name), allow_pickle=False)
    return _cache[name]


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_rel(a, b, floor=None):
    """max |a-b| / max(|b|, floor); floor defaults to the rms of b (element-wise tolerance that
    does not blow up on near-zero entries)."""
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    if floor is None:
        floor = float(b.detach().pow(2).mean().sqrt()) + 1e-30
    return ((a - b).abs() / b.abs().clamp_min(floor)).max().item()


def check_against_golden(g, key, t, tol, what="", metric="max"):
    """t (tensor) vs golden entry `key`: full tensor if stored, else the 64-point slice and
    the three checksums (sum, abs-sum, l2)."""
    t = t.detach().float().cpu()
    if key + "/full" in g.files:
        ref = torch.from_numpy(g[key + "/full"])
        assert tuple(ref.shape) == tuple(t.shape), (key, ref.shape, t.shape)
        e = max_rel(t, ref) if metric == "max" else rel_err(t, ref)
        assert e < tol, "%s %s: %s rel err %.3e >= %.1e" % (what, key, metric, e, tol)
        return e
    sl = torch.from_numpy(g[key + "/slice"])
    idx = synth.slice_idx(t.numel())
    mine = t.flatten()[idx]
    floor = float(g[key + "/sum3"][2]) / max(1.0, t.numel() ** 0.5) + 1e-30     # rms of the full tensor
    e = max_rel(mine, sl, floor=floor) if metric == "max" else rel_err(mine, sl)
    assert e < tol, "%s %s: slice %s rel err %.3e >= %.1e" % (what, key, metric, e, tol)
    s = g[key + "/sum3"]
    td = t.double()
    l2 = td.pow(2).sum().sqrt().item()
    assert abs(l2 - s[2]) <= tol * max(s[2], 1e-30) * 2, "%s %s: l2 %.9g vs %.9g" % (what, key, l2, s[2])
    asum = td.abs().sum().item()
    assert abs(asum - s[1]) <= tol * max(s[1], 1e-30) * 2, "%s %s: abs-sum %.9g vs %.9g" % (what, key, asum, s[1])
    return e
