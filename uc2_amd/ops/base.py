"""Constants of the C ABI (epilogue kinds, GEMM variants, flag bits), the instrumentation timers bench.py installs, and the
device-side dropout seed state.  Part of uc2_amd.ops (see ops/__init__.py)."""

import torch

from .. import _lib
from ..config import state


EPI_NONE, EPI_GELU, EPI_DGELU, EPI_ADD, EPI_TANH = 0, 1, 2, 3, 4


class GemmTimer:
    """HIP-event timing of every bf16 GEMM launch, grouped by kernel instantiation
    (trans_a, trans_b, kernel variant, accumulate-into-fp32); bench.py reports the group with the largest
    total time as the dominant kernel of the step"""
    VARIANT_TEMPLATE = {0: "128,64,2", 1: "256,64,3", 2: "256,32,3"}

    def __init__(self):
        self.groups = {}
        self.bytes = {}

    def add(self, key, flops, e0, e1, nbytes=0.0):
        """nbytes: ALGORITHMIC HBM bytes of the launch -- every operand read once, every result written once (fp32 partial tiles
        of a split-K launch included; the reduction pass is another kernel)"""
        self.groups.setdefault(key, []).append((flops, e0, e1))
        self.bytes[key] = self.bytes.get(key, 0.0) + nbytes

    def bytes_per_launch(self):
        """{kernel name: algorithmic bytes per launch, averaged over the group's launches}"""
        return {self.kernel_name(k): self.bytes.get(k, 0.0) / max(len(v), 1) for k, v in self.groups.items()}

    @staticmethod
    def kernel_name(key):
        if key[0] == "fp8":
            return "gemm_bf16_fast_kernel<false, false, true, 256, 64, 3, true> (fp8 e4m3, epilogue %d)" % key[1]
        ta, tb, variant, atomic, epi = key
        b = lambda x: "true" if x else "false"
        tacc = b(not atomic)
        if variant == 12:
            return "gemm_bf16_pp16_kernel<%s, %s, true, %d, 2>" % (b(ta), b(tb), epi)
        if variant in (5, 8, 9):
            return "gemm_bf16_pp_kernel<%s, %s, %s, %d, %d>" % (b(ta), b(tb), b(not atomic), epi, {8: 2, 9: 1, 5: 0}[variant])
        if variant == 99:
            return "gemm_bf16_kernel<%s, %s, %s>" % (b(ta), b(tb), tacc)
        if variant in (6, 7):
            return "gemm_bf16_ws_kernel<%s, %s, %s, 64, 3, %d>" % (b(ta), b(tb), tacc, 4 if variant == 6 else 8)
        return "gemm_bf16_fast_kernel<%s, %s, %s, %s>" % (b(ta), b(tb), tacc, GemmTimer.VARIANT_TEMPLATE.get(variant, "?"))

    def summary(self):
        """[(kernel name, launches, total flops, total seconds)] sorted by total time, descending;
        call after a device synchronize"""
        out = []
        for key, pairs in self.groups.items():
            tot_f = sum(f for f, _, _ in pairs)
            tot_t = sum(e0.elapsed_time(e1) for _, e0, e1 in pairs) * 1e-3
            out.append((self.kernel_name(key), len(pairs), tot_f, tot_t))
        return sorted(out, key=lambda r: -r[3])


class HbmTimer:
    """HIP-event timing (on the launch stream) of the HBM-bound kernels of the step, each with its ALGORITHMIC
    bytes (operands read once + results written once) -> GB/s against the HBM roofline"""

    def __init__(self):
        self.groups = {}

    def tick(self, name, nbytes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.groups.setdefault(name, []).append((nbytes, e0, e1))
        return e1

    def summary(self):
        out = []
        for name, recs in self.groups.items():
            tot_b = sum(r[0] for r in recs)
            tot_t = sum(r[1].elapsed_time(r[2]) for r in recs) * 1e-3
            out.append((name, len(recs), tot_b, tot_t))
        return sorted(out, key=lambda r: -r[3])


class _Timed:
    """with _Timed(name, bytes): launch  -- no-op unless bench.py installed config.state.hbm_timer"""
    __slots__ = ("e1",)

    def __init__(self, name, nbytes):
        t = state.hbm_timer
        self.e1 = t.tick(name, nbytes) if t is not None else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.e1 is not None:
            self.e1.record()
        return False


GEMM_AUTO, GEMM_GENERIC = -2, 99      # include/uc2_hip.h: UC2_GEMM_AUTO / UC2_GEMM_GENERIC
GEMM_DEFER_REDUCE = 1
GEMM_AUX_DERIV = 2                    # EPI_GELU saves gelu'(pre), EPI_DGELU multiplies by it as is
_FORCED = [None]                       # tests/diagnostics only (force_variant); production passes the plan per call


class force_variant:
    """with ops.force_variant(8): ...  -- every gemm() inside that does not name a variant itself uses this one
    (A/B tests and the bench_*.py diagnostics; the training path passes its tuned variant per call)"""

    def __init__(self, variant, flags=0):
        self.v = (variant, flags)

    def __enter__(self):
        self.prev, _FORCED[0] = _FORCED[0], self.v
        return self

    def __exit__(self, *exc):
        _FORCED[0] = self.prev
        return False


def _require_cuda(t):
    if not t.is_cuda:
        raise _lib.Uc2Error("uc2_amd kernels run on the GPU only (tensor on %s); there is no CPU fallback" % t.device)


# --------------------------------------------------------------------------------------
# dropout seed state (device side, so a captured hipGraph draws fresh masks on every replay)
# --------------------------------------------------------------------------------------
class _Rng:
    def __init__(self):
        self.state = {}
        self._scope = None

    def buf(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = str(device)
        if key not in self.state:
            self.state[key] = torch.full((1,), torch.initial_seed() & 0x7FFFFFFFFFFF, dtype=torch.int64, device=device)
        return self.state[key]

    def snapshot(self, device):
        """advance the stream and return a private copy for one forward/backward pair.  Inside `with rng.scope():` (one
        model forward) every caller gets the SAME copy -- one add + one clone per forward instead of one pair per
        BertLayer / LayerNorm (26 tiny launches per forward at 12 layers) -- and tells its sites apart with rng.site()."""
        sc = self._scope
        if sc is not None:
            key = str(torch.device(device))
            if key not in sc:
                sc[key] = self._fresh(device)
            return sc[key]
        return self._fresh(device)

    def _fresh(self, device):
        # (forwards never overlap each other -- ops.accum_pass orders every pass's stream behind the caller's, which waits for the
        #  previous forward -- so the one seed cell per device is advanced in forward order whatever stream a pass runs on)
        b = self.buf(device)
        b.add_(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFF)
        return b.clone()

    def site(self, imm):
        """seed offset of one dropout site.  Outside a scope: `imm` itself (every caller has its own seed copy).  Inside: the
        small site numbers (layer * 16 + k) are spread over 63 bits, so that two sites sharing the forward's seed do not draw
        masks that are XOR-shifted copies of each other (the kernels hash seed ^ index)."""
        if self._scope is None:
            return int(imm)
        return (int(imm) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF

    class _Scope:
        def __init__(self, rng):
            self.rng = rng

        def __enter__(self):
            self.prev, self.rng._scope = self.rng._scope, {}
            return self

        def __exit__(self, *exc):
            self.rng._scope = self.prev
            return False

    def scope(self):
        return _Rng._Scope(self)

    def manual_seed(self, seed, device="cuda"):
        self.buf(torch.device(device)).fill_(seed & 0x7FFFFFFFFFFF)


rng = _Rng()
