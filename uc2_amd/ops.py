"""Thin Python wrappers over the C-ABI kernels + the autograd Functions built from them.

Every function here enqueues HIP kernels from libuc2_hip.so on torch's current stream.
torch is used only for device memory (torch.empty), streams and autograd graph edges.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import call, dt, ptr, stream
from .store import store_of

EPI_NONE, EPI_GELU, EPI_DGELU, EPI_ADD, EPI_TANH = 0, 1, 2, 3, 4
ATTN_IMPL = 0          # 0 auto, 1 fp32-math kernels, 2 MFMA kernels (tests flip this)
GEMM_TIMER = None      # bench.py installs a GemmTimer to time one GEMM kernel variant with HIP events


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
        if variant in (8, 9):
            return "gemm_bf16_pp_kernel<%s, %s, %s, %d, %d>" % (b(ta), b(tb), b(not atomic), epi, 2 if variant == 8 else 1)
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


HBM_TIMER = None       # bench.py installs an HbmTimer: HIP-event timing of the HBM-bound kernels with their algorithmic bytes


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
    """with _Timed(name, bytes): launch  -- no-op unless bench.py installed ops.HBM_TIMER"""
    __slots__ = ("e1",)

    def __init__(self, name, nbytes):
        t = HBM_TIMER
        self.e1 = t.tick(name, nbytes) if t is not None else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.e1 is not None:
            self.e1.record()
        return False


GEMM_AUTO, GEMM_GENERIC = -2, 99      # include/uc2_hip.h: UC2_GEMM_AUTO / UC2_GEMM_GENERIC
_EXTRA_FLAGS = 0                       # diagnostics: UC2_GEMM_EXTRA_FLAGS=<int> is OR-ed into the flags of every uc2_gemm call (A/B of a
                                       # UC2_GEMM_DIAG route inside the whole training step; 0 in production)
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


# --------------------------------------------------------------------------------------
# raw kernel wrappers
# --------------------------------------------------------------------------------------
# Item queue of the persistent ping-pong GEMM (include/uc2_hip.h uc2_gemm_queued): dynamic work distribution from the third
# item of a workgroup on, for steps that overlap GEMMs with a communication kernel.  One 9-int queue per (device, stream):
# launches on one stream are serialised and the kernel leaves its queue zeroed.  UC2_GEMM_QUEUE=1 / ops.GEMM_QUEUE = True.
GEMM_QUEUE = False      # (with the weight-gradient side stream on one GPU the queue changed nothing: 61.6-61.7 ms without, 61.8-61.9 with)
_GEMM_QUEUES = {}


def _gemm_queue(device):
    key = (device.index, stream())
    q = _GEMM_QUEUES.get(key)
    if q is None:
        q = torch.zeros(16, dtype=torch.int32, device=device)
        _GEMM_QUEUES[key] = q
    return q


def gemm(a, b, M, N, K, *, ta=False, tb=False, out=None, out_f32=False, bias=None, epi=EPI_NONE,
         aux_in=None, aux_out=None, accumulate=False, split_k=1, lda=None, ldb=None, ldc=None, variant=None, flags=0,
         qkv_rows_d=0):
    """C[M,N] (=|+=) epi(sum_k A(m,k) B(n,k) + bias[n]); see uc2_amd/csrc/gemm.hip.
    variant: kernel to use for THIS call (None = the library's default for the shape); the plan travels with the
    call, the library holds no kernel-selection state."""
    _require_cuda(a)
    dtype = a.dtype
    assert b.dtype == dtype
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else dtype, device=a.device)
    c_f32 = out.dtype == torch.float32
    lda = lda if lda is not None else a.stride(0)
    ldb = ldb if ldb is not None else b.stride(0)
    ldc = ldc if ldc is not None else out.stride(0)
    ldaux = 0
    for x in (aux_in, aux_out):
        if x is not None and x.dim() == 2:
            ldaux = x.stride(0)
    flags |= _EXTRA_FLAGS
    if variant is None:
        if _FORCED[0] is not None:
            variant, fflags = _FORCED[0]
            flags |= fflags
        else:
            variant = GEMM_AUTO
    ws = None
    if variant in (8, 12) and c_f32 and split_k > 1:
        ws = _splitk_workspace(a.device, split_k * M * N * 4)
    two_stage = ws is not None
    timer = GEMM_TIMER
    if timer is not None and dtype == torch.bfloat16 and variant != GEMM_AUTO:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                # on torch's current stream == the stream the kernel is launched on
    else:
        e0 = None
    # qkv_rows_d = D > 0: a weight gradient whose rows come out in the head-interleaved q|k|v order (dW = dqkv^T x with interleaved
    # dqkv): the reduction pass puts them back into the parameter arena's order (uc2_gemm_splitk_reduce_qkv); two-stage only
    if qkv_rows_d and not two_stage:
        raise _lib.Uc2Error("qkv_rows_d needs the two-stage split-K path (variant 8 / 12, fp32 output, split_k > 1)")
    defer = two_stage and (e0 is not None or qkv_rows_d > 0)         # (timing: the GEMM kernel alone, the reduction pass separately)
    if defer:
        flags |= GEMM_DEFER_REDUCE
    if GEMM_QUEUE and dtype == torch.bfloat16:
        call("uc2_gemm_queued", dt(dtype), int(ta), int(tb), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc, int(c_f32),
             ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, int(accumulate), split_k, variant,
             ptr(ws), 0 if ws is None else ws.numel(), flags, ptr(_gemm_queue(a.device)), stream())
    else:
        call("uc2_gemm", dt(dtype), int(ta), int(tb), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc, int(c_f32),
             ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, int(accumulate), split_k, variant,
             ptr(ws), 0 if ws is None else ws.numel(), flags, stream())
    if e0 is not None:
        e1.record()
    if defer:
        if qkv_rows_d:
            call("uc2_gemm_splitk_reduce_qkv", M, N, ptr(out), ldc, split_k, int(accumulate), ptr(ws), ws.numel(), int(qkv_rows_d), stream())
        else:
            call("uc2_gemm_splitk_reduce", M, N, ptr(out), ldc, split_k, int(accumulate), ptr(ws), ws.numel(), stream())
    if e0 is not None:
        if variant in (8, 9, 12):  # ping-pong kernels: transposed accumulators unless fp32 atomics; the epilogue kind is a template argument
            epi_t = int(epi) + 4 if (flags & GEMM_AUX_DERIV and epi in (EPI_GELU, EPI_DGELU)) else int(epi)   # EPI_GELU_D = 5, EPI_MUL = 6
            key = (bool(ta), bool(tb), variant, bool(c_f32 and not two_stage), epi_t)
        else:
            key = (bool(ta), bool(tb), variant, bool(c_f32 and split_k > 1), 0)
        esz = a.element_size()
        nbytes = esz * (M * K + N * K) + out.element_size() * M * N * (split_k if two_stage else (2 if accumulate else 1))
        for x in (aux_in, aux_out):
            if x is not None:
                nbytes += x.element_size() * x.numel()
        timer.add(key, 2.0 * M * N * K, e0, e1, float(nbytes))
    return out


_SPLITK_WS = {}


def _splitk_workspace(device, nbytes):
    """caller-owned device scratch for the two-stage split-K reduction of the ping-pong kernel, handed to uc2_gemm with
    each call: grown on demand, one per (device, stream); every user runs on that stream, in order.  None when it
    cannot be (re)allocated (stream capture): the kernel then reduces with fp32 atomics."""
    key = (device.type, device.index, stream())      # per stream, like the item queues: two streams that both run split-K
    ws = _SPLITK_WS.get(key)                          # GEMMs (WGRAD_SIDE_STREAM) must not share one set of partial tiles
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            return None
        new = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        if ws is not None:
            ws.record_stream(torch.cuda.current_stream(device))      # the old buffer may still be read by kernels in flight
        _SPLITK_WS[key] = ws = new
    return ws


def _wgrad_split(dtype, n_out, n_in, rows):
    """split-K factor for a weight-gradient GEMM (contraction over `rows` tokens): the output is small
    (n_out x n_in), so the token axis is split until there are ~2 workgroups per CU (256 CUs); each slice keeps
    >= 1024 rows so the fp32 atomic reduction stays a small fraction of the traffic"""
    tm, tn = (64, 64) if dtype == torch.float32 else (256, 128)
    tiles = ((n_out + tm - 1) // tm) * ((n_in + tn - 1) // tn)
    s = (512 + tiles - 1) // max(tiles, 1)
    return max(1, min(s, (rows + 1023) // 1024))


# ---- per-shape kernel selection: measured once per (layout, shape) on the device, then cached ----------
AUTOTUNE = True
_TUNE = {}
_BORROWED = {}            # shapes without a measured plan that run on the plan of the nearest tuned token count (linear_dgrad)


def gemm_fallbacks(reset=False):
    """calls since load (or the last reset) whose plan named a ping-pong kernel but ran on another one (library counter)"""
    return int(_lib.load().uc2_gemm_fallback_count(int(bool(reset))))
_MAX_TUNED = 256          # cap on tuned shapes (each tuning costs ~30 candidates x 7 launches + a host sync)


def _bucket_key(key):
    ta, tb, M, N, K, wgrad = key
    r = lambda x: (x + 511) // 512 * 512
    return (ta, tb, M, N, r(K), wgrad) if wgrad else (ta, tb, r(M), N, K, wgrad)


def _plan_fits(plan, key):
    """can the kernel of `plan` run the shape `key` (else the library would silently take its generic kernel)"""
    v, sp = plan
    ta, tb, M, N, K, wgrad = key
    if v in (8, 9, 12, 13):
        rows = 192 if v == 9 else 256
        kt = K // 64
        per = ((kt + sp - 1) // sp + 1) & ~1
        # 32-bit staging offsets: an operand of 4 GiB or more does not run on the ping-pong kernels (gemm_fast.hip); contiguous
        # operands assumed here -- the library re-checks with the real leading dimensions, counts what it re-routes
        # (uc2_gemm_fallback_count) and refuses a UC2_GEMM_DEFER_REDUCE call it cannot honour
        if 2 * M * K >= 1 << 32 or 2 * N * K >= 1 << 32:
            return False
        return M % rows == 0 and N % 256 == 0 and K % 128 == 0 and kt - (sp - 1) * per >= 2
    if wgrad and sp * 1024 > K:
        return False
    return K % 64 == 0
_FWD_CANDIDATES = ((99, 1), (0, 1), (1, 1), (2, 1), (6, 1), (7, 1), (8, 1), (12, 1))     # (kernel variant, split_k); 99 = generic kernel, 8 = ping-pong, 12 = ping-pong on the 16x16x32 MFMA
_WGRAD_SPLITS = (2, 3, 4, 6, 8, 12, 16)


def _plan_key_str(key):
    ta, tb, M, N, K, wgrad = key
    return "%s%s %dx%dx%d%s" % ("T" if ta else "N", "T" if tb else "N", M, N, K, " wgrad" if wgrad else "")


def save_plans(path):
    """write the tuned (variant, split_k) table; a committed copy (uc2_amd/gemm_plans.json) is preloaded at
    import so that steady-state runs and profiles start without tuning launches"""
    import json
    with open(path, "w") as f:
        json.dump({_plan_key_str(k): list(v) for k, v in sorted(_TUNE.items())}, f, indent=1)


def load_plans(path):
    import json
    import os
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        table = json.load(f)
    n = 0
    for ks, v in table.items():
        parts = ks.split()
        ta, tb = parts[0][0] == "T", parts[0][1] == "T"
        M, N, K = (int(x) for x in parts[1].split("x"))
        _TUNE[(ta, tb, M, N, K, len(parts) > 2)] = (int(v[0]), int(v[1]))
        n += 1
    return n


def _time_gemm(fn, reps=5):
    fn()
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def gemm_plan(dtype, ta, tb, M, N, K, wgrad=False):
    """(variant, split_k) for one bf16 GEMM shape.  The first call for a shape times the candidate kernels
    (gemm.hip generic, gemm_fast.hip ring variants, split-K factors for weight gradients) on scratch buffers
    and caches the winner; later calls are a dict lookup.  fp32 (parity mode) and small shapes use defaults."""
    if dtype != torch.bfloat16:
        return -2, (_wgrad_split(dtype, M, N, K) if wgrad else 1)
    global GEMM_TIMER
    key = (bool(ta), bool(tb), M, N, K, wgrad)
    hit = _TUNE.get(key)
    if hit is not None:
        return hit
    default = (-2, _wgrad_split(dtype, M, N, K) if wgrad else 1)
    if (not AUTOTUNE) or float(M) * N * K < 2.0 ** 31 or torch.cuda.is_current_stream_capturing():
        return default
    # The token dimension (M forward / dgrad, K for weight gradients) changes almost every step under the reference's
    # token-bucket batching (data/sampler.py:11-59): tune one representative per 512-token bucket and reuse its plan
    # if the kernel accepts the real shape (tile divisibility is re-checked by the library, which falls back to the
    # generic kernel), and stop tuning after _MAX_TUNED shapes.
    bkey = _bucket_key(key)
    hit = _TUNE.get(bkey)
    if hit is not None and _plan_fits(hit, key):
        _TUNE[key] = hit
        return hit
    if len(_TUNE) >= _MAX_TUNED:
        return default
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    out = torch.zeros((M, N), dtype=torch.float32 if wgrad else torch.bfloat16, device=dev)
    lib = _lib.load()
    cands = [(v, s) for v in (99, 1, 0, 6) for s in _WGRAD_SPLITS if s * 1024 <= K] if wgrad else list(_FWD_CANDIDATES)
    if (not wgrad) and M % 192 == 0 and N % 256 == 0 and not ta:
        cands.append((9, 1))                      # ping-pong kernel with 192-row tiles (tile-count quantisation at N = 768)
    if wgrad and M % 256 == 0 and N % 256 == 0:
        # persistent ping-pong kernel: one (tile, split) item per CU, or two
        # (a split must leave every slice an even number >= 2 of k-tiles -- _plan_fits -- so the candidates are the valid
        #  factors closest to one item per CU, two, and a half: K = 9984 admits 13 and 26 but not 28, 14 or 56)
        tiles = (M // 256) * (N // 256)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        valid = [s for s in range(1, 129) if s * 256 <= K and _plan_fits((8, s), key)]
        for target in (cus, 2 * cus, cus // 2):
            if valid:
                sp_t = min(valid, key=lambda s: (abs(tiles * s - target), s))
                cands.append((8, sp_t))
                cands.append((12, sp_t))              # the same schedule on the 16x16x32 MFMA (within a few % of 8 either way here)
        cands = list(dict.fromkeys(cands))
    best, best_t = default, None
    timer_was, GEMM_TIMER = GEMM_TIMER, None          # tuning launches are not part of anybody's timed region
    # forward GEMMs (X W^T) are timed with a bias like the encoder's (the rolling-epilogue kernel only takes those)
    tbias = torch.zeros(N, dtype=torch.float32, device=dev) if (not wgrad and not ta and not tb) else None
    try:
        for v, sp in cands:
            t = _time_gemm(lambda: gemm(a, b, M, N, K, ta=ta, tb=tb, out=out, bias=tbias, accumulate=wgrad, split_k=sp, variant=v))
            if best_t is None or t < best_t:
                best, best_t = (v, sp), t
    finally:
        GEMM_TIMER = timer_was
    _TUNE[key] = best
    _TUNE.setdefault(bkey, best)
    return best


PP_SKEW = {}          # epilogue kind -> ping-pong start skew (UC2_GEMM_SKEW(n) flag); experiment knob, empty = off


def _gemm_planned(a, b, M, N, K, ta, tb, wgrad=False, **kw):
    """one GEMM with its tuned (variant, split_k) plan, passed to the library with the call"""
    v, sp = gemm_plan(a.dtype, ta, tb, M, N, K, wgrad)
    flags = kw.pop("flags", 0)
    if PP_SKEW and v in (8, 9, 12):
        flags |= (PP_SKEW.get(kw.get("epi", EPI_NONE), 0) & 15) << 4
    return gemm(a, b, M, N, K, ta=ta, tb=tb, split_k=sp, variant=v, flags=flags, **kw)


# --------------------------------------------------------------------------------------
# fp8 (e4m3) inputs for the forward / input-gradient GEMMs (BASELINE.json configs[4]); weight gradients stay bf16
# --------------------------------------------------------------------------------------
_FP8_CELLS = {}


def _fp8_cell(device):
    """a zeroed 4-byte amax cell + a 4-byte scale cell.  Cells come from a pool that is zero-filled once per 4096
    quantisations (a torch.zeros per tensor was one fill launch each, 384 per uc2-large step)"""
    pool = _FP8_CELLS.get(device)
    if pool is None or pool[1] >= pool[0].numel():
        pool = [torch.zeros(4096, dtype=torch.int32, device=device), 0, torch.empty(4096, dtype=torch.float32, device=device)]
        _FP8_CELLS[device] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1], pool[2][i:i + 1]


def fp8_amax(x2, amax=None):
    """amax cell (int32 bit pattern of the running maximum of |x|) of a contiguous tensor; pass `amax` to keep accumulating"""
    if amax is None:
        amax = _fp8_cell(x2.device)[0]
    assert x2.is_contiguous()
    call("uc2_fp8_amax", dt(x2.dtype), x2.numel(), ptr(x2), ptr(amax), stream())
    return amax


def fp8_quantize(x2, transpose=False, amax=None):
    """per-tensor power-of-two scaling, all on the device, two launches (amax, then scale + quantise):
    (x8 uint8 [rows, cols] or [cols, rows], scale fp32 [1]).  `amax`: a cell already holding the maximum (weights are
    quantised in both orientations from one amax pass)"""
    rows, cols = x2.shape
    assert x2.is_contiguous()
    if amax is None:
        amax = fp8_amax(x2)
    scale = _fp8_cell(x2.device)[1]
    out = torch.empty((cols, rows) if transpose else (rows, cols), dtype=torch.uint8, device=x2.device)
    call("uc2_fp8_quant_amax", dt(x2.dtype), rows, cols, ptr(x2), x2.stride(0), ptr(amax), ptr(scale), ptr(out), out.stride(0),
         int(transpose), stream())
    return out, scale


FP8_DELAYED = os.environ.get("UC2_FP8_DELAYED", "1") != "0"      # activations: delayed scaling (one pass) from the second use of a tensor role on
_FP8_HIST = {}             # tensor role -> [three amax cells (int32), index of the cell holding the previous maximum]
AMAX_CELLS = 16            # include/uc2_hip.h UC2_AMAX_CELLS: a maximum is kept in 16 cells (producers spread their atomics), three groups per role


def _fp8_rotate(h):
    """(previous, next, clear) device pointers of a role's three cell groups, and advance the role's history by one use"""
    cells, i = h
    base, step = cells.data_ptr(), 4 * AMAX_CELLS
    h[1] = (i + 1) % 3
    return base + i * step, base + ((i + 1) % 3) * step, base + ((i + 2) % 3) * step


_ST_UID = [0]


def _st_uid(st):
    """a number that names this parameter store for the life of the process (id() of a collected store can come back with another model)"""
    u = st.__dict__.get("_fp8_uid")
    if u is None:
        _ST_UID[0] += 1
        u = st.__dict__["_fp8_uid"] = _ST_UID[0]
    return u


_FP8_PREQ = {}             # data_ptr of a layer output -> (consumer layer id, (e4m3 copy, scale)) written by that layer's last LayerNorm for the
                           # next layer's QKV GEMM.  Cleared at the start of every top-level model forward (fp8_new_forward): an entry a
                           # forward that raised left behind must not meet the next step's tensor at a recycled address


def fp8_new_forward():
    """called by VLXLMRModel.forward before the first layer: drops a hand-over left by a forward pass that did not finish"""
    _FP8_PREQ.clear()
FP8_TAG = None             # set by the model's forward (task name, loss or scores): part of every role key, saved by BertLayerFn for its backward


def fp8_quantize_act(x2, key=None):
    """e4m3 copy + scale of an activation.  key = the tensor's role (store, layer, name): the first use computes the maximum just in
    time (two passes); every later use quantises with half the scale of the PREVIOUS use's maximum while accumulating its own for the
    next one -- one pass, no amax launch (uc2_fp8_quant_delayed)."""
    if key is None or not FP8_DELAYED or torch.cuda.is_current_stream_capturing():
        return fp8_quantize(x2)
    h = _FP8_HIST.get(key)
    if h is None:
        cells = torch.zeros(3 * AMAX_CELLS, dtype=torch.int32, device=x2.device)
        x8, scale = fp8_quantize(x2, amax=fp8_amax(x2, cells[0:1]))
        _FP8_HIST[key] = [cells, 0]
        return x8, scale
    rows, cols = x2.shape
    assert x2.is_contiguous()
    scale = _fp8_cell(x2.device)[1]
    out = torch.empty((rows, cols), dtype=torch.uint8, device=x2.device)
    prev, nxt, clr = _fp8_rotate(h)
    call("uc2_fp8_quant_delayed", dt(x2.dtype), rows, cols, ptr(x2), x2.stride(0), prev, nxt, clr, ptr(scale), ptr(out), out.stride(0), stream())
    return out, scale


def gemm_fp8(a8, sa, b8, sb, bias=None, epi=EPI_NONE, aux_in=None, aux_out=None, flags=0):
    """bf16 C[M,N] = epi((A8 . B8^T) / (sa * sb) + bias); A8 [M,K], B8 [N,K] e4m3 bytes, K % 128 == 0"""
    M, K = a8.shape
    N = b8.shape[0]
    assert b8.shape[1] == K
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    ldaux = 0
    for x in (aux_in, aux_out):
        if x is not None and x.dim() == 2:
            ldaux = x.stride(0)
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("uc2_gemm_fp8", M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), ptr(sa), ptr(sb), ptr(out), out.stride(0),
         ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, flags, stream())
    if timer is not None:
        e1.record()
        timer.add(("fp8", int(epi)), 2.0 * M * N * K, e0, e1)
    return out


def gemm_fp8_q(a8, sa, b8, sb, q_key, bias=None, epi=EPI_NONE, aux_in=None, aux_out=None, flags=0):
    """gemm_fp8 whose epilogue also writes the e4m3 copy of its output for the next GEMM (uc2_gemm_fp8_q; delayed scaling on the
    history of the CONSUMER's tensor role q_key).  -> (out, (q8, scale)), or None when that role has no history yet (its first use
    initialises it just in time, fp8_quantize_act) or the ping-pong kernel does not take the call."""
    h = _FP8_HIST.get(q_key) if (FP8_DELAYED and q_key is not None and not torch.cuda.is_current_stream_capturing()) else None
    if h is None:
        return None
    M, K = a8.shape
    N = b8.shape[0]
    if M % 256 or N % 256 or K % 256:                     # (the ping-pong kernel's shapes: do not advance the history for a call that cannot run)
        return None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    q8 = torch.empty((M, N), dtype=torch.uint8, device=a8.device)
    scale = _fp8_cell(a8.device)[1]
    ldaux = 0
    for x in (aux_in, aux_out):
        if x is not None and x.dim() == 2:
            ldaux = x.stride(0)
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    i_was = h[1]
    prev, nxt, clr = _fp8_rotate(h)
    rc = _lib.load().uc2_gemm_fp8_q(M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), ptr(sa), ptr(sb), ptr(out), out.stride(0),
                                    ptr(bias), epi, ptr(aux_in), ptr(aux_out), ldaux, flags, ptr(q8), q8.stride(0), prev, nxt, clr, ptr(scale), stream())
    if rc == -2:
        h[1] = i_was
        return None
    _lib.check(rc)
    if timer is not None:
        e1.record()
        timer.add(("fp8", int(epi)), 2.0 * M * N * K, e0, e1)
    return out, (q8, scale)


class _Fp8WeightItem(ctypes.Structure):          # Uc2Fp8WeightItem
    _fields_ = [("w", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("out", ctypes.c_void_p), ("out_t", ctypes.c_void_p),
                ("amax", ctypes.c_void_p), ("scale", ctypes.c_void_p)]


FP8_WEIGHT_BATCH = os.environ.get("UC2_FP8_WEIGHT_BATCH", "1") != "0"      # all e4m3 weight copies of a store in one call per optimizer step
# attention kernels write the e4m3 copies of ctx / dqkv themselves (uc2_attn_fwd_q / uc2_attn_bwd_q): OFF -- measured break-even on
# uc2-large (63.5-63.8 ms per step without, 63.9-64.4 with: the copies leave as 16-byte pieces of 32 different rows per wave
# instruction, +59 us forward / +31 us backward per launch against a 53 us stand-alone pass; profiles/r05_experiments.md section 2)
FP8_ATTN_FUSED = os.environ.get("UC2_FP8_ATTN_Q", "0") != "0"


def _fp8_weight(st, p_first, p_last, shape, transpose):
    """e4m3 copy (+ scale) of a weight span, re-quantised when the parameters change (AdamW step, load_state_dict).
    Every span a forward / backward has asked for is remembered; when the store's version moves, ALL of them are quantised again,
    both orientations, by one call (uc2_fp8_quant_weights_batch: 3 launches per 32 weights instead of 3 launches per weight)."""
    cache = st.__dict__.setdefault("_fp8_cache", {})
    key = (st.offsets[id(p_first)], st.offsets[id(p_last)], bool(transpose))
    hit = cache.get(key)
    if hit is not None and hit[2] == st.version:
        return hit[0], hit[1]
    w = st.span(st.data, p_first, p_last, shape)
    rows, cols = w.shape
    if FP8_WEIGHT_BATCH and rows % 64 == 0 and cols % 64 == 0 and w.is_contiguous() and not torch.cuda.is_current_stream_capturing():
        spans = st.__dict__.setdefault("_fp8_spans", {})
        skey = key[:2]
        if skey not in spans:
            dev = w.device
            spans[skey] = (p_first, p_last, shape, torch.empty((rows, cols), dtype=torch.uint8, device=dev),
                           torch.empty((cols, rows), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                           torch.empty(1, dtype=torch.float32, device=dev))
            todo = [skey]                                 # a span seen for the first time: quantise it alone, now
        else:
            todo = list(spans)                            # the parameters changed: every known span in one batch
        arr = (_Fp8WeightItem * len(todo))()
        for n_, k_ in enumerate(todo):
            pf, pl, shp, w8_, wt8_, am_, sc_ = spans[k_]
            wk = st.span(st.data, pf, pl, shp)
            arr[n_] = _Fp8WeightItem(wk.data_ptr(), wk.shape[0], wk.shape[1], w8_.data_ptr(), wt8_.data_ptr(), am_.data_ptr(), sc_.data_ptr())
        rc = _lib.load().uc2_fp8_quant_weights_batch(len(todo), arr, stream())
        if rc == 0:
            for k_ in todo:
                _, _, _, w8_, wt8_, _, sc_ = spans[k_]
                cache[(k_[0], k_[1], False)] = (w8_, sc_, st.version)
                cache[(k_[0], k_[1], True)] = (wt8_, sc_, st.version)
            hit = cache[key]
            return hit[0], hit[1]
        if rc != -2:
            _lib.check(rc)
        del spans[skey]
    akey = (key[0], key[1], "amax")
    ahit = cache.get(akey)
    if ahit is None or ahit[1] != st.version:                # one amax pass serves both orientations
        ahit = (fp8_amax(w), st.version)
        cache[akey] = ahit
    w8, sc = fp8_quantize(w, transpose, amax=ahit[0])
    cache[key] = (w8, sc, st.version)
    return w8, sc


def linear_fwd_fp8(x2, st, p_first, p_last, shape, bias, epi=EPI_NONE, aux_out=None, flags=0, role=None, tag=None, pre_q=None, q_key=None):
    """pre_q: (x8, scale) already produced by the GEMM that made x2 (its fused e4m3 stream); q_key: the consumer's tensor role of THIS
    GEMM's output -- returns (y, (y8, scale) or None) then"""
    w8, sw = _fp8_weight(st, p_first, p_last, shape, False)
    x8, sx = pre_q if pre_q is not None else fp8_quantize_act(x2, None if role is None else (_st_uid(st), st.offsets[id(p_first)], "fwd", role, tag))
    if q_key is not None:
        r = gemm_fp8_q(x8, sx, w8, sw, q_key, bias=bias, epi=epi, aux_out=aux_out, flags=flags)
        return r if r is not None else (gemm_fp8(x8, sx, w8, sw, bias=bias, epi=epi, aux_out=aux_out, flags=flags), None)
    return gemm_fp8(x8, sx, w8, sw, bias=bias, epi=epi, aux_out=aux_out, flags=flags)


def linear_dgrad_fp8(dy2, st, p_first, p_last, shape, epi=EPI_NONE, aux_in=None, colsum_out=None, flags=0, role=None, tag=None, pre_q=None, q_key=None):
    """dX = epi(dY W): the k-contiguous operand is the transposed e4m3 copy of W ([in, out])"""
    wt8, sw = _fp8_weight(st, p_first, p_last, shape, True)
    d8, sd = pre_q if pre_q is not None else fp8_quantize_act(dy2, None if role is None else (_st_uid(st), st.offsets[id(p_first)], "bwd", role, tag))
    if q_key is not None:
        r = gemm_fp8_q(d8, sd, wt8, sw, q_key, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)
        return r if r is not None else (gemm_fp8(d8, sd, wt8, sw, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags), None)
    return gemm_fp8(d8, sd, wt8, sw, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)


def linear_fwd(x2, w, bias, epi=EPI_NONE, aux_out=None, flags=0):
    M, K = x2.shape
    N = w.shape[0]
    return _gemm_planned(x2, w, M, N, K, False, False, bias=bias, epi=epi, aux_out=aux_out, flags=flags)


EPI_DROPADD = 10                 # internal to the ping-pong kernel (uc2_gemm_drop_residual)
# dropout + residual of the dense -> dropout -> LayerNorm tails in the GEMM epilogue: bit 0 = the attention-output tail, bit 1 = the FFN tail
LN_FUSE = int(os.environ.get("UC2_LN_FUSE", "3"))
LN_FUSE_MIN_ROWS = int(os.environ.get("UC2_LN_FUSE_MIN_ROWS", "16384"))


def linear_drop_residual(x2, w, bias, res2, drop_p, seed, seed_imm):
    """s = dropout(x2 w^T + bias) + res2 (model/layer.py:111-115, :152-156 up to the LayerNorm) from one GEMM launch, with the mask
    ln_fwd / ln_bwd derive from (seed, seed_imm); None (nothing launched) when the ping-pong kernel does not take the shape"""
    M, K = x2.shape
    N = w.shape[0]
    if x2.dtype != torch.bfloat16 or M % 256 or N % 256 or K % 128:
        return None
    out = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    timer = GEMM_TIMER
    e0 = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    flags = _EXTRA_FLAGS
    if PP_SKEW:
        flags |= (PP_SKEW.get(EPI_ADD, 0) & 15) << 4
    rc = _lib.load().uc2_gemm_drop_residual(M, N, K, ptr(x2), x2.stride(0), ptr(w), w.stride(0), ptr(out), N, ptr(bias), ptr(res2),
                                            res2.stride(0), float(drop_p), ptr(seed), seed_imm, flags,
                                            ptr(_gemm_queue(x2.device)) if GEMM_QUEUE else None, stream())
    if rc == -2:
        return None
    _lib.check(rc)
    if e0 is not None:
        e1.record()
        timer.add((False, False, 12, False, EPI_DROPADD), 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + 2 * M * N))
    return out


DGRAD_TRANSPOSED_W = os.environ.get("UC2_DGRAD_WT", "1") != "0"     # bf16: dX = dY W reads a k-contiguous copy W^T (store.compute_t)
DGRAD_WT_MIN_ROWS = 16384        # ... from this many tokens (its own knob, not the side stream's: at the reference's 104-pair
                                 # micro-batch the k-contiguous form is no faster on the ring kernels and 8 % slower for the
                                 # gelu'-multiply GEMM, scratch/nn_dgrad_small.py; at 38 400 rows 2-4 % faster, at 98 304 4-10 %)
DGRAD_ROUTES = {}                # (M, N, K, epilogue) -> "W^T" | "W": which form linear_dgrad took, for the run record (bench.py)


def linear_dgrad(dy2, w, epi=EPI_NONE, aux_in=None, colsum_out=None, flags=0, wt=None):
    """dX[M,K] = epi(dY[M,N] @ W[N,K])  (W in nn.Linear layout).  With EPI_DGELU, colsum_out (fp32 [K]) += column sums
    of dX = the bias gradient of the layer below (fused into the GEMM epilogue where the kernel supports it).
    wt: optional k-contiguous copy W^T [K, N] (ParamStore.compute_t): the GEMM then reads both operands k-contiguously, which
    the ping-pong kernel on the 16x16x32 MFMA does 4-10 % faster than the transposing LDS read of W."""
    M, N = dy2.shape
    K = w.shape[1]
    if colsum_out is not None and epi != EPI_DGELU:
        raise _lib.Uc2Error("colsum_out needs EPI_DGELU")
    # The plan table is keyed by (layout, shape), not by epilogue: the k-contiguous form is taken where the plan of the NN
    # shape is the 16x16x32 kernel (it has every epilogue of this path for it).  The lookup must never start a tuning pass
    # inside a backward (a dozen timed launches + a host sync for a shape the forward never ran): untuned shapes keep W.
    route = "W"
    if wt is not None:
        key = (False, False, M, K, N, False)
        hit = _TUNE.get(key) or _TUNE.get(_bucket_key(key)) or _BORROWED.get(key)
        if hit is None and M >= DGRAD_WT_MIN_ROWS:
            # no plan for this token count: take the plan of the nearest tuned token count of the same (N, K) -- above the
            # threshold the choice between the kernels does not depend on M any more (every committed plan there is variant 12)
            near = [(abs(k[2] - M), v) for k, v in _TUNE.items() if not k[0] and not k[1] and not k[5] and k[3] == K and k[4] == N
                    and k[2] >= DGRAD_WT_MIN_ROWS]
            if near:
                hit = min(near, key=lambda t: t[0])[1]
                _BORROWED[key] = hit                    # not a measured plan: kept out of _TUNE (save_plans, bench.py's gemm_plans)
        if hit is not None and hit[0] == 12 and _plan_fits(hit, key):
            route = "W^T"
    if wt is not None:
        DGRAD_ROUTES[(M, K, N, int(epi))] = route
    if route == "W^T":
        if PP_SKEW:
            flags |= (PP_SKEW.get(epi, 0) & 15) << 4
        return gemm(dy2, wt, M, K, N, split_k=1, variant=hit[0], epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)
    return _gemm_planned(dy2, w, M, K, N, False, True, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)


WGRAD_SPARE = 0                 # with WGRAD_SIDE_STREAM: weight-gradient GEMMs leave 8 * WGRAD_SPARE CUs free (UC2_GEMM_SPARE)


def _linear_wgrad_now(dy2, x2, dw, db):
    M, N = dy2.shape
    K = x2.shape[1]
    _gemm_planned(dy2, x2, N, K, M, True, True, wgrad=True, out=dw, accumulate=True, flags=(WGRAD_SPARE & 7) << 28)
    if db is not None:
        colsum_accum(dy2, db)


# Weight-gradient GEMMs are off the critical path of backward (nothing downstream in the same backward pass reads
# dW), so they are enqueued on a side HIP stream and overlap the dgrad / LayerNorm / attention chain on the main
# stream.  Every consumer of gradients (optimizer, clipping, all-reduce, end of autograd's backward) joins first.
WGRAD_SIDE_STREAM = True        # round 3, inside the 1024-pair step on one box: 63.1-63.3 ms against 63.8-64.4 on the main stream alone
                                # (the tails and launch gaps of the memory-bound kernels fill with weight-gradient tiles); the persistent
                                # GEMMs then take their work items from the per-XCD queue (workgroups of two kernels share the CUs).
                                # Leaving CUs free for the other stream (UC2_GEMM_SPARE) made it slower (65.5 ms at 16-24 CUs).
                                # UC2_WGRAD_SIDE=0 turns it off.
_side_streams = {}
_side_dirty = set()
_join_queued = [False]


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        # (same priority as the main stream: torch on ROCm offers (0, -1) only; a high-priority side stream measured 111.8-116.7 ms
        #  per step against 111.3-111.5)
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


_side_keep = []         # tensors the side stream reads: kept alive until the join (see _on_side_stream)


def join_side_streams(compute=True):
    """make the current stream wait for every weight-gradient kernel enqueued on a side stream and (compute=True: what every
    consumer of gradients asks for -- optimizer, clipping, all-reduce, zero_grad) for the forward / backward passes that the
    accumulation overlap put on its own two streams (accum_overlap below)"""
    for key in list(_side_dirty):
        torch.cuda.current_stream(torch.device(*key)).wait_stream(_side_streams[key])
    _side_dirty.clear()
    _side_keep.clear()                 # from here on the main stream is ordered behind their last reader: the blocks may be reused
    _join_queued[0] = False            # (a backward that raised never ran its callback: the next one must queue a new join)
    if compute:
        join_accum_streams()


def _queue_pass_callback(fn):
    """autograd end-of-pass callback that runs `fn` on the stream that is current NOW (the stream of the backward node that
    registers it).  The engine runs final callbacks in the thread and stream context of whoever called backward(); a pass whose
    forward ran on another stream (utils/pipeline.py, accum_overlap) must flush its deferred launches there."""
    s = torch.cuda.current_stream()

    def run():
        with torch.cuda.stream(s):
            fn()
    torch.autograd.Variable._execution_engine.queue_callback(run)


def _end_of_backward_join():
    _join_queued[0] = False
    join_side_streams(compute=False)


def pending_side_stream(device):
    """the weight-gradient side stream of `device` if kernels have been enqueued on it since the last join, else None.
    GradSync orders a layer's all-reduce behind it (uc2_comm_allreduce_bucket_after) instead of joining it into the main stream."""
    key = (device.type, device.index)
    return _side_streams[key] if key in _side_dirty else None


# --------------------------------------------------------------------------------------
# Gradient accumulation, overlapped without an API change.  The reference's loop runs micro-batch after micro-batch
# (pretrain.py:514-566: forward, backward, forward, backward, ..., all-reduce, clip, step; config/uc2_pretrain.json:17-19: 104
# pairs x 3).  At that size most kernels of a pass leave CUs idle (9 984 tokens = 117 tiles of 256 x 256 for 256 CUs), and nothing
# in micro-batch i+1's FORWARD depends on micro-batch i's BACKWARD: the weights only change at the optimizer step.  The top-level
# models (VLXLMRForPretraining / VLXLMRForImageTextRetrieval) therefore run a TRAINING forward of fewer than ACCUM_OVERLAP_MAX_ROWS
# tokens on one of two library-owned streams, alternating per call; autograd runs each node's backward on its forward's stream, so
# backward i is on stream i & 1 and the loop's next forward, enqueued right after it on the other stream, runs beside it:
#   * entry: the pass's stream waits for the caller's current stream (inputs, the optimizer's weights); exit: the caller's stream
#     waits for the pass's stream (the returned losses / scores are safe to use there) -- NOT for any backward;
#   * gradient accumulation stays in order: a pass's first backward node (_AccumMarker) makes its stream wait for the other one
#     (the arena's += are not atomic);
#   * every consumer of gradients (AdamW.step, clip_grad_norm_, all_reduce_and_rescale_tensors, zero_grad -- they all call
#     join_side_streams()) makes its stream wait for both; the per-layer all-reduce hooks of GradSync run inside the pass.
# Same micro-batches, same dropout seeds in the same order, bit-identical gradients (same kernels, same accumulation order).
# Off for: eval / no-grad forwards, fp8 stores (delayed-scaling histories assume one in-order stream), stores whose bf16 copies
# are re-cast at every forward (store.auto_sync: the re-cast would race with the backward beside it; AdamW.step turns auto_sync
# off), stream capture, UC2_ACCUM_OVERLAP=0.  Measured: profiles/r06_experiments.md.
ACCUM_OVERLAP = os.environ.get("UC2_ACCUM_OVERLAP", "1") != "0"
ACCUM_OVERLAP_MAX_ROWS = 16384
_accum = {}


class _AccumState:
    def __init__(self, device):
        self.device = device
        self.streams = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        self.k = 0                         # eligible forwards since the last join
        self.used = [False, False]
        self.passes = 0                    # (statistics: passes that ran on the overlap streams)


def _accum_state(device):
    key = (device.type, device.index)
    st = _accum.get(key)
    if st is None:
        st = _accum[key] = _AccumState(device)
    return st


def join_accum_streams():
    """the current stream waits for every pass enqueued on the accumulation-overlap streams; the next pass starts on stream 0"""
    for st in _accum.values():
        if st.used[0] or st.used[1]:
            cur = torch.cuda.current_stream(st.device)
            for i in (0, 1):
                if st.used[i]:
                    cur.wait_stream(st.streams[i])
            st.used = [False, False]
        st.k = 0


class _AccumMarker(torch.autograd.Function):
    """identity on a pass's output: its backward is the first node of the pass's backward (it runs on the pass's stream) and
    orders that stream behind the other one, i.e. behind the previous micro-batch's backward"""

    @staticmethod
    def forward(ctx, x, state, idx):
        ctx.state, ctx.idx = state, idx
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        st, i = ctx.state, ctx.idx
        if st.used[1 - i]:
            st.streams[i].wait_stream(st.streams[1 - i])
        return g, None, None


class accum_pass:
    """`with accum_pass(model_store, rows, tensors) as ap: out = ap.mark(forward(...))` -- see the block comment above.
    Inactive (a plain pass on the caller's stream) whenever one of the conditions does not hold."""

    def __init__(self, store, rows, tensors, fp8=False):
        self.state = None
        t0 = next((t for t in tensors if torch.is_tensor(t) and t.is_cuda), None)
        if not (ACCUM_OVERLAP and t0 is not None and torch.is_grad_enabled() and 0 < rows < ACCUM_OVERLAP_MAX_ROWS and not fp8
                and not (store.shadow is not None and store.auto_sync) and not torch.cuda.is_current_stream_capturing()):
            return
        self.state = _accum_state(t0.device)
        self.store = store
        self.tensors = [t for t in tensors if torch.is_tensor(t) and t.is_cuda]

    def __enter__(self):
        st = self.state
        if st is None:
            return self
        self.store.pin_grad_accumulators()          # (on the caller's stream, before the switch: see store.py)
        self.idx = st.k & 1
        st.k += 1
        st.passes += 1
        S = st.streams[self.idx]
        self.main = torch.cuda.current_stream(st.device)
        S.wait_stream(self.main)
        for t in self.tensors:                      # allocated on the caller's stream, read by this pass (and its backward) on S
            t.record_stream(S)
        st.used[self.idx] = True
        self._ctx = torch.cuda.stream(S)
        self._ctx.__enter__()
        return self

    def mark(self, out):
        """tag the tensors of `out` that carry a graph (losses / scores) and make them usable on the caller's stream"""
        st = self.state
        if st is None:
            return out

        def one(t):
            if not torch.is_tensor(t):
                return t
            if t.requires_grad:
                t = _AccumMarker.apply(t, st, self.idx)
            if t.is_cuda:
                t.record_stream(self.main)
            return t
        if isinstance(out, (tuple, list)):
            return type(out)(one(t) for t in out)
        return one(out)

    def __exit__(self, *exc):
        st = self.state
        if st is None:
            return False
        self._ctx.__exit__(*exc)
        self.main.wait_stream(st.streams[self.idx])
        return False


WGRAD_SIDE_MIN_ROWS = 16384      # below this many tokens the step is close to host-bound and the extra event / stream traffic makes it
                                 # erratic (104-pair micro-batches: 27.6-45.5 ms per optimizer step with the side stream, 30.8 without)


def _side_route(rows):
    return WGRAD_SIDE_STREAM and rows >= WGRAD_SIDE_MIN_ROWS and not torch.cuda.is_current_stream_capturing()


def _on_side_stream(dev, fn, inputs):
    """run fn() on the side stream of `dev`, ordered after everything enqueued so far on the current stream; `inputs` are the
    tensors it reads.  They are kept ALIVE (a reference, not Tensor.record_stream) until the join: a block marked with
    record_stream cannot be reused before the side stream's event has completed on the device, and with the host several steps
    ahead of the GPU the caching allocator then answers every new request with a fresh hipMalloc -- round 4 measured 32-187 device
    allocations and 21-110 GB of pool growth INSIDE a 10-step timed region, and one run in eight at 146-165 ms per step instead
    of 110.  Released after the join, the blocks return to the main stream's pool in stream order: no event, no growth.
    The side stream is joined at the end of the backward pass (autograd callback), or right away outside one."""
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)                       # inputs were produced on the main stream
    with torch.cuda.stream(side):
        fn()
    _side_keep.extend(t for t in inputs if t is not None)
    was_clean = not _side_dirty
    _side_dirty.add((dev.type, dev.index))
    if was_clean or not _join_queued[0]:
        # (keyed on the dirty set going non-empty, not only on the flag: a backward that raised after queueing never runs its
        #  callback, and a flag left set would keep every later backward from registering the join)
        try:
            _queue_pass_callback(_end_of_backward_join)
            _join_queued[0] = True
        except RuntimeError:                     # not inside a backward pass: join right away
            join_side_streams()


def linear_wgrad(dy2, x2, dw, db):
    """dW[N,K] += dY^T X ; db[N] += colsum(dY)  (fp32 accumulation buffers)"""
    if not _side_route(dy2.shape[0]):
        return _linear_wgrad_now(dy2, x2, dw, db)
    _on_side_stream(dy2.device, lambda: _linear_wgrad_now(dy2, x2, dw, db), (dy2, x2))


# Below WGRAD_SIDE_MIN_ROWS tokens a layer's four weight gradients are too small to fill the chip one by one (9-36 output tiles
# each at ~10 k tokens = a single round on half of the 256 CUs): BertLayerFn.backward hands them to ONE launch of the persistent
# ping-pong kernel (uc2_gemm_wgrad_group) at the end of the layer's backward.  104-pair micro-batch: 215 -> 136 us per layer.
WGRAD_GROUP = os.environ.get("UC2_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP_SIDE = os.environ.get("UC2_GROUP_SIDE", "1") == "1"      # the grouped launch runs on the side stream (round 4, reference regime,
                                                                     # same box x 3: 27.70-27.72 -> 27.11-27.27 ms per optimizer step, steady;
                                                                     # round 3's per-GEMM side stream at this size was erratic)


class _WgradItem(ctypes.Structure):             # include/uc2_hip.h: Uc2WgradItem
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("lddy", ctypes.c_int),
                ("ldx", ctypes.c_int), ("lddw", ctypes.c_int), ("n_out", ctypes.c_int), ("n_in", ctypes.c_int),
                ("split_k", ctypes.c_int)]


_CUS = {}


def _num_cus(device):
    key = (device.type, device.index)
    if key not in _CUS:
        _CUS[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return _CUS[key]


def _group_split(tiles, ktiles, cus):
    """common split-K factor of a grouped weight-gradient launch: fewest rounds of equal items, counting ~8 k-tile times per
    round for the partial-tile epilogue, the next item's first fetch and the reduction's share (measured at 108 tiles x 156
    k-tiles: split 2 = 136 us, 3 = 170, 4 = 166, 6 = 183; one k-tile = 1.65 us)"""
    best, best_t = 1, None
    for s in range(1, 17):
        per = ((ktiles + s - 1) // s + 1) & ~1
        if per < 2 or (s - 1) * per >= ktiles or ((ktiles - (s - 1) * per) & 1):
            continue
        rounds = (tiles * s + cus - 1) // cus
        t = rounds * (per + 8) + 0.5 * s
        if best_t is None or t < best_t:
            best, best_t = s, t
    return best


def wgrad_group(triples):
    """dW_i += dY_i^T X_i for every (dY_i, X_i, dW_i) of `triples` (at most four, same token count), one launch + one reduction;
    falls back to one GEMM per item where the grouped kernel does not apply"""
    dy0 = triples[0][0]
    rows = dy0.shape[0]
    ok = (WGRAD_GROUP and dy0.dtype == torch.bfloat16 and 1 <= len(triples) <= 4 and rows % 128 == 0
          and all(dy.shape[0] == rows and x.shape[0] == rows and dy.shape[1] % 256 == 0 and x.shape[1] % 256 == 0
                  and dy.is_contiguous() and x.is_contiguous() and dw.dtype == torch.float32 for dy, x, dw in triples))
    if ok:
        tiles = sum((dy.shape[1] // 256) * (x.shape[1] // 256) for dy, x, _ in triples)
        split = _group_split(tiles, rows // 64, _num_cus(dy0.device))
        arr = (_WgradItem * len(triples))()
        for i, (dy, x, dw) in enumerate(triples):
            arr[i] = _WgradItem(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), dy.stride(0), x.stride(0), dw.stride(0),
                                dy.shape[1], x.shape[1], split)
        lib = _lib.load()
        need = lib.uc2_gemm_wgrad_group_workspace(len(triples), arr)
        ws = _splitk_workspace(dy0.device, need)             # (None: it would have to grow inside a stream capture)
        rc = -2 if ws is None else lib.uc2_gemm_wgrad_group(dt(dy0.dtype), len(triples), arr, rows, ptr(ws), ws.numel(), stream())
        if rc == 0:
            return
        if rc != -2:
            _lib.check(rc)
    for dy, x, dw in triples:
        _linear_wgrad_now(dy, x, dw, None)


def colsum_accum(x2, out, rowmask=None):
    """out[n] += sum over (masked) rows of x2[:, n]; out is fp32"""
    M, N = x2.shape
    call("uc2_colsum_accum", dt(x2.dtype), M, N, ptr(x2), x2.stride(0), ptr(rowmask), ptr(out), stream())


def _fp8_hist_for(q_key, device):
    """the history of a tensor role if a producer may fuse its quantisation now (fp8 delayed scaling on, the role has been used)"""
    if q_key is None or not FP8_DELAYED or torch.cuda.is_current_stream_capturing():
        return None
    return _FP8_HIST.get(q_key)


def ln_fwd(x2, res2, gamma, beta, eps, drop_p=0.0, seed=None, seed_imm=0, want_stats=True, drop_after=False, q_key=None):
    """y = LN(dropout(x) + res) (drop_after False: the encoder's dense->dropout->LN tails) or
    y = dropout(LN(x + res)) (drop_after True: the embedding tails, model/model.py:331-333,361-363).
    q_key (fp8 mode): the tensor role of y at the GEMM that reads it -- returns a 4th value, (y8, scale) written by the same kernel
    (uc2_ln_fwd_q, delayed scaling) or None when that role has no history yet / the kernel does not take the shape"""
    M, H = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(M, dtype=torch.float32, device=x2.device) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device) if want_stats else None
    h = _fp8_hist_for(q_key, x2.device) if x2.dtype == torch.bfloat16 and H % 8 == 0 and H <= 1024 else None
    q = None
    with _Timed("ln_fwd", M * H * x2.element_size() * (3 if res2 is not None else 2)):
        if h is not None:
            y8 = torch.empty((M, H), dtype=torch.uint8, device=x2.device)
            scale = _fp8_cell(x2.device)[1]
            i_was = h[1]
            prev, nxt, clr = _fp8_rotate(h)
            rc = _lib.load().uc2_ln_fwd_q(dt(x2.dtype), M, H, ptr(x2), ptr(res2), ptr(gamma), ptr(beta), eps, drop_p, int(drop_after),
                                          ptr(seed), seed_imm, ptr(y), ptr(mean), ptr(rstd), ptr(y8), prev, nxt, clr, ptr(scale), stream())
            if rc == -2:
                h[1] = i_was
                h = None
            else:
                _lib.check(rc)
                q = (y8, scale)
        if h is None:
            call("uc2_ln_fwd", dt(x2.dtype), M, H, ptr(x2), ptr(res2), ptr(gamma), ptr(beta), eps, drop_p, int(drop_after),
                 ptr(seed), seed_imm, ptr(y), ptr(mean), ptr(rstd), stream())
    return (y, mean, rstd, q) if q_key is not None else (y, mean, rstd)


LN_REDUCE_SIDE = os.environ.get("UC2_LN_REDUCE_SIDE", "1") != "0"


def ln_bwd(dy2, x2, res2, gamma, mean, rstd, dgamma, dbeta, drop_p=0.0, seed=None, seed_imm=0, need_dres=True,
           dbias=None, drop_after=False, q_key=None):
    """returns (dx, dres); with drop_p == 0 they are the same tensor.  dbias (optional, fp32 [H]) accumulates
    the column sum of dx: the bias gradient of the dense layer that produced x, for free in the same pass.
    Two kernels: the streaming pass (dx, dres, per-workgroup partial column sums) and a small reduction of the partials into
    dgamma / dbeta / dbias.  Nothing in the backward chain reads those three, so where the weight gradients run on the side
    stream the reduction goes there too: beside a persistent weight-gradient GEMM that owns every CU, the 5 us kernel waited
    ~115 us for a CU with the whole input-gradient chain queued behind it (24 times per step)."""
    M, H = x2.shape
    lib = _lib.load()
    ws = torch.empty(lib.uc2_ln_bwd_workspace(M, H) // 4, dtype=torch.float32, device=x2.device)
    dx = torch.empty_like(x2)
    dres = torch.empty_like(x2) if (drop_p > 0.0 and need_dres and drop_after != 1) else None       # (drop_after: False / True / 2)
    streams = 3 + (1 if res2 is not None else 0) + (1 if dres is not None else 0)     # dy, x, (res) in; dx, (dres) out
    d = dt(x2.dtype)
    h = _fp8_hist_for(q_key, x2.device) if x2.dtype == torch.bfloat16 else None
    q = None
    with _Timed("ln_bwd", M * H * x2.element_size() * streams):
        if h is not None:                  # fp8 mode: the same pass writes the e4m3 copy of dx the input-gradient GEMM reads
            d8 = torch.empty((M, H), dtype=torch.uint8, device=x2.device)
            scale = _fp8_cell(x2.device)[1]
            i_was = h[1]
            prev, nxt, clr = _fp8_rotate(h)
            rc = lib.uc2_ln_bwd_partial_q(d, M, H, ptr(dy2), ptr(x2), ptr(res2), ptr(gamma), ptr(mean), ptr(rstd), drop_p,
                                          int(drop_after), ptr(seed), seed_imm, ptr(dx), ptr(dres), int(dbias is not None), ptr(ws),
                                          ptr(d8), prev, nxt, clr, ptr(scale), stream())
            if rc == -2:
                h[1] = i_was
                h = None
            else:
                _lib.check(rc)
                q = (d8, scale)
        if h is None:
            call("uc2_ln_bwd_partial", d, M, H, ptr(dy2), ptr(x2), ptr(res2), ptr(gamma), ptr(mean), ptr(rstd), drop_p,
                 int(drop_after), ptr(seed), seed_imm, ptr(dx), ptr(dres), int(dbias is not None), ptr(ws), stream())

    _ln_bwd_second_stage(d, M, H, ws, dgamma, dbeta, dbias, x2.device)
    if q_key is not None:
        return dx, (dres if dres is not None else dx), q
    return dx, (dres if dres is not None else dx)


def _ln_bwd_second_stage(d, M, H, ws, dgamma, dbeta, dbias, device):
    """the reduction of a LayerNorm backward's partial column sums (ws) into dgamma / dbeta / dbias: on the side stream, queued for
    the end of the pass, or right away (see ln_bwd)"""
    def reduce():
        call("uc2_ln_bwd_reduce", d, M, H, ptr(ws), ptr(dgamma), ptr(dbeta), ptr(dbias), stream())
    if dgamma is not None or dbeta is not None or dbias is not None:
        if LN_REDUCE_SIDE and _side_route(M):      # (below WGRAD_SIDE_MIN_ROWS tokens it made the regime erratic: 27.0-30.8 ms against 27.0-27.1)
            _on_side_stream(device, reduce, (ws,))
        elif not (LN_REDUCE_BATCH and M < WGRAD_SIDE_MIN_ROWS and _defer_ln_reduction(d, M, H, ws, dgamma, dbeta, dbias)):
            reduce()


# Small token counts (the reference's 104-pair micro-batches): the second stage of every LayerNorm backward of a pass goes out as ONE
# launch at the end of the pass (uc2_ln_bwd_reduce_batch) -- 28 launches of 5.7 us on the input-gradient chain otherwise.  The
# pending list belongs to one autograd graph task; entries a failed pass left behind are dropped, not reduced.
LN_REDUCE_BATCH = os.environ.get("UC2_LN_REDUCE_BATCH", "1") != "0"
_LN_BATCH_MAX = 32
_ln_pending = []
_ln_pending_task = [-1]


class _LnReduceItem(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("ws", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p)]


def _defer_ln_reduction(d, M, H, ws, dgamma, dbeta, dbias):
    """queue the reduction for the end of the current backward pass; False outside one (the caller reduces right away)"""
    task = torch._C._current_graph_task_id()
    if task < 0 or torch.cuda.is_current_stream_capturing():
        return False
    if _ln_pending_task[0] != task:
        del _ln_pending[:]                       # (left by a pass that raised)
        try:
            _queue_pass_callback(flush_ln_reductions)
        except RuntimeError:
            return False
        _ln_pending_task[0] = task
    _ln_pending.append((d, H, M, ws, dgamma, dbeta, dbias))
    if len(_ln_pending) >= _LN_BATCH_MAX:
        _flush_ln(keep_task=True)
    return True


def _flush_ln(keep_task):
    items, _ln_pending[:] = list(_ln_pending), []
    if not keep_task:
        _ln_pending_task[0] = -1
    groups = {}
    for it in items:
        groups.setdefault(it[:2], []).append(it)
    for (d, H), its in groups.items():
        arr = (_LnReduceItem * len(its))(*[_LnReduceItem(M, ptr(ws), ptr(dg), ptr(dbt), ptr(dbs)) for (_, _, M, ws, dg, dbt, dbs) in its])
        call("uc2_ln_bwd_reduce_batch", d, len(its), arr, H, stream())


def flush_ln_reductions(end_of_pass=True):
    """reduce every pending LayerNorm backward now: the end-of-backward callback, and (end_of_pass False) BertLayerFn.backward
    before it hands a layer's gradients to GradSync's all-reduce hook"""
    if _ln_pending:
        _flush_ln(keep_task=not end_of_pass)
    elif end_of_pass:
        _ln_pending_task[0] = -1


ATTN_QKV_INTERLEAVED = 16          # include/uc2_hip.h UC2_ATTN_QKV_INTERLEAVED, OR-ed into `impl`


def _attn_q(q_key, qkv, impl, ilv):
    """the delayed-scaling history of the tensor role `q_key` if the attention kernel may write the e4m3 copy itself"""
    if q_key is None or not FP8_ATTN_FUSED or qkv.dtype != torch.bfloat16 or ilv or (ATTN_IMPL if impl is None else impl) == 1:
        return None
    return _fp8_hist_for(q_key, qkv.device)


def attn_fwd(qkv, mask2d, B, L, nh, D, drop_p=0.0, seed=None, seed_imm=0, impl=None, want_lse=True, ilv=False, q_key=None):
    """ilv: qkv is [B L, nh, 3, D] (q|k|v of a head adjacent per token) instead of [B L, 3, nh, D].
    q_key (fp8 mode): the tensor role of ctx at the GEMM that reads it -- returns a third value, (ctx8, scale) written by the same
    kernel (uc2_attn_fwd_q, delayed scaling) or None when that role has no history yet / the MFMA kernels do not take the shape"""
    H = nh * D
    ctx = torch.empty((B * L, H), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, nh, L), dtype=torch.float32, device=qkv.device) if want_lse else None
    h = _attn_q(q_key, qkv, impl, ilv)
    if h is not None:
        c8 = torch.empty((B * L, H), dtype=torch.uint8, device=qkv.device)
        scale = _fp8_cell(qkv.device)[1]
        i_was = h[1]
        prev, nxt, clr = _fp8_rotate(h)
        with _Timed("attn_fwd", B * L * H * qkv.element_size() * 4 + B * nh * L * 4 + B * L * H):
            rc = _lib.load().uc2_attn_fwd_q(B, L, nh, D, ptr(qkv), ptr(mask2d), 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(lse),
                                            ptr(c8), prev, nxt, clr, ptr(scale), stream())
        if rc == 0:
            return ctx, lse, (c8, scale)
        h[1] = i_was
        if rc != -2:
            _lib.check(rc)
    with _Timed("attn_fwd", B * L * H * qkv.element_size() * 4 + B * nh * L * 4):       # q,k,v in; ctx, lse out
        call("uc2_attn_fwd", dt(qkv.dtype), (ATTN_IMPL if impl is None else impl) | (ATTN_QKV_INTERLEAVED if ilv else 0), B, L, nh, D, ptr(qkv), ptr(mask2d),
             1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(lse), stream())
    return (ctx, lse, None) if q_key is not None else (ctx, lse)


def attn_bwd(qkv, mask2d, ctx, dctx, lse, B, L, nh, D, drop_p=0.0, seed=None, seed_imm=0, impl=None, dbias=None, ilv=False, q_key=None):
    """dqkv; with dbias (fp32 [3H]) also dbias += column sums of dqkv = the gradient of the fused q|k|v bias (always in the
    reference order q | k | v).  ilv: qkv and dqkv are in the head-interleaved layout (see attn_fwd).
    q_key (fp8 mode): returns (dqkv, (dqkv8, scale) or None), the e4m3 copy written by the same kernel (uc2_attn_bwd_q)"""
    dqkv = torch.empty_like(qkv)
    h = _attn_q(q_key, qkv, impl, ilv)
    if h is not None:
        d8 = torch.empty(qkv.shape, dtype=torch.uint8, device=qkv.device)
        scale = _fp8_cell(qkv.device)[1]
        i_was = h[1]
        prev, nxt, clr = _fp8_rotate(h)
        with _Timed("attn_bwd", B * L * nh * D * qkv.element_size() * 8 + B * nh * L * 4 + B * L * nh * D * 3):
            rc = _lib.load().uc2_attn_bwd_q(B, L, nh, D, ptr(qkv), ptr(mask2d), 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx),
                                            ptr(lse), ptr(dqkv), ptr(dbias), ptr(_gemm_queue(qkv.device)[12:14]) if GEMM_QUEUE else None,
                                            ptr(d8), prev, nxt, clr, ptr(scale), stream())
        if rc == 0:
            return dqkv, (d8, scale)
        h[1] = i_was
        if rc != -2:
            _lib.check(rc)
    impl = (ATTN_IMPL if impl is None else impl) | (ATTN_QKV_INTERLEAVED if ilv else 0)
    with _Timed("attn_bwd", B * L * nh * D * qkv.element_size() * 8 + B * nh * L * 4):   # qkv, ctx, dctx, lse in; dqkv out
        if GEMM_QUEUE and qkv.dtype == torch.bfloat16:       # N > 1: the persistent kernels share the chip with the all-reduce kernels
            call("uc2_attn_bwd_queued", dt(qkv.dtype), impl, B, L, nh, D, ptr(qkv), ptr(mask2d),
                 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias),
                 ptr(_gemm_queue(qkv.device)[12:14]), stream())
        else:
            call("uc2_attn_bwd", dt(qkv.dtype), impl, B, L, nh, D, ptr(qkv), ptr(mask2d),
                 1.0 / math.sqrt(D), drop_p, ptr(seed), seed_imm, ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dbias), stream())
    return (dqkv, None) if q_key is not None else dqkv


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("uc2_cast", dt(x.dtype), dt(dtype), x.numel(), ptr(x), ptr(out), stream())
    return out


def _mask2d(attention_mask, B, L):
    """[B,1,1,L] additive float mask (model/model.py:433-436) -> contiguous fp32 [B, L]"""
    m = attention_mask
    if m.dim() == 4:
        if m.shape[1] != 1 or m.shape[2] != 1:
            raise _lib.Uc2Error("only key masks of shape [B,1,1,L] are supported, got %s" % (tuple(m.shape),))
        m = m.reshape(m.shape[0], m.shape[3])
    if m.shape != (B, L):
        raise _lib.Uc2Error("attention mask shape %s does not match hidden states [%d,%d]" % (tuple(m.shape), B, L))
    if m.dtype != torch.float32:
        m = m.float()
    return m.contiguous()


# --------------------------------------------------------------------------------------
# one BertLayer = one autograd node (reference model/layer.py:159-170)
# --------------------------------------------------------------------------------------
_P_NAMES = ("qw", "qb", "kw", "kb", "vw", "vb", "ow", "ob", "g1", "b1", "iw", "ib", "fw", "fb", "g2", "b2")


def layer_params(layer):
    a, it, o = layer.attention, layer.intermediate, layer.output
    s = a.self
    return (s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
            a.output.dense.weight, a.output.dense.bias, a.output.LayerNorm.weight, a.output.LayerNorm.bias,
            it.dense.weight, it.dense.bias, o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)


QKV_INTERLEAVED = os.environ.get("UC2_QKV_ILV", "1") != "0"     # head-interleaved q|k|v activations from QKV_ILV_MIN_ROWS tokens
QKV_ILV_MIN_ROWS = 16384


def _ilv_wgrad_plan(n_out, n_in, rows, device):
    """(variant, split_k) of the two-stage ping-pong weight-gradient GEMM dW[n_out, n_in] += dY^T X over `rows` tokens, or None if
    that kernel cannot take the shape.  The interleaved route needs this path: its reduction pass is what puts the rows of dWqkv
    back into the parameter arena's order."""
    key = (True, True, n_out, n_in, rows, True)
    v, sp = gemm_plan(torch.bfloat16, True, True, n_out, n_in, rows, True)
    if v in (8, 12) and sp > 1 and _plan_fits((v, sp), key):
        return v, sp
    tiles = (n_out // 256) * (n_in // 256)
    if n_out % 256 or n_in % 256 or tiles == 0:
        return None
    valid = [s_ for s_ in range(2, 129) if s_ * 256 <= rows and _plan_fits((12, s_), key)]
    if not valid:
        return None
    cus = _num_cus(device)
    return 12, min(valid, key=lambda s_: (abs(tiles * s_ - cus), s_))


# One C call per layer and direction (include/uc2_hip.h: uc2_bert_layer_fwd / _bwd) for the plain route -- no fp8, no head-interleaved
# q|k|v, no fused dropout-residual tails, no k-contiguous W^T copies, i.e. the reference's micro-batch sizes: the same kernels with
# the same arguments in the same order as the per-kernel calls of BertLayerFn below, so the same bits; what it saves is host time
# (~20 ctypes calls, their argument marshalling and the timing hooks per layer).
NATIVE_LAYER = os.environ.get("UC2_NATIVE_LAYER", "1") != "0"


class _GemmPlanC(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int), ("split_k", ctypes.c_int), ("flags", ctypes.c_int)]


_VP, _CI, _CF, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64


class _BertLayerC(ctypes.Structure):            # Uc2BertLayer
    _fields_ = ([(n, _CI) for n in ("dtype", "B", "L", "H", "nh", "I", "attn_impl")] + [(n, _CF) for n in ("eps", "p_hidden", "p_attn")]
                + [("seed", _VP)] + [(n, _U64) for n in ("site_attn", "site_ln1", "site_ln2")]
                + [(n, _VP) for n in ("wqkv", "wo", "wi", "wf", "bqkv", "bo", "g1", "b1", "bi", "bf", "g2", "b2", "mask", "x", "qkv", "ctx",
                                      "lse", "o1", "mean1", "rstd1", "a", "pre", "u", "o2", "mean2", "rstd2", "y")]
                + [(n, _GemmPlanC) for n in ("plan_qkv", "plan_o", "plan_i", "plan_f")] + [("queue", _VP)])


class _BertLayerGradC(ctypes.Structure):        # Uc2BertLayerGrad
    _fields_ = ([(n, _VP) for n in ("dy", "d_o2", "dz2", "d_pre", "da", "d_o1", "dz1", "dctx", "dqkv", "dx", "ws1", "ws2", "dbi", "dbqkv",
                                    "attn_queue")]
                + [(n, _GemmPlanC) for n in ("plan_df", "plan_di", "plan_do", "plan_dqkv")])


def _plan_c(dtype, tb, M, N, K, epi=EPI_NONE, flags=0):
    """the (variant, split_k, flags) _gemm_planned + gemm would pass for this GEMM"""
    v, sp = gemm_plan(dtype, False, tb, M, N, K, False)
    if PP_SKEW and v in (8, 9, 12):
        flags |= (PP_SKEW.get(epi, 0) & 15) << 4
    return _GemmPlanC(v, sp, flags | _EXTRA_FLAGS)


def _native_layer_ok(dtype, M, fp8, ilv):
    return (NATIVE_LAYER and not fp8 and ilv is None and GEMM_TIMER is None and HBM_TIMER is None and _FORCED[0] is None
            and not (LN_FUSE and dtype == torch.bfloat16 and M >= LN_FUSE_MIN_ROWS)
            and not (DGRAD_TRANSPOSED_W and dtype == torch.bfloat16 and M >= DGRAD_WT_MIN_ROWS))


class BertLayerFn(torch.autograd.Function):
    """x -> LN(x + Wo.Attn(x)) -> LN(a + W2.gelu(W1.a)).  10 kernel launches forward, 21 backward.
    Weight/bias/LN gradients are accumulated by the kernels directly into the fp32 gradient arena
    (installed as .grad); autograd only carries dx."""

    @staticmethod
    def forward(ctx, x, mask2d, layer, cfg, *params):
        st = store_of(layer)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        P = dict(zip(_P_NAMES, params))
        B, L, H = x.shape
        nh = cfg["nh"]
        D = H // nh
        M = B * L
        x2 = x.reshape(M, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        training = cfg["training"]
        p_h = cfg["p_hidden"] if training else 0.0
        p_a = cfg["p_attn"] if training else 0.0
        seed = rng.snapshot(x.device) if (p_h > 0 or p_a > 0) else None
        sid = cfg["layer_id"] * 16
        s_attn, s_ln1, s_ln2 = rng.site(sid + 1), rng.site(sid + 2), rng.site(sid + 3)      # dropout sites of this layer

        wqkv = st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype)
        bqkv = st.span(st.data, P["qb"], P["vb"], (3 * H,))
        # Head-interleaved q|k|v (round 4): the QKV GEMM runs on a row-permuted copy of [Wq; Wk; Wv] (store.qkv_interleaved), so a
        # head's q | k | v is ONE 384-byte segment per token instead of three 128-byte segments 1536 bytes apart -- the attention
        # kernels' access pattern is what bounds them (forward 4.2 -> 4.7 TB/s).  Same dot products in the same order: qkv, dqkv,
        # ctx and every gradient are bit-identical to the plain layout.  Needs the MFMA attention kernels and, for the backward,
        # the two-stage ping-pong weight-gradient GEMM (its reduction pass un-permutes the rows of dWqkv).
        ilv = None
        if (QKV_INTERLEAVED and dtype == torch.bfloat16 and M >= QKV_ILV_MIN_ROWS and M % 128 == 0 and not cfg.get("fp8")
                and ATTN_IMPL in (0, 2) and D in (32, 64) and _lib.load().uc2_attn_mfma_supported(L, D)
                and not torch.cuda.is_current_stream_capturing()):
            plan = _ilv_wgrad_plan(3 * H, H, M, x.device) if any(ctx.needs_input_grad) or training else (12, 2)
            pack = st.qkv_interleaved(P["qw"], P["vw"], P["qb"], nh) if plan is not None else None
            if pack is not None:
                ilv = (pack, plan)
                wqkv, bqkv = pack[0], pack[2]
        if not any(ctx.needs_input_grad):
            # forward-only (retrieval scoring, validation, the hard-negative scoring pass): nothing is kept for a
            # backward -- no gelu' stream out of the FFN1 GEMM, no LayerNorm statistics, no log-sum-exp
            qkv = linear_fwd(x2, wqkv, bqkv)
            ctxv, _ = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, want_lse=False, ilv=ilv is not None)
            del qkv
            fuse = int(LN_FUSE) if (dtype == torch.bfloat16 and M >= LN_FUSE_MIN_ROWS) else 0
            o1 = linear_drop_residual(ctxv, st.compute(P["ow"], dtype), P["ob"].data, x2, p_h, seed, s_ln1) if fuse & 1 else None
            if o1 is not None:
                a, _, _ = ln_fwd(o1, None, P["g1"].data, P["b1"].data, 1e-12, want_stats=False)
            else:
                o1 = linear_fwd(ctxv, st.compute(P["ow"], dtype), P["ob"].data)
                a, _, _ = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1, want_stats=False)
            del o1, ctxv
            u = linear_fwd(a, st.compute(P["iw"], dtype), P["ib"].data, EPI_GELU, None)
            o2 = linear_drop_residual(u, st.compute(P["fw"], dtype), P["fb"].data, a, p_h, seed, s_ln2) if fuse & 2 else None
            if o2 is not None:
                y, _, _ = ln_fwd(o2, None, P["g2"].data, P["b2"].data, 1e-12, want_stats=False)
            else:
                o2 = linear_fwd(u, st.compute(P["fw"], dtype), P["fb"].data)
                y, _, _ = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2, want_stats=False)
            del u
            return y.view(B, L, H)
        fp8 = bool(cfg.get("fp8")) and dtype == torch.bfloat16 and H % 128 == 0 and P["iw"].shape[0] % 128 == 0
        I_ = P["iw"].shape[0]
        pre = torch.empty((M, I_), dtype=dtype, device=x.device)
        native = _native_layer_ok(dtype, M, fp8, ilv)
        if native:
            dev = x.device
            e = lambda *shape, dt_=dtype: torch.empty(shape, dtype=dt_, device=dev)
            f32 = torch.float32
            qkv, ctxv, lse = e(M, 3 * H), e(M, H), e(B, nh, L, dt_=f32)
            o1, mean1, rstd1, a = e(M, H), e(M, dt_=f32), e(M, dt_=f32), e(M, H)
            u, o2, mean2, rstd2, y = e(M, I_), e(M, H), e(M, dt_=f32), e(M, dt_=f32), e(M, H)
            c = _BertLayerC()
            c.dtype, c.B, c.L, c.H, c.nh, c.I, c.attn_impl = dt(dtype), B, L, H, nh, I_, ATTN_IMPL
            c.eps, c.p_hidden, c.p_attn = 1e-12, p_h, p_a
            c.seed, c.site_attn, c.site_ln1, c.site_ln2 = ptr(seed), s_attn, s_ln1, s_ln2
            c.wqkv, c.wo, c.wi, c.wf = wqkv.data_ptr(), st.compute(P["ow"], dtype).data_ptr(), st.compute(P["iw"], dtype).data_ptr(), st.compute(P["fw"], dtype).data_ptr()
            c.bqkv, c.bo, c.g1, c.b1 = bqkv.data_ptr(), P["ob"].data.data_ptr(), P["g1"].data.data_ptr(), P["b1"].data.data_ptr()
            c.bi, c.bf, c.g2, c.b2 = P["ib"].data.data_ptr(), P["fb"].data.data_ptr(), P["g2"].data.data_ptr(), P["b2"].data.data_ptr()
            c.mask, c.x = mask2d.data_ptr(), x2.data_ptr()
            c.qkv, c.ctx, c.lse, c.o1, c.mean1, c.rstd1, c.a = (qkv.data_ptr(), ctxv.data_ptr(), lse.data_ptr(), o1.data_ptr(), mean1.data_ptr(),
                                                                 rstd1.data_ptr(), a.data_ptr())
            c.pre, c.u, c.o2, c.mean2, c.rstd2, c.y = pre.data_ptr(), u.data_ptr(), o2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), y.data_ptr()
            c.plan_qkv, c.plan_o = _plan_c(dtype, False, M, 3 * H, H), _plan_c(dtype, False, M, H, H)
            c.plan_i, c.plan_f = _plan_c(dtype, False, M, I_, H, EPI_GELU, GEMM_AUX_DERIV), _plan_c(dtype, False, M, H, I_)
            c.queue = ptr(_gemm_queue(dev)) if (GEMM_QUEUE and dtype == torch.bfloat16) else None
            _lib.check(_lib.load().uc2_bert_layer_fwd(ctypes.byref(c), stream()))
            ctx.native_c = c
            fused1 = fused2 = False
        elif fp8:
            # e4m3 operands for the four forward GEMMs (per-tensor scales computed on the device), bf16 outputs
            # tensor roles (keys of the delayed-scaling histories): a role is named by its CONSUMER; the layer input's by the layer id,
            # so that the layer above can write the e4m3 copy from its last LayerNorm (handed over through _FP8_PREQ)
            kx = (_st_uid(st), ("layer", cfg["layer_id"]), "fwd", "x", FP8_TAG)
            ky = (_st_uid(st), ("layer", cfg["layer_id"] + 1), "fwd", "x", FP8_TAG)
            ka = (_st_uid(st), st.offsets[id(P["iw"])], "fwd", "a", FP8_TAG)
            xq = _FP8_PREQ.pop(x2.data_ptr(), None)
            if xq is not None:                         # written for THIS layer (id) by the layer below, same shape -- or not used
                xq = xq[1] if (xq[0] == cfg["layer_id"] and tuple(xq[1][0].shape) == (M, H)) else None
            w8_, sw_ = _fp8_weight(st, P["qw"], P["vw"], (3 * H, H), False)
            x8_, sx_ = xq if xq is not None else fp8_quantize_act(x2, kx)
            qkv = gemm_fp8(x8_, sx_, w8_, sw_, bias=bqkv)
            # (the attention kernel writes the e4m3 copy of ctx the output projection reads; same role key as the stand-alone pass)
            ctxv, lse, cq = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, q_key=(_st_uid(st), st.offsets[id(P["ow"])], "fwd", "ctx", FP8_TAG))
            o1 = linear_fwd_fp8(ctxv, st, P["ow"], P["ow"], (H, H), P["ob"].data, role="ctx", tag=FP8_TAG, pre_q=cq)
            a, mean1, rstd1, aq = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1, q_key=ka)
            # (the FFN1 GEMM's epilogue writes the e4m3 copy of u that FFN2 reads: no quantisation pass over [tokens, 4H])
            u, uq = linear_fwd_fp8(a, st, P["iw"], P["iw"], (I_, H), P["ib"].data, EPI_GELU, pre, flags=GEMM_AUX_DERIV, role="a", tag=FP8_TAG,
                                   pre_q=aq, q_key=(_st_uid(st), st.offsets[id(P["fw"])], "fwd", "u", FP8_TAG))
            o2 = linear_fwd_fp8(u, st, P["fw"], P["fw"], (H, I_), P["fb"].data, role="u", tag=FP8_TAG, pre_q=uq)
        else:
            qkv = linear_fwd(x2, wqkv, bqkv)
            ctxv, lse = attn_fwd(qkv, mask2d, B, L, nh, D, p_a, seed, s_attn, ilv=ilv is not None)
            # The dense -> dropout -> + residual tails: with LN_FUSE the Wo / FFN2 GEMM writes the pre-LayerNorm SUM (dropout mask and
            # residual in its epilogue), the LayerNorm reads one tensor and hashes no mask; o1 / o2 then hold the sums and the
            # backward runs the LayerNorm in its drop_after = 2 form (fused1 / fused2 say which form each tail took)
            fuse = int(LN_FUSE) if M >= LN_FUSE_MIN_ROWS else 0
            o1 = linear_drop_residual(ctxv, st.compute(P["ow"], dtype), P["ob"].data, x2, p_h, seed, s_ln1) if fuse & 1 else None
            fused1 = o1 is not None
            if fused1:
                a, mean1, rstd1 = ln_fwd(o1, None, P["g1"].data, P["b1"].data, 1e-12)
            else:
                o1 = linear_fwd(ctxv, st.compute(P["ow"], dtype), P["ob"].data)
                a, mean1, rstd1 = ln_fwd(o1, x2, P["g1"].data, P["b1"].data, 1e-12, p_h, seed, s_ln1)
            # `pre` holds gelu'(a W1^T + b1), not the pre-activation itself (UC2_GEMM_AUX_DERIV): one more exp2 beside
            # the forward's Phi(x) there, and the backward's dGELU epilogue becomes a plain multiply
            u = linear_fwd(a, st.compute(P["iw"], dtype), P["ib"].data, EPI_GELU, pre, flags=GEMM_AUX_DERIV)
            o2 = linear_drop_residual(u, st.compute(P["fw"], dtype), P["fb"].data, a, p_h, seed, s_ln2) if fuse & 2 else None
            fused2 = o2 is not None
            if not fused2:
                o2 = linear_fwd(u, st.compute(P["fw"], dtype), P["fb"].data)
        if native:
            pass
        elif fp8:
            fused1 = fused2 = False
            y, mean2, rstd2, yq = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2, q_key=ky)
            _FP8_PREQ.clear()                          # (at most one hand-over alive: the last layer's copy has no fp8 consumer)
            if yq is not None:
                _FP8_PREQ[y.data_ptr()] = (cfg["layer_id"] + 1, yq)
        elif fused2:
            y, mean2, rstd2 = ln_fwd(o2, None, P["g2"].data, P["b2"].data, 1e-12)
        else:
            y, mean2, rstd2 = ln_fwd(o2, a, P["g2"].data, P["b2"].data, 1e-12, p_h, seed, s_ln2)
        ctx.ln_fused = (fused1, fused2)

        ctx.save_for_backward(x2, mask2d, qkv, ctxv, lse, o1, mean1, rstd1, a, pre, u, o2, mean2, rstd2, seed)
        ctx.layer, ctx.cfg, ctx.shape, ctx.p = layer, cfg, (B, L, H, nh, D), (p_h, p_a, (s_attn, s_ln1, s_ln2))
        ctx.params, ctx.fp8, ctx.fp8_tag = params, fp8, FP8_TAG
        ctx.native = native
        ctx.ilv_plan = ilv[1] if ilv is not None else None
        return y.view(B, L, H)

    @staticmethod
    def backward(ctx, dy):
        x2, mask2d, qkv, ctxv, lse, o1, mean1, rstd1, a, pre, u, o2, mean2, rstd2, seed = ctx.saved_tensors
        B, L, H, nh, D = ctx.shape
        p_h, p_a, (s_attn, s_ln1, s_ln2) = ctx.p
        P = dict(zip(_P_NAMES, ctx.params))
        st = store_of(ctx.layer)
        dtype = x2.dtype
        M = B * L
        dy2 = dy.reshape(M, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        G = st.grad_buf

        # LN2 and FFN
        fp8 = ctx.fp8
        I_ = P["iw"].shape[0]
        nat = None
        if ctx.native:
            # everything of this layer's backward except the weight gradients and the LayerNorm reductions: one C call
            dev = x2.device
            e = lambda *shape: torch.empty(shape, dtype=dtype, device=dev)
            lib = _lib.load()
            nws = lib.uc2_ln_bwd_workspace(M, H) // 4
            ws1, ws2 = torch.empty(nws, dtype=torch.float32, device=dev), torch.empty(nws, dtype=torch.float32, device=dev)
            d_o2, d_pre, da, d_o1, dctx, dqkv = e(M, H), e(M, I_), e(M, H), e(M, H), e(M, H), e(M, 3 * H)
            dz2, dz1 = (e(M, H), e(M, H)) if p_h > 0.0 else (None, None)
            dxn = e(M, H) if ctx.needs_input_grad[0] else None
            g = _BertLayerGradC()
            g.dy, g.d_o2, g.dz2, g.d_pre, g.da = dy2.data_ptr(), d_o2.data_ptr(), ptr(dz2), d_pre.data_ptr(), da.data_ptr()
            g.d_o1, g.dz1, g.dctx, g.dqkv, g.dx = d_o1.data_ptr(), ptr(dz1), dctx.data_ptr(), dqkv.data_ptr(), ptr(dxn)
            g.ws1, g.ws2 = ws1.data_ptr(), ws2.data_ptr()
            g.dbi = G(P["ib"]).data_ptr()
            g.dbqkv = st.grad_span(P["qb"], P["vb"], (3 * H,)).data_ptr()
            g.attn_queue = ptr(_gemm_queue(dev)[12:14]) if (GEMM_QUEUE and dtype == torch.bfloat16) else None
            g.plan_df = _plan_c(dtype, True, M, I_, H, EPI_DGELU, GEMM_AUX_DERIV)
            g.plan_di, g.plan_do, g.plan_dqkv = _plan_c(dtype, True, M, H, I_, EPI_ADD), _plan_c(dtype, True, M, H, H), _plan_c(dtype, True, M, H, 3 * H, EPI_ADD)
            c = ctx.native_c
            # (the parameter pointers are taken again here, like the per-kernel route does: a store re-created between forward and
            #  backward -- set_compute_dtype, load_state_dict into a new arena -- must not leave this call with stale addresses)
            c.wqkv = st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype).data_ptr()
            c.wo, c.wi, c.wf = st.compute(P["ow"], dtype).data_ptr(), st.compute(P["iw"], dtype).data_ptr(), st.compute(P["fw"], dtype).data_ptr()
            c.g1, c.g2 = P["g1"].data.data_ptr(), P["g2"].data.data_ptr()
            _lib.check(lib.uc2_bert_layer_bwd(ctypes.byref(c), ctypes.byref(g), stream()))
            d_ = dt(dtype)
            _ln_bwd_second_stage(d_, M, H, ws2, G(P["g2"]), G(P["b2"]), G(P["fb"]), dev)
            _ln_bwd_second_stage(d_, M, H, ws1, G(P["g1"]), G(P["b1"]), G(P["ob"]), dev)
            nat = (d_o2, d_pre, d_o1, dqkv, dxn)
        if nat is not None:
            pass
        elif fp8:
            d_o2, dz2, dq2 = ln_bwd(dy2, o2, a, P["g2"].data, mean2, rstd2, G(P["g2"]), G(P["b2"]), p_h, seed, s_ln2, dbias=G(P["fb"]),
                                    q_key=(_st_uid(st), st.offsets[id(P["fw"])], "bwd", "d_o2", ctx.fp8_tag))
        else:
            d_o2, dz2 = ln_bwd(dy2, o2, None if ctx.ln_fused[1] else a, P["g2"].data, mean2, rstd2, G(P["g2"]), G(P["b2"]), p_h, seed, s_ln2,
                               dbias=G(P["fb"]), drop_after=2 if ctx.ln_fused[1] else False)
        I_ = P["iw"].shape[0]
        # k-contiguous copies W^T for the input-gradient GEMMs (bf16; refreshed once per optimizer step, one launch for all)
        # (from DGRAD_WT_MIN_ROWS tokens)
        use_wt = DGRAD_TRANSPOSED_W and dtype == torch.bfloat16 and not fp8 and M >= DGRAD_WT_MIN_ROWS
        WT = (lambda pf, pl=None, shp=None: st.compute_t(pf, pl, shp)) if use_wt else (lambda *a_: None)
        # small token counts: the four weight gradients go out as ONE grouped launch at the end (wgrad_group)
        grouped = [] if (WGRAD_GROUP and dtype == torch.bfloat16 and M < WGRAD_SIDE_MIN_ROWS and M % 128 == 0) else None
        wgrad = (lambda dyv, xv, dwv: grouped.append((dyv, xv, dwv))) if grouped is not None else (lambda dyv, xv, dwv: linear_wgrad(dyv, xv, dwv, None))
        wgrad(d_o2, u, G(P["fw"]))
        if nat is not None:
            pass
        elif fp8:
            d_pre, dq = linear_dgrad_fp8(d_o2, st, P["fw"], P["fw"], (H, I_), EPI_DGELU, pre, colsum_out=G(P["ib"]), flags=GEMM_AUX_DERIV, role="d_o2",
                                         tag=ctx.fp8_tag, pre_q=dq2, q_key=(_st_uid(st), st.offsets[id(P["iw"])], "bwd", "d_pre", ctx.fp8_tag))
        else:
            d_pre = linear_dgrad(d_o2, st.compute(P["fw"], dtype), EPI_DGELU, pre, colsum_out=G(P["ib"]),
                                 flags=GEMM_AUX_DERIV, wt=WT(P["fw"]))                         # + d(intermediate bias)
        wgrad(d_pre, a, G(P["iw"]))
        if nat is not None:
            pass
        elif fp8:
            da = linear_dgrad_fp8(d_pre, st, P["iw"], P["iw"], (I_, H), EPI_ADD, dz2, role="d_pre", tag=ctx.fp8_tag, pre_q=dq)
        else:
            da = linear_dgrad(d_pre, st.compute(P["iw"], dtype), EPI_ADD, dz2, wt=WT(P["iw"]))
        # LN1, output projection, attention, fused QKV
        if nat is not None:
            pass
        elif fp8:
            d_o1, dz1, dq1 = ln_bwd(da, o1, x2, P["g1"].data, mean1, rstd1, G(P["g1"]), G(P["b1"]), p_h, seed, s_ln1, dbias=G(P["ob"]),
                                    q_key=(_st_uid(st), st.offsets[id(P["ow"])], "bwd", "d_o1", ctx.fp8_tag))
        else:
            d_o1, dz1 = ln_bwd(da, o1, None if ctx.ln_fused[0] else x2, P["g1"].data, mean1, rstd1, G(P["g1"]), G(P["b1"]), p_h, seed, s_ln1,
                               dbias=G(P["ob"]), drop_after=2 if ctx.ln_fused[0] else False)
            dq1 = None
        wgrad(d_o1, ctxv, G(P["ow"]))
        dwqkv = st.grad_span(P["qw"], P["vw"], (3 * H, H))
        if nat is not None:
            wgrad(dqkv, x2, dwqkv)
            return BertLayerFn._finish_backward(ctx, grouped, dy2, None if dxn is None else dxn.view(B, L, H))
        dctx = linear_dgrad_fp8(d_o1, st, P["ow"], P["ow"], (H, H), role="d_o1", tag=ctx.fp8_tag, pre_q=dq1) if fp8 else linear_dgrad(d_o1, st.compute(P["ow"], dtype), wt=WT(P["ow"]))
        dwqkv = st.grad_span(P["qw"], P["vw"], (3 * H, H))
        dbqkv = st.grad_span(P["qb"], P["vb"], (3 * H,))
        # d(q|k|v bias) comes out of attn_bwd: column sums of the dQ/dK/dV accumulators, added up per workgroup in LDS and
        # flushed with one global atomic per column and workgroup (the separate column-sum pass re-read dqkv: 97 us)
        # (the fp32-math kernels run the column-sum pass inside uc2_attn_bwd)
        ilv_plan = ctx.ilv_plan
        dqq = None
        if fp8:
            dqkv, dqq = attn_bwd(qkv, mask2d, ctxv, dctx, lse, B, L, nh, D, p_a, seed, s_attn, dbias=dbqkv,
                                 q_key=(_st_uid(st), st.offsets[id(P["qw"])], "bwd", "dqkv", ctx.fp8_tag))
        else:
            dqkv = attn_bwd(qkv, mask2d, ctxv, dctx, lse, B, L, nh, D, p_a, seed, s_attn, dbias=dbqkv, ilv=ilv_plan is not None)
        if ilv_plan is not None:
            # dqkv is head-interleaved: dWqkv comes out with its rows in that order, the split-K reduction puts them back
            # (dqkv^T x on the two-stage ping-pong kernel: _ilv_wgrad_plan); the input gradient contracts over the interleaved
            # index on both operands (W' / W'^T)
            v_, sp_ = ilv_plan
            w_ilv, wt_ilv, _ = st.qkv_interleaved(P["qw"], P["vw"], P["qb"], nh)

            def _wg():
                gemm(dqkv, x2, 3 * H, H, M, ta=True, tb=True, out=dwqkv, accumulate=True, split_k=sp_, variant=v_, qkv_rows_d=D)
            if _side_route(M):
                _on_side_stream(dqkv.device, _wg, (dqkv, x2))
            else:
                _wg()
            dx = None
            if ctx.needs_input_grad[0]:
                dx = linear_dgrad(dqkv, w_ilv, EPI_ADD, dz1, wt=wt_ilv if use_wt else None).view(B, L, H)
        else:
            wgrad(dqkv, x2, dwqkv)
            dx = None
            if ctx.needs_input_grad[0]:
                if fp8:
                    dx = linear_dgrad_fp8(dqkv, st, P["qw"], P["vw"], (3 * H, H), EPI_ADD, dz1, role="dqkv", tag=ctx.fp8_tag, pre_q=dqq).view(B, L, H)
                else:
                    dx = linear_dgrad(dqkv, st.compute_span(P["qw"], P["vw"], (3 * H, H), dtype), EPI_ADD, dz1,
                                      wt=WT(P["qw"], P["vw"], (3 * H, H))).view(B, L, H)
        return BertLayerFn._finish_backward(ctx, grouped, dy2, dx)

    @staticmethod
    def _finish_backward(ctx, grouped, dy2, dx):
        """the layer's grouped weight-gradient launch and the gradient-ready hook (both routes of backward end here)"""
        if grouped:
            if WGRAD_GROUP_SIDE and WGRAD_SIDE_STREAM and not torch.cuda.is_current_stream_capturing():
                # the layer's grouped weight-gradient launch beside the next layer's backward (small token counts: the main chain's
                # kernels leave CUs idle at their round tails and between launches)
                trip = list(grouped)
                _on_side_stream(dy2.device, lambda: wgrad_group(trip), [t for tr in trip for t in tr[:2]])
            else:
                wgrad_group(grouped)
        hook = ctx.cfg.get("grad_ready_hook")
        if hook is not None:
            # this layer's gradients are all enqueued: the main stream holds the bias / LayerNorm gradients, the side stream the
            # four dW GEMMs.  The hook's all-reduce waits for both streams itself (ops.pending_side_stream ->
            # uc2_comm_allreduce_bucket_after); joining the side stream into the main stream here (round 3) serialised every
            # layer boundary of the main stream behind that layer's weight-gradient GEMMs exactly when N > 1.
            # (small batches: the layer's LayerNorm sums are still pending -- but only flush them per layer when the hook is about to
            #  start a reduction: with an unarmed GradSync, or one rank and no communicator, that would undo the one-launch batching
            #  of every non-final micro-step)
            owner = getattr(hook, "__self__", None)
            will = getattr(owner, "will_reduce", None)
            if will is None or will():
                flush_ln_reductions(end_of_pass=False)
            hook(ctx.layer)
        return (dx, None, None, None) + (None,) * len(ctx.params)


# --------------------------------------------------------------------------------------
# generic building blocks for embeddings and heads
# --------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); `owner` is the module that owns weight/bias (for the store lookup).
    act: EPI_NONE | EPI_GELU | EPI_TANH.  weight_t=True means `weight` is stored [in, out] and used
    transposed (RegionFeatureRegression: F.linear(h, W_img^T), model/model.py:1155)."""

    @staticmethod
    def forward(ctx, x, owner, act, weight_t, weight, bias, rows=None):
        """rows = (r0, r1): use only output rows r0..r1-1 of weight / bias (the q, k or v third of a packed in_proj)"""
        st = store_of(owner)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = st.compute(weight, dtype)
        b = bias.data if bias is not None else None
        if rows is not None:
            assert not weight_t
            w = w[rows[0]:rows[1]]
            b = b[rows[0]:rows[1]] if b is not None else None
        M, K = x2.shape
        N = w.shape[1] if weight_t else w.shape[0]
        pre = torch.empty((M, N), dtype=dtype, device=x.device) if act == EPI_GELU else None
        if weight_t:
            y = gemm(x2, w, M, N, K, tb=True, bias=b, epi=act, aux_out=pre)
        else:
            y = gemm(x2, w, M, N, K, bias=b, epi=act, aux_out=pre)
        ctx.save_for_backward(x2, pre if act == EPI_GELU else (y if act == EPI_TANH else None))
        ctx.owner, ctx.act, ctx.weight_t, ctx.wb, ctx.shp, ctx.rows = owner, act, weight_t, (weight, bias), shp, rows
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        weight, bias = ctx.wb
        st = store_of(ctx.owner)
        dtype = x2.dtype
        M, K = x2.shape
        N = dy.shape[-1]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.act == EPI_GELU:                 # dpre = dy * gelu'(pre)
            dpre = _dgelu(dy2, aux)
        elif ctx.act == EPI_TANH:
            dpre = torch.empty_like(dy2)
            call("uc2_dtanh", dt(dtype), dy2.numel(), ptr(aux), ptr(dy2), ptr(dpre), stream())
        else:
            dpre = dy2
        w = st.compute(weight, dtype)
        dw = st.grad_buf(weight)
        db = st.grad_buf(bias) if bias is not None else None
        if ctx.rows is not None:
            r0, r1 = ctx.rows
            w, dw = w[r0:r1], dw[r0:r1]
            db = db[r0:r1] if db is not None else None
        if ctx.weight_t:                        # weight [K_in, N_out]: dW[K,N] += X^T dPre
            gemm(x2, dpre, K, N, M, ta=True, tb=True, out=dw, accumulate=True,
                 split_k=_wgrad_split(dtype, K, N, M))
            if db is not None:
                colsum_accum(dpre, db)
        else:
            linear_wgrad(dpre, x2, dw, db)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.weight_t:                    # dX[M,K] = dPre[M,N] W[K,N]^T
                dx = gemm(dpre, w, M, K, N)
            else:
                dx = linear_dgrad(dpre, w)
            dx = dx.view(ctx.shp)
        # one gradient slot per input actually passed (`rows` is only ever passed as a tuple, never as an explicit None)
        return (dx, None, None, None, None, None) + ((None,) if ctx.rows is not None else ())


def _dgelu(dy2, pre):
    """dy * gelu'(pre), elementwise (head transforms; the encoder FFN fuses this into its dgrad GEMM)"""
    out = torch.empty_like(dy2)
    call("uc2_dgelu", dt(dy2.dtype), dy2.numel(), ptr(pre), ptr(dy2), ptr(out), stream())
    return out


class GeluFn(torch.autograd.Function):
    """x * 0.5 * (1 + erf(x / sqrt 2)) as a stand-alone activation (model/layer.py:31-37)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        call("uc2_gelu", dt(x.dtype), x.numel(), ptr(x), ptr(y), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _dgelu(dy.contiguous(), x)


class FusedQKVFn(torch.autograd.Function):
    """q|k|v = x [Wq;Wk;Wv]^T + [bq;bk;bv] in ONE GEMM over the adjacent arena slices of the three nn.Linear
    parameters (model/layer.py:76-78 runs three); gradients go straight into the matching gradient-arena span"""

    @staticmethod
    def forward(ctx, x, owner, qw, qb, kw, kb, vw, vb):
        st = store_of(owner)
        dtype = x.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        H = qw.shape[1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        wqkv = st.compute_span(qw, vw, (3 * qw.shape[0], H), dtype)
        bqkv = st.span(st.data, qb, vb, (3 * qw.shape[0],))
        qkv = linear_fwd(x2, wqkv, bqkv)
        ctx.save_for_backward(x2)
        ctx.owner, ctx.ps, ctx.shp = owner, (qw, qb, kw, kb, vw, vb), x.shape
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        (x2,) = ctx.saved_tensors
        qw, qb, kw, kb, vw, vb = ctx.ps
        st = store_of(ctx.owner)
        H = qw.shape[1]
        dqkv = dqkv.contiguous()
        linear_wgrad(dqkv, x2, st.grad_span(qw, vw, (3 * qw.shape[0], H)), st.grad_span(qb, vb, (3 * qw.shape[0],)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dqkv, st.compute_span(qw, vw, (3 * qw.shape[0], H), x2.dtype)).view(ctx.shp)
        return (dx,) + (None,) * 7


class TiedSubsetDecoderFn(torch.autograd.Function):
    """logits over a SUBSET of the tied decoder's columns: (z E^T + bias)[:, ids] == z E[ids]^T + bias[ids]
    (forward_mmxlm_soft, model/model.py:639-642, keeps 2857 of 250 002 columns): the full-vocabulary logits are
    never formed; dE rows / dbias entries of the subset are accumulated into the gradient arena (ids unique)."""

    @staticmethod
    def forward(ctx, z, owner, weight, bias, ids):
        st = store_of(owner)
        dtype = z.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        n, H = z.shape
        nv = ids.numel()
        z = z.contiguous()
        wsub = torch.empty((nv, H), dtype=dtype, device=z.device)
        wc = st.compute(weight, dtype)
        call("uc2_select_rows", dt(dtype), nv, H, ptr(wc), wc.stride(0), ptr(ids), ptr(wsub), H, 0, stream())
        bsub = torch.empty(nv, dtype=torch.float32, device=z.device)
        call("uc2_gather_f32", nv, ptr(bias.data), ptr(ids), ptr(bsub), 0, stream())
        y = gemm(z, wsub, n, nv, H, bias=bsub)
        ctx.save_for_backward(z, wsub, ids)
        ctx.owner, ctx.wb = owner, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, wsub, ids = ctx.saved_tensors
        weight, bias = ctx.wb
        st = store_of(ctx.owner)
        n, H = z.shape
        nv = ids.numel()
        dy = dy.contiguous()
        dwsub = torch.zeros((nv, H), dtype=torch.float32, device=z.device)
        gemm(dy, z, nv, H, n, ta=True, tb=True, out=dwsub, accumulate=True)
        dE = st.grad_buf(weight)
        call("uc2_select_rows", 0, nv, H, ptr(dwsub), H, ptr(ids), ptr(dE), dE.stride(0), 2, stream())
        dbsub = torch.zeros(nv, dtype=torch.float32, device=z.device)
        colsum_accum(dy, dbsub)
        call("uc2_gather_f32", nv, ptr(dbsub), ptr(ids), ptr(st.grad_buf(bias)), 1, stream())
        dz = gemm(dy, wsub, n, H, nv, tb=True)
        return dz, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """y = LN(dropout(x) + residual) * gamma + beta   (residual optional); with drop_after the dropout
    sits on the output instead: y = dropout(LN(x + residual) * gamma + beta)"""

    @staticmethod
    def forward(ctx, x, residual, owner, eps, drop_p, seed_imm, gamma, beta, beta_extra, drop_after=False):
        st = store_of(owner)
        shp = x.shape
        H = shp[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, H)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        seed = rng.snapshot(x.device) if drop_p > 0 else None
        seed_imm = rng.site(seed_imm)
        b = beta.data if beta_extra is None else (beta.data + beta_extra.data)
        y, mean, rstd = ln_fwd(x2, r2, gamma.data, b, eps, drop_p, seed, seed_imm, drop_after=drop_after)
        ctx.save_for_backward(x2, r2, mean, rstd, seed)
        ctx.owner, ctx.gb, ctx.cfg, ctx.shp = owner, (gamma, beta, beta_extra), (drop_p, seed_imm, drop_after), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, mean, rstd, seed = ctx.saved_tensors
        gamma, beta, beta_extra = ctx.gb
        drop_p, seed_imm, drop_after = ctx.cfg
        st = store_of(ctx.owner)
        H = x2.shape[1]
        dy2 = dy.reshape(-1, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dbeta = st.grad_buf(beta)
        dx, dres = ln_bwd(dy2, x2, r2, gamma.data, mean, rstd, st.grad_buf(gamma), dbeta, drop_p, seed, seed_imm,
                          need_dres=r2 is not None, drop_after=drop_after)
        dextra = None
        if beta_extra is not None and ctx.needs_input_grad[8]:
            # d(beta + extra) flows to both; beta got it through the arena, extra gets a fresh column sum
            if drop_after and drop_p > 0:
                raise _lib.Uc2Error("beta_extra with output dropout is not supported")
            dextra = torch.zeros(H, dtype=torch.float32, device=dy.device)
            colsum_accum(dy2, dextra)
        return (dx.view(ctx.shp) if ctx.needs_input_grad[0] else None,
                dres.view(ctx.shp) if (r2 is not None and ctx.needs_input_grad[1]) else None,
                None, None, None, None, None, None, dextra, None)


class EmbedTextFn(torch.autograd.Function):
    """word[ids] + pos[pos_ids] + type[type_ids or 0]  (model/model.py:322-330), output in compute dtype"""

    @staticmethod
    def forward(ctx, owner, dtype, ids, pos_ids, type_ids, word, pos, typ, word_pad=-1, pos_pad=-1):
        B, T = ids.shape
        H = word.shape[1]
        out = torch.empty((B, T, H), dtype=dtype, device=ids.device)
        ids_c, pos_c = ids.contiguous(), pos_ids.contiguous()
        ty_c = type_ids.contiguous() if type_ids is not None else None
        call("uc2_embed_fwd", dt(dtype), B * T, H, ptr(ids_c), ptr(pos_c), ptr(ty_c), 0, ptr(word.data), ptr(pos.data),
             ptr(typ.data), ptr(out), stream())
        ctx.save_for_backward(ids_c, pos_c, ty_c)
        ctx.owner, ctx.tabs, ctx.H, ctx.pads = owner, (word, pos, typ), H, (int(word_pad), int(pos_pad))
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, pos_ids, type_ids = ctx.saved_tensors
        word, pos, typ = ctx.tabs
        st = store_of(ctx.owner)
        H = ctx.H
        d2 = dout.reshape(-1, H)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        rows = d2.shape[0]
        dtyp = st.grad_buf(typ)
        # ids are [B, T]: position / type rows repeat down the batch and are summed in registers (uc2_embed_bwd_seq); -2 = shape not
        # taken, the row-per-wave kernel then adds every token's row with atomics
        Bn, Tn = ids.shape
        rc = _lib.load().uc2_embed_bwd_seq(dt(d2.dtype), Bn, Tn, H, ptr(ids), ptr(pos_ids), ptr(type_ids), ptr(d2), ptr(st.grad_buf(word)),
                                           ptr(st.grad_buf(pos)), ptr(dtyp), ctx.pads[0], ctx.pads[1], stream()) if EMBED_BWD_SEQ else -2
        if rc == -2:
            call("uc2_embed_bwd", dt(d2.dtype), rows, H, ptr(ids), ptr(pos_ids), ptr(type_ids), ptr(d2),
                 ptr(st.grad_buf(word)), ptr(st.grad_buf(pos)), ptr(dtyp), ctx.pads[0], ctx.pads[1], stream())
        else:
            _lib.check(rc)
        if type_ids is None:         # constant type 0: its row gets the column sum (no atomic pile-up on one row)
            colsum_accum(d2, dtyp[0])
        return (None,) * 10


EMBED_BWD_SEQ = os.environ.get("UC2_EMBED_BWD_SEQ", "1") != "0"      # per-position embedding backward (uc2_embed_bwd_seq)


class GatherRowsFn(torch.autograd.Function):
    """torch.gather(src, 1, index[..., None].expand(H)) (model/model.py:420-425)"""

    @staticmethod
    def forward(ctx, src, index):
        B, S, H = src.shape
        L = index.shape[1]
        src = src.contiguous()
        idx = index.contiguous()
        out = torch.empty((B, L, H), dtype=src.dtype, device=src.device)
        call("uc2_gather_rows_fwd", dt(src.dtype), B, S, L, H, ptr(src), ptr(idx), ptr(out), stream())
        ctx.save_for_backward(idx)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, L, H = dout.shape
        dout = dout.contiguous()
        dsrc = torch.empty((B, ctx.S, H), dtype=dout.dtype, device=dout.device)
        call("uc2_gather_rows_bwd", dt(dout.dtype), B, ctx.S, L, H, ptr(dout), ptr(idx), ptr(dsrc), stream())
        return dsrc, None


class GatherCatRowsFn(torch.autograd.Function):
    """torch.gather(torch.cat([a, b], 1), 1, index[..., None].expand(H)) (model/model.py:412-425) without the concatenated tensor:
    the gather reads the text and image embeddings in place; the backward writes their two gradients as separate tensors"""

    @staticmethod
    def forward(ctx, a, b, index):
        B, S1, H = a.shape
        S2 = b.shape[1]
        L = index.shape[1]
        a, b = a.contiguous(), b.contiguous()
        idx = index.contiguous()
        out = torch.empty((B, L, H), dtype=a.dtype, device=a.device)
        call("uc2_gather_rows2_fwd", dt(a.dtype), B, S1, S2, L, H, ptr(a), ptr(b), ptr(idx), ptr(out), stream())
        ctx.save_for_backward(idx)
        ctx.S = (S1, S2)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, L, H = dout.shape
        S1, S2 = ctx.S
        dout = dout.contiguous()
        da = torch.empty((B, S1, H), dtype=dout.dtype, device=dout.device)
        db = torch.empty((B, S2, H), dtype=dout.dtype, device=dout.device)
        call("uc2_gather_rows2_bwd", dt(dout.dtype), B, S1, S2, L, H, ptr(dout), ptr(idx), ptr(da), ptr(db), stream())
        return da, db, None


class SelectRowsFn(torch.autograd.Function):
    """hidden[mask] for a boolean mask over rows (model/model.py:653-657); rows = flat row indices"""

    @staticmethod
    def forward(ctx, hidden2, rows):
        R, H = hidden2.shape
        n = rows.numel()
        out = torch.empty((n, H), dtype=hidden2.dtype, device=hidden2.device)
        call("uc2_select_rows", dt(hidden2.dtype), n, H, ptr(hidden2), hidden2.stride(0), ptr(rows), ptr(out), H, 0,
             stream())
        ctx.save_for_backward(rows)
        ctx.R = R
        return out

    @staticmethod
    def backward(ctx, dout):
        (rows,) = ctx.saved_tensors
        n, H = dout.shape
        dout = dout.contiguous()
        dsrc = torch.zeros((ctx.R, H), dtype=dout.dtype, device=dout.device)
        call("uc2_select_rows", dt(dout.dtype), n, H, ptr(dout), H, ptr(rows), ptr(dsrc), H, 1, stream())
        return dsrc, None


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits, labels, ignore_index, reduction='none'); logits are consumed (overwritten
    by dlogits in backward).  Also returns argmax (int64) as a non-differentiable side output."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index, n_valid_cols):
        n = logits.shape[0]
        V = n_valid_cols
        labels = labels.contiguous()
        loss = torch.empty(n, dtype=torch.float32, device=logits.device)
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        am = torch.empty(n, dtype=torch.int64, device=logits.device)
        call("uc2_ce_fwd", dt(logits.dtype), n, V, ptr(logits), logits.stride(0), ptr(labels), ignore_index, ptr(loss),
             ptr(lse), ptr(am), stream())
        ctx.save_for_backward(logits, labels, lse)
        ctx.cfg = (ignore_index, V)
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, gloss, _gam):
        logits, labels, lse = ctx.saved_tensors
        ignore_index, V = ctx.cfg
        n = logits.shape[0]
        g = gloss.contiguous().float()
        dlog = logits            # in place: the logits buffer becomes dlogits
        call("uc2_ce_bwd", dt(logits.dtype), n, V, ptr(dlog), dlog.stride(0), ptr(labels), ignore_index, ptr(lse),
             ptr(g), stream())
        return dlog, None, None, None


class KLDivFn(torch.autograd.Function):
    """F.kl_div(F.log_softmax(pred, -1), target, reduction='none') (model/model.py:764-768)"""

    @staticmethod
    def forward(ctx, pred, target, n_valid_cols):
        n, V = pred.shape[0], n_valid_cols
        target = target.contiguous().float()
        lse = torch.empty(n, dtype=torch.float32, device=pred.device)
        call("uc2_ce_fwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), None, -100, None, ptr(lse), None, stream())
        loss = torch.empty((n, V), dtype=torch.float32, device=pred.device)
        call("uc2_kl_fwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), ptr(target), ptr(lse), ptr(loss), stream())
        ctx.save_for_backward(pred, target, lse)
        ctx.V = V
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, target, lse = ctx.saved_tensors
        n, V = pred.shape[0], ctx.V
        g = gloss.contiguous().float()
        dpred = torch.zeros_like(pred) if pred.shape[1] > V else torch.empty_like(pred)
        call("uc2_kl_bwd", dt(pred.dtype), n, V, ptr(pred), pred.stride(0), ptr(target), ptr(lse), ptr(g), ptr(dpred),
             stream())
        return dpred, None, None


class MSEFn(torch.autograd.Function):
    """F.mse_loss(pred, target, reduction='none') (model/model.py:684-686)"""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        target = target.contiguous().float()
        loss = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
        call("uc2_mse", dt(pred.dtype), pred.numel(), ptr(pred), ptr(target), None, ptr(loss), None, stream())
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, target = ctx.saved_tensors
        g = gloss.contiguous().float()
        dpred = torch.empty_like(pred)
        call("uc2_mse", dt(pred.dtype), pred.numel(), ptr(pred), ptr(target), ptr(g), None, ptr(dpred), stream())
        return dpred, None


class TripletFn(torch.autograd.Function):
    """sigmoid -> view(-1, sample_size) -> clamp(margin + neg - pos, 0) (model/itm.py:45-53)"""

    @staticmethod
    def forward(ctx, scores, sample_size, margin):
        s = scores.contiguous().view(-1)
        n = s.numel() // sample_size
        loss = torch.empty((n, sample_size - 1), dtype=torch.float32, device=s.device)
        call("uc2_triplet", dt(s.dtype), n, sample_size, margin, ptr(s), None, ptr(loss), None, stream())
        ctx.save_for_backward(s)
        ctx.cfg = (n, sample_size, margin, scores.shape)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (s,) = ctx.saved_tensors
        n, ss, margin, shp = ctx.cfg
        g = gloss.contiguous().float()
        ds = torch.empty_like(s)
        call("uc2_triplet", dt(s.dtype), n, ss, margin, ptr(s), ptr(g), None, ptr(ds), stream())
        return ds.view(shp), None, None


def add_rowvec(a, b, vec, rowmask, out_dtype):
    """out = a + b + (rowmask ? vec : 0) row-wise; a may be fp32 while out is the compute dtype"""
    H = a.shape[-1]
    a2 = a.reshape(-1, H)
    if not a2.is_contiguous():
        a2 = a2.contiguous()
    b2 = None
    if b is not None:
        b2 = b.reshape(-1, H)
        if not b2.is_contiguous():
            b2 = b2.contiguous()
        assert b2.dtype == out_dtype
    out = torch.empty(a2.shape, dtype=out_dtype, device=a.device)
    call("uc2_add_rowvec", dt(a2.dtype), dt(out_dtype), a2.shape[0], H, ptr(a2), ptr(b2), ptr(vec), ptr(rowmask),
         ptr(out), stream())
    return out.view(a.shape)


class MaskEmbedFn(torch.autograd.Function):
    """cast(img_feat) + mask_embedding(img_masks) with row 0 == 0 and no gradient to row 0
    (nn.Embedding(2, img_dim, padding_idx=0), model/model.py:347,353-356)"""

    @staticmethod
    def forward(ctx, owner, img_feat, img_masks, weight, out_dtype):
        m8 = img_masks.reshape(-1).to(torch.uint8).contiguous()
        out = add_rowvec(img_feat, None, weight.data[1], m8, out_dtype)
        ctx.save_for_backward(m8)
        ctx.owner, ctx.weight = owner, weight
        return out

    @staticmethod
    def backward(ctx, dout):
        (m8,) = ctx.saved_tensors
        st = store_of(ctx.owner)
        d2 = dout.reshape(-1, dout.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        colsum_accum(d2, st.grad_buf(ctx.weight)[1], m8)
        return None, None, None, None, None


class AddRowFn(torch.autograd.Function):
    """a + b + table[row]  (transformed_im + transformed_pos + type embedding, model/model.py:360)"""

    @staticmethod
    def forward(ctx, a, b, table, row):
        out = add_rowvec(a, b, table.data[row], None, a.dtype)
        ctx.table, ctx.row = table, row
        return out

    @staticmethod
    def backward(ctx, dout):
        st = getattr(ctx.table, "_uc2_store", None)
        d2 = dout.reshape(-1, dout.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        if st is not None:
            colsum_accum(d2, st.grad_buf(ctx.table)[ctx.row])
            dtab = None
        else:
            dtab = torch.zeros_like(ctx.table.data)
            colsum_accum(d2, dtab[ctx.row])
        return dout, dout, dtab, None


_DEC_ROWS = 8192          # rows per decoder chunk: 2 * rows * 250 112 bytes of logits must stay below 2^32 (the ping-pong
                          # kernel addresses its operands with 32-bit byte offsets), and a chunk's logits are 4.1 GB


def _dec_chunks(npad):
    """equal row chunks (multiples of 256, at most _DEC_ROWS): 9216 masked rows are 2 x 4608, not 8192 + 1024 -- the
    remainder chunk ran the vocabulary-long GEMMs on a handful of tiles"""
    if npad <= 0:
        return []
    n = (npad + _DEC_ROWS - 1) // _DEC_ROWS
    rows = ((npad + n - 1) // n + 255) // 256 * 256 if npad % 256 == 0 else _DEC_ROWS
    return [(r0, min(npad, r0 + rows)) for r0 in range(0, npad, rows)]


class DecoderCEFn(torch.autograd.Function):
    """tied-decoder logits + cross entropy in one node (model/layer.py:257-265, model/model.py:590-596).
    The vocabulary tables are padded to whole 256-row GEMM tiles inside the arena (store.padded) and the masked rows
    to a multiple of 256 (zero rows, ignored labels), so the three decoder GEMMs -- logits = z E^T + bias,
    dE += dlogits^T z, dz = dlogits E -- all run on the persistent 256x256 MFMA kernel; logits live in ONE
    [rows, 250112] bf16 buffer per chunk of 8192 rows that the backward overwrites with dlogits, and dE goes straight
    into the word-embedding gradient arena."""

    @staticmethod
    def forward(ctx, z, owner, weight, bias, labels, ignore_index):
        st = store_of(owner)
        dtype = z.dtype
        if dtype == torch.bfloat16:
            st.sync_shadow()
        n, H = z.shape
        V = weight.shape[0]
        in_arena = st.owns(weight) and st.owns(bias)
        Wp = st.padded(st.data if dtype == torch.float32 else st.shadow, weight) if in_arena else st.compute(weight, dtype)
        bp = st.padded(st.data, bias) if in_arena else bias.data
        Vp = Wp.shape[0] if Wp.shape[0] % 8 == 0 else (V + 7) // 8 * 8
        tile = 256 if (dtype == torch.bfloat16 and n >= 256) else 1
        npad = (n + tile - 1) // tile * tile
        z = z.contiguous()
        labels = labels.contiguous()
        if npad != n:                                 # zero rows / ignored labels up to whole tiles
            zp = torch.zeros((npad, H), dtype=dtype, device=z.device)
            zp[:n].copy_(z)
            lp = torch.full((npad,), ignore_index, dtype=labels.dtype, device=z.device)
            lp[:n].copy_(labels)
            z, labels = zp, lp
        loss = torch.empty(npad, dtype=torch.float32, device=z.device)
        lse = torch.empty(npad, dtype=torch.float32, device=z.device)
        am = torch.empty(npad, dtype=torch.int64, device=z.device)
        chunks = []
        for r0, r1 in _dec_chunks(npad):
            m = r1 - r0
            logits = torch.empty((m, Vp), dtype=dtype, device=z.device)
            if Wp.shape[0] == Vp and Vp != V:         # whole padded tiles: N = Vp (padding columns = padding bias = 0)
                _gemm_planned(z[r0:r1], Wp, m, Vp, H, False, False, out=logits, bias=bp)
            else:
                gemm(z[r0:r1], Wp, m, V, H, out=logits, bias=bp)
            call("uc2_ce_fwd", dt(dtype), m, V, ptr(logits), Vp, ptr(labels[r0:r1]), ignore_index, ptr(loss[r0:r1]),
                 ptr(lse[r0:r1]), ptr(am[r0:r1]), stream())
            chunks.append(logits)
        ctx.save_for_backward(z, labels, lse, *chunks)
        ctx.owner, ctx.wb, ctx.cfg = owner, (weight, bias), (ignore_index, V, Vp, n, npad)
        loss, am = loss[:n], am[:n]
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, gloss, _g):
        z, labels, lse = ctx.saved_tensors[:3]
        chunks = ctx.saved_tensors[3:]
        weight, bias = ctx.wb
        ignore_index, V, Vp, n, npad = ctx.cfg
        st = store_of(ctx.owner)
        dtype = z.dtype
        H = z.shape[1]
        g = gloss.contiguous().float()
        if npad != n:
            gp = torch.zeros(npad, dtype=torch.float32, device=g.device)
            gp[:n].copy_(g)
            g = gp
        in_arena = st.owns(weight) and st.owns(bias)
        Wp = st.padded(st.data if dtype == torch.float32 else st.shadow, weight) if in_arena else st.compute(weight, dtype)
        st.grad_buf(weight)
        st.grad_buf(bias)
        dE = st.padded(st.grad, weight) if in_arena else st.grad_buf(weight)
        db = st.padded(st.grad, bias) if in_arena else st.grad_buf(bias)
        full = Wp.shape[0] == Vp and Vp != V
        dz = torch.empty((npad, H), dtype=dtype, device=z.device)
        for ci, (r0, r1) in enumerate(_dec_chunks(npad)):
            m = r1 - r0
            dlog = chunks[ci]                          # in place: the logits buffer becomes dlogits (padding columns zeroed)
            # dlogits in place, with dbias[V] += colsum(dlogits) from the same pass when the rows are vectorisable
            rc = _lib.load().uc2_ce_bwd_colsum(dt(dtype), m, V, ptr(dlog), Vp, ptr(labels[r0:r1]), ignore_index,
                                               ptr(lse[r0:r1]), ptr(g[r0:r1]), ptr(db), Vp if full else V, stream())
            have_db = rc == 0
            if rc not in (0, -2):
                _lib.check(rc)
            if not have_db:
                call("uc2_ce_bwd", dt(dtype), m, V, ptr(dlog), Vp, ptr(labels[r0:r1]), ignore_index, ptr(lse[r0:r1]),
                     ptr(g[r0:r1]), stream())
            # dE[V,H] += dlogits^T z ; dz = dlogits E
            if full:
                _gemm_planned(dlog, z[r0:r1], Vp, H, m, True, True, wgrad=True, out=dE, accumulate=True, lda=Vp)
                if not have_db:
                    call("uc2_colsum_accum", dt(dtype), m, Vp, ptr(dlog), Vp, None, ptr(db), stream())
                if dtype == torch.bfloat16 and m >= 256:
                    # m x H is only (m/256) x 3 tiles (96 at 8192 rows) under a contraction of 250 112: split it over the
                    # vocabulary like a weight gradient (fp32 partial tiles + one reduction pass), then round once
                    dz32 = torch.zeros((m, H), dtype=torch.float32, device=z.device)
                    _gemm_planned(dlog, Wp, m, H, Vp, False, True, wgrad=True, out=dz32, accumulate=True, lda=Vp)
                    call("uc2_cast", dt(torch.float32), dt(dtype), m * H, ptr(dz32), ptr(dz[r0:r1]), stream())
                else:
                    _gemm_planned(dlog, Wp, m, H, Vp, False, True, out=dz[r0:r1], lda=Vp)
            else:
                gemm(dlog, z[r0:r1], V, H, m, ta=True, tb=True, out=dE, accumulate=True, lda=Vp,
                     split_k=_wgrad_split(dtype, V, H, m))
                if not have_db:
                    call("uc2_colsum_accum", dt(dtype), m, V, ptr(dlog), Vp, None, ptr(db), stream())
                gemm(dlog, Wp, m, H, V, tb=True, out=dz[r0:r1], lda=Vp)
        return dz[:n], None, None, None, None, None


class OTDistFn(torch.autograd.Function):
    """optimal_transport_dist of the scattered-back text / image embeddings (model/ot.py:66-82, model/model.py:701-720):
    dist [B] fp32; the transport plan is a constant of the backward (the reference detaches it)"""

    @staticmethod
    def forward(ctx, seq, scatter, txt_pad, img_pad, T, R, beta, iters):
        B, L, H = seq.shape
        seq = seq.contiguous()
        scatter = scatter.contiguous()
        tp = txt_pad.to(torch.uint8).contiguous()
        ip = img_pad.to(torch.uint8).contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.uc2_ot_workspace(B, T, R, H), dtype=torch.uint8, device=seq.device)
        dist = torch.empty(B, dtype=torch.float32, device=seq.device)
        Tm = torch.empty((B, R, T), dtype=torch.float32, device=seq.device)
        call("uc2_ot_fwd", dt(seq.dtype), B, L, T, R, H, ptr(seq), ptr(scatter), ptr(tp), ptr(ip), float(beta), int(iters),
             ptr(dist), ptr(Tm), ptr(ws), stream())
        ctx.save_for_backward(Tm, ws)
        ctx.cfg = (B, L, T, R, H, seq.dtype)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        Tm, ws = ctx.saved_tensors
        B, L, T, R, H, dtype = ctx.cfg
        g = gdist.contiguous().float()
        dseq = torch.zeros((B, L, H), dtype=dtype, device=Tm.device)
        call("uc2_ot_bwd", dt(dtype), B, L, T, R, H, ptr(Tm), ptr(ws), ptr(g), ptr(dseq), stream())
        return dseq, None, None, None, None, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(QK^T/sqrt(d) + mask) V over a packed [B*L, 3H] projection (one node; used by MultiheadAttention)"""

    @staticmethod
    def forward(ctx, qkv2, mask2d, B, L, nh, D, drop_p, seed_imm):
        seed = rng.snapshot(qkv2.device) if drop_p > 0 else None
        seed_imm = rng.site(seed_imm)
        ctxv, lse = attn_fwd(qkv2, mask2d, B, L, nh, D, drop_p, seed, seed_imm)
        ctx.save_for_backward(qkv2, mask2d, ctxv, lse, seed)
        ctx.cfg = (B, L, nh, D, drop_p, seed_imm)
        return ctxv

    @staticmethod
    def backward(ctx, dctx):
        qkv2, mask2d, ctxv, lse, seed = ctx.saved_tensors
        B, L, nh, D, drop_p, seed_imm = ctx.cfg
        dqkv = attn_bwd(qkv2, mask2d, ctxv, dctx.contiguous(), lse, B, L, nh, D, drop_p, seed, seed_imm)
        return dqkv, None, None, None, None, None, None, None


class AttentionGeneralFn(torch.autograd.Function):
    """softmax(scale q k^T + key_mask + attn_mask) v with separate q / k / v [B*L, nh*D] tensors (cross-attention,
    additive attn_mask): the general form behind MultiheadAttention (model/attention.py:12-264); fp32 math; dropout on the
    probabilities with counter-based masks (the seed copy is a third, non-differentiable output for need_weights)"""

    @staticmethod
    def forward(ctx, q2, k2, v2, key_mask, attn_mask, B, Lq, Lk, nh, D, drop_p=0.0, seed=None, seed_imm=0):
        """seed: the caller's dropout seed copy (rng.snapshot; it also feeds attn_general_probs_mean), seed_imm: its site number"""
        q2, k2, v2 = q2.contiguous(), k2.contiguous(), v2.contiguous()
        H = nh * D
        out = torch.empty((B * Lq, H), dtype=q2.dtype, device=q2.device)
        lse = torch.empty((B, nh, Lq), dtype=torch.float32, device=q2.device)
        call("uc2_attn_general_fwd", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(v2), H, ptr(key_mask),
             ptr(attn_mask), 1.0 / math.sqrt(D), ptr(out), H, ptr(lse), float(drop_p), ptr(seed), seed_imm, stream())
        ctx.save_for_backward(q2, k2, v2, key_mask, attn_mask, out, lse, seed)
        ctx.cfg = (B, Lq, Lk, nh, D, float(drop_p), seed_imm)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q2, k2, v2, key_mask, attn_mask, out, lse, seed = ctx.saved_tensors
        B, Lq, Lk, nh, D, drop_p, seed_imm = ctx.cfg
        H = nh * D
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        delta = torch.empty((B, nh, Lq), dtype=torch.float32, device=q2.device)
        call("uc2_attn_general_bwd", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(v2), H, ptr(key_mask),
             ptr(attn_mask), 1.0 / math.sqrt(D), ptr(out), ptr(dout), H, ptr(lse), ptr(delta), ptr(dq), H, ptr(dk), H,
             ptr(dv), H, drop_p, ptr(seed), seed_imm, stream())
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None


def attn_general_probs_mean(q2, k2, key_mask, attn_mask, lse, B, Lq, Lk, nh, D, drop_p=0.0, seed=None, seed_imm=0):
    out = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q2.device)
    H = nh * D
    call("uc2_attn_general_probs_mean", dt(q2.dtype), B, Lq, Lk, nh, D, ptr(q2), H, ptr(k2), H, ptr(key_mask), ptr(attn_mask),
         1.0 / math.sqrt(D), ptr(lse), ptr(out), float(drop_p), ptr(seed), seed_imm, stream())
    return out


def attn_probs_mean(qkv2, mask2d, B, L, nh, D):
    out = torch.empty((B, L, L), dtype=torch.float32, device=qkv2.device)
    call("uc2_attn_probs_mean", dt(qkv2.dtype), B, L, nh, D, ptr(qkv2), ptr(mask2d), 1.0 / math.sqrt(D), ptr(out),
         stream())
    return out


import os as _os
if _os.environ.get("UC2_AUTOTUNE", "1") == "0":     # variable-shape runs that must never stall on a tuning pass
    AUTOTUNE = False
if _os.environ.get("UC2_WGRAD_SIDE"):            # experiment: "1" or "1:<spare>" (side stream for weight gradients, 8 * spare CUs left free)
    _v = _os.environ["UC2_WGRAD_SIDE"].split(":")
    WGRAD_SIDE_STREAM = _v[0] == "1"
    WGRAD_SPARE = int(_v[1]) if len(_v) > 1 else 0
if _os.environ.get("UC2_GEMM_EXTRA_FLAGS"):
    _EXTRA_FLAGS = int(_os.environ["UC2_GEMM_EXTRA_FLAGS"], 0)
if _os.environ.get("UC2_GEMM_QUEUE"):
    GEMM_QUEUE = _os.environ["UC2_GEMM_QUEUE"] == "1"
if _os.environ.get("UC2_PP_SKEW"):          # e.g. "1:2,2:2" = skew 2 for the GELU and dGELU epilogue GEMMs
    PP_SKEW = {int(k): int(v) for k, v in (kv.split(":") for kv in _os.environ["UC2_PP_SKEW"].split(","))}
if _os.environ.get("UC2_GEMM_PLANS", "1") != "0":
    load_plans(_os.environ.get("UC2_GEMM_PLANS_FILE", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                                                                    "gemm_plans.json")))
