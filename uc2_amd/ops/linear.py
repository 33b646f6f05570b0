"""Linear-layer building blocks on the planned GEMMs: forward, fused dropout + residual tail, input gradients (W or W^T), weight
gradients (side stream / grouped launch).  Part of uc2_amd.ops."""
import ctypes
import functools

import torch

from .. import _lib
from .._lib import call, dt, ptr, stream
from ..config import cfg as knobs, state
from .base import EPI_ADD, EPI_DGELU, EPI_NONE, GEMM_GENERIC
from .streams import _on_side_stream, _side_route
from .gemm import _BORROWED, _TUNE, _bucket_key, _gemm_planned, _gemm_queue, _plan_fits, _splitk_workspace, gemm


def linear_fwd(x2, w, bias, epi=EPI_NONE, aux_out=None, flags=0):
    M, K = x2.shape
    N = w.shape[0]
    return _gemm_planned(x2, w, M, N, K, False, False, bias=bias, epi=epi, aux_out=aux_out, flags=flags)


EPI_DROPADD = 10                 # internal to the ping-pong kernel (uc2_gemm_drop_residual)


# knobs.ln_fuse: dropout + residual of the dense -> dropout -> LayerNorm tails in the GEMM epilogue (bit 0 = the attention-output
# tail, bit 1 = the FFN tail), from knobs.ln_fuse_min_rows tokens
def linear_drop_residual(x2, w, bias, res2, drop_p, seed, seed_imm):
    """s = dropout(x2 w^T + bias) + res2 (model/layer.py:111-115, :152-156 up to the LayerNorm) from one GEMM launch, with the mask
    ln_fwd / ln_bwd derive from (seed, seed_imm); None (nothing launched) when the ping-pong kernel does not take the shape"""
    M, K = x2.shape
    N = w.shape[0]
    if x2.dtype != torch.bfloat16 or M % 256 or N % 256 or K % 128:
        return None
    out = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    timer = state.gemm_timer
    e0 = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    flags = knobs.gemm_extra_flags
    if knobs.pp_skew:
        flags |= (knobs.pp_skew.get(EPI_ADD, 0) & 15) << 4
    rc = _lib.load().uc2_gemm_drop_residual(M, N, K, ptr(x2), x2.stride(0), ptr(w), w.stride(0), ptr(out), N, ptr(bias), ptr(res2),
                                            res2.stride(0), float(drop_p), ptr(seed), seed_imm, flags,
                                            ptr(_gemm_queue(x2.device)) if knobs.gemm_queue else None, stream())
    if rc == -2:
        return None
    _lib.check(rc)
    if e0 is not None:
        e1.record()
        timer.add((False, False, 12, False, EPI_DROPADD), 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + 2 * M * N))
    return out
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
        if hit is None and M >= knobs.dgrad_wt_min_rows:
            # no plan for this token count: take the plan of the nearest tuned token count of the same (N, K) -- above the
            # threshold the choice between the kernels does not depend on M any more (every committed plan there is variant 12)
            near = [(abs(k[2] - M), v) for k, v in _TUNE.items() if not k[0] and not k[1] and not k[5] and k[3] == K and k[4] == N
                    and k[2] >= knobs.dgrad_wt_min_rows]
            if near:
                hit = min(near, key=lambda t: t[0])[1]
                _BORROWED[key] = hit                    # not a measured plan: kept out of _TUNE (save_plans, bench.py's gemm_plans)
        if hit is not None and hit[0] == 12 and _plan_fits(hit, key):
            route = "W^T"
    if wt is not None:
        DGRAD_ROUTES[(M, K, N, int(epi))] = route
    if route == "W^T":
        if knobs.pp_skew:
            flags |= (knobs.pp_skew.get(epi, 0) & 15) << 4
        return gemm(dy2, wt, M, K, N, split_k=1, variant=hit[0], epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)
    return _gemm_planned(dy2, w, M, K, N, False, True, epi=epi, aux_in=aux_in, aux_out=colsum_out, flags=flags)


def _linear_wgrad_now(dy2, x2, dw, db):
    M, N = dy2.shape
    K = x2.shape[1]
    M0 = M // 256 * 256
    if dy2.dtype == torch.bfloat16 and M0 >= 1024 and M0 != M and dy2.is_contiguous() and x2.is_contiguous():
        # a ragged token count (the image projection of a variable-length batch: B x max(num_bb) rows; the encoder layers pad
        # theirs, ops.padded_rows): the contraction runs over the tokens, so a count that is not a multiple of 64 sends the WHOLE
        # weight gradient to the generic kernel -- and every new count is a new shape for the plan table.  Whole 256-row tiles on
        # the planned kernel, the < 256 remaining rows on the generic one, both accumulating into dw.
        _gemm_planned(dy2[:M0], x2[:M0], N, K, M0, True, True, wgrad=True, out=dw, accumulate=True, flags=(knobs.wgrad_spare & 7) << 28)
        gemm(dy2[M0:], x2[M0:], N, K, M - M0, ta=True, tb=True, out=dw, accumulate=True, split_k=1, variant=GEMM_GENERIC)
    else:
        _gemm_planned(dy2, x2, N, K, M, True, True, wgrad=True, out=dw, accumulate=True, flags=(knobs.wgrad_spare & 7) << 28)
    if db is not None:
        colsum_accum(dy2, db)


def linear_wgrad(dy2, x2, dw, db):
    """dW[N,K] += dY^T X ; db[N] += colsum(dY)  (fp32 accumulation buffers)"""
    if not _side_route(dy2.shape[0]):
        return _linear_wgrad_now(dy2, x2, dw, db)
    _on_side_stream(dy2.device, lambda: _linear_wgrad_now(dy2, x2, dw, db), (dy2, x2))


# Below WGRAD_SIDE_MIN_ROWS tokens a layer's four weight gradients are too small to fill the chip one by one (9-36 output tiles
# each at ~10 k tokens = a single round on half of the 256 CUs): BertLayerFn.backward hands them to ONE launch of the persistent
# ping-pong kernel (uc2_gemm_wgrad_group) at the end of the layer's backward.  104-pair micro-batch: 215 -> 136 us per layer.
# (knobs.wgrad_group; knobs.wgrad_group_side puts that launch on the side stream)
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


@functools.lru_cache(maxsize=256)
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
    ok = (knobs.wgrad_group and dy0.dtype == torch.bfloat16 and 1 <= len(triples) <= 4 and rows % 128 == 0
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
