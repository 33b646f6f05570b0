"""ctypes binding of libuc2_hip.so (the C-ABI declared in include/uc2_hip.h).

The library is loaded on first use.  If it is missing or a symbol is absent this
raises: there is NO CPU fallback anywhere in uc2_amd (the oracle under oracle/ is
test infrastructure and is never imported from here).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

from .config import cfg

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = cfg.lib_path      # (UC2_LIB_PATH: A/B of two builds on one box)
_lib = None
ABI_VERSION = 13         # include/uc2_hip.h; bumped whenever a signature changes

P, I, F, U64, I64, SZ = c_void_p, c_int, c_float, c_uint64, c_int64, c_size_t

# name -> (restype, argtypes); must list every symbol of include/uc2_hip.h
SIGNATURES = {
    "uc2_abi_version": (I, []),
    "uc2_last_error": (c_char_p, []),
    "uc2_device_info": (I, [P, P, P, I]),
    "uc2_gemm": (I, [I, I, I, I, I, I, P, I, P, I, P, I, I, P, I, P, P, I, I, I, I, P, SZ, I, P]),
    "uc2_gemm_queued": (I, [I, I, I, I, I, I, P, I, P, I, P, I, I, P, I, P, P, I, I, I, I, P, SZ, I, P, P]),
    "uc2_gemm_splitk_reduce": (I, [I, I, P, I, I, I, P, SZ, P]),
    "uc2_bert_layer_fwd": (I, [P, P]),
    "uc2_bert_layer_bwd": (I, [P, P, P]),
    "uc2_gemm_drop_residual": (I, [I, I, I, P, I, P, I, P, I, P, P, I, F, P, U64, I, P, P]),
    "uc2_gemm_fallback_count": (ctypes.c_longlong, [I]),
    "uc2_gemm_fp8_route_count": (ctypes.c_longlong, [I, I]),
    "uc2_gemm_wgrad_group_workspace": (SZ, [I, P]),
    "uc2_gemm_wgrad_group": (I, [I, I, P, I, P, SZ, P]),
    "uc2_fp8_quant_weights_batch": (I, [I, P, P]),
    "uc2_attn_fwd_q": (I, [I, I, I, I, P, P, F, F, P, U64, P, P, P, P, P, P, P, P]),
    "uc2_attn_bwd_q": (I, [I, I, I, I, P, P, F, F, P, U64, P, P, P, P, P, P, P, P, P, P, P, P]),
    "uc2_fp8_amax": (I, [I, SZ, P, P, P]),
    "uc2_fp8_scale": (I, [P, P, P]),
    "uc2_fp8_quant": (I, [I, I, I, P, I, P, P, I, I, P]),
    "uc2_fp8_quant_amax": (I, [I, I, I, P, I, P, P, P, I, I, P]),
    "uc2_fp8_quant_delayed": (I, [I, I, I, P, I, P, P, P, P, P, I, P]),
    "uc2_gemm_fp8_q": (I, [I, I, I, P, I, P, I, P, P, P, I, P, I, P, P, I, I, P, I, P, P, P, P, P]),
    "uc2_gemm_fp8_drop_residual": (I, [I, I, I, P, I, P, I, P, P, P, I, P, P, I, F, P, U64, P]),
    "uc2_gemm_fp8": (I, [I, I, I, P, I, P, I, P, P, P, I, P, I, P, P, I, I, P]),
    "uc2_ln_fwd": (I, [I, I, I, P, P, P, P, F, F, I, P, U64, P, P, P, P]),
    "uc2_ln_bwd_workspace": (SZ, [I, I]),
    "uc2_ln_bwd": (I, [I, I, I, P, P, P, P, P, P, F, I, P, U64, P, P, P, P, P, P, P]),
    "uc2_ln_bwd_partial": (I, [I, I, I, P, P, P, P, P, P, F, I, P, U64, P, P, I, P, P]),
    "uc2_ln_bwd_reduce": (I, [I, I, I, P, P, P, P, P]),
    "uc2_ln_fwd_q": (I, [I, I, I, P, P, P, P, F, F, I, P, U64, P, P, P, P, P, P, P, P, P]),
    "uc2_ln_bwd_partial_q": (I, [I, I, I, P, P, P, P, P, P, F, I, P, U64, P, P, I, P, P, P, P, P, P, P]),
    "uc2_ln_bwd_reduce_batch": (I, [I, I, P, I, P]),
    "uc2_attn_fwd": (I, [I, I, I, I, I, I, P, P, F, F, P, U64, P, P, P]),
    "uc2_attn_bwd": (I, [I, I, I, I, I, I, P, P, F, F, P, U64, P, P, P, P, P, P]),
    "uc2_attn_bwd_queued": (I, [I, I, I, I, I, I, P, P, F, F, P, U64, P, P, P, P, P, P, P]),
    "uc2_attn_mfma_supported": (I, [I, I]),
    "uc2_attn_probs_mean": (I, [I, I, I, I, I, P, P, F, P, P]),
    "uc2_attn_general_fwd": (I, [I, I, I, I, I, I, P, I, P, I, P, I, P, P, F, P, I, P, F, P, U64, P]),
    "uc2_attn_general_bwd": (I, [I, I, I, I, I, I, P, I, P, I, P, I, P, P, F, P, P, I, P, P, P, I, P, I, P, I, F, P, U64, P]),
    "uc2_attn_general_probs_mean": (I, [I, I, I, I, I, I, P, I, P, I, P, P, F, P, P, F, P, U64, P]),
    "uc2_position_ids": (I, [I, I, P, I64, P, P]),
    "uc2_embed_fwd": (I, [I, I, I, P, P, P, I, P, P, P, P, P]),
    "uc2_embed_bwd": (I, [I, I, I, P, P, P, P, P, P, P, I64, I64, P]),
    "uc2_embed_bwd_seq": (I, [I, I, I, I, P, P, P, P, P, P, P, I64, I64, P]),
    "uc2_gather_rows_fwd": (I, [I, I, I, I, I, P, P, P, P]),
    "uc2_gather_rows_bwd": (I, [I, I, I, I, I, P, P, P, P]),
    "uc2_transpose_batch": (I, [I, P, P, P, P]),
    "uc2_qkv_interleave_batch": (I, [I, P, I, I, I, P, P, P, P, P]),
    "uc2_gemm_splitk_reduce_qkv": (I, [I, I, P, I, I, I, P, SZ, I, P]),
    "uc2_gather_rows2_fwd": (I, [I, I, I, I, I, I, P, P, P, P, P]),
    "uc2_gather_rows2_bwd": (I, [I, I, I, I, I, I, P, P, P, P, P]),
    "uc2_collate_regions": (I, [I, I, I, I, P, P, P, P, P]),
    "uc2_collate_index": (I, [I, I, I, I, P, P, P, I64, P, P, P, P, P, P, P, P, P]),
    "uc2_select_rows": (I, [I, I, I, P, I, P, P, I, I, P]),
    "uc2_rank_of_target": (I, [I, I, I, P, P, I64, P, P, P]),
    "uc2_gather_f32": (I, [I, P, P, P, I, P]),
    "uc2_colsum_accum": (I, [I, I, I, P, I, P, P, P]),
    "uc2_add_rowvec": (I, [I, I, I, I, P, P, P, P, P, P]),
    "uc2_ce_fwd": (I, [I, I, I, P, I, P, I64, P, P, P, P]),
    "uc2_ce_bwd": (I, [I, I, I, P, I, P, I64, P, P, P]),
    "uc2_ce_bwd_colsum": (I, [I, I, I, P, I, P, I64, P, P, P, I, P]),
    "uc2_kl_fwd": (I, [I, I, I, P, I, P, P, P, P]),
    "uc2_kl_bwd": (I, [I, I, I, P, I, P, P, P, P, P]),
    "uc2_mse": (I, [I, SZ, P, P, P, P, P, P]),
    "uc2_triplet": (I, [I, I, I, F, P, P, P, P, P]),
    "uc2_dtanh": (I, [I, SZ, P, P, P, P]),
    "uc2_gelu": (I, [I, SZ, P, P, P]),
    "uc2_dgelu": (I, [I, SZ, P, P, P, P]),
    "uc2_cast": (I, [I, I, SZ, P, P, P]),
    "uc2_ot_workspace": (SZ, [I, I, I, I]),
    "uc2_ot_fwd": (I, [I, I, I, I, I, I, P, P, P, P, F, I, P, P, P, P]),
    "uc2_ot_bwd": (I, [I, I, I, I, I, I, P, P, P, P, P]),
    "uc2_adamw_chunk_bytes": (SZ, []),
    "uc2_adamw_step": (I, [P, I, I, I, P, P, P, P, P, P, P, P, P, I, P]),
    "uc2_sumsq_partials": (I, [SZ, P, P, P]),
    "uc2_sumsq_blocks": (I, []),
    "uc2_clip_coef": (I, [P, I, F, P, P, P]),
    "uc2_scale": (I, [SZ, P, P, F, P]),
    "uc2_comm_unique_id_bytes": (I, []),
    "uc2_comm_unique_id": (I, [P, I]),
    "uc2_comm_init": (I, [I, I, P, I]),
    "uc2_comm_rank": (I, []),
    "uc2_comm_world": (I, []),
    "uc2_comm_version": (I, [P, I]),
    "uc2_comm_allreduce_bucket": (I, [P, SZ, I, I, P]),
    "uc2_comm_allreduce_bucket_after": (I, [P, SZ, I, I, P, P]),
    "uc2_comm_broadcast": (I, [P, SZ, I, I, P]),
    "uc2_comm_wait": (I, [P]),
    "uc2_comm_destroy": (I, []),
    "uc2_comm_abort": (I, []),
}


class Uc2Error(RuntimeError):
    pass


def load():
    """dlopen the library and declare every signature; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Uc2Error(
            "uc2_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C uc2_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Uc2Error("uc2_amd: symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.uc2_abi_version() != ABI_VERSION:
        raise Uc2Error("uc2_amd: %s has ABI version %d, this package needs %d: rebuild it (make -C uc2_amd/csrc)"
                       % (LIB_PATH, lib.uc2_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().uc2_last_error()
        raise Uc2Error("uc2 kernel call failed (rc=%d): %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """raw device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """raw hipStream_t of torch's current stream on the current device (every kernel launch asks: the two C accessors cost
    ~0.5 us, torch.cuda.current_stream().cuda_stream ~10 us of Python)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def dt(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise Uc2Error("uc2_amd supports float32 and bfloat16 compute, got %s" % dtype)


def call(name, *args):
    check(getattr(load(), name)(*args))
