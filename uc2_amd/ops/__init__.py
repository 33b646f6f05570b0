"""uc2_amd.ops -- the Python side of the C ABI (include/uc2_hip.h): raw kernel wrappers, GEMM planning, fp8 state, streams and the
torch.autograd.Function nodes the model modules are built from.  Split by concern (round 6; one 2 500-line module before):

    base       constants of the ABI, timers, dropout seed state
    streams    weight-gradient side stream, accumulation-overlap streams, joins
    gemm       uc2_gemm wrapper, per-shape plans, tuner
    fp8        e4m3 mode: scales, delayed-scaling histories, weight copies, e4m3 GEMMs
    linear     linear forward / input gradient / weight gradient on the planned GEMMs
    kernels    LayerNorm, attention, cast wrappers
    layer      BertLayerFn (one autograd node per encoder layer)
    functions  the other autograd Functions (embeddings, heads, losses)

Knobs live in uc2_amd.config.cfg (read from UC2_* once), run-time hooks in uc2_amd.config.state; everything else is re-exported
here, so `from uc2_amd import ops; ops.gemm(...)` keeps working."""
from .. import _lib  # noqa: F401
from .._lib import call, dt, ptr, stream  # noqa: F401
from ..config import cfg, state  # noqa: F401
from ..config import cfg as knobs  # noqa: F401
from .base import (  # noqa: F401
    EPI_ADD, EPI_DGELU, EPI_GELU, EPI_NONE, EPI_TANH, GEMM_AUTO, GEMM_AUX_DERIV, GEMM_DEFER_REDUCE, GEMM_GENERIC, GemmTimer,
    HbmTimer, _FORCED, _Rng, _Timed, _require_cuda, force_variant, rng,
)
from .streams import (  # noqa: F401
    _AccumMarker, _AccumState, _accum, _accum_state, _end_of_backward_join, _join_queued, _on_side_stream,
    _queue_pass_callback, _side_dirty, _side_keep, _side_route, _side_stream, _side_streams, accum_pass, forget_accum_history, join_accum_streams,
    join_side_streams, pending_side_stream,
)
from .gemm import (  # noqa: F401
    _BORROWED, _FWD_CANDIDATES, _GEMM_QUEUES, _MAX_TUNED, _SPLITK_WS, _TUNE, _WGRAD_SPLITS, _bucket_key, _gemm_planned,
    _gemm_queue, _plan_fits, _plan_key_str, _splitk_workspace, _time_gemm, _wgrad_split, gemm, gemm_fallbacks, gemm_plan,
    load_plans, save_plans,
)
from .fp8 import (  # noqa: F401
    AMAX_CELLS, _FP8_CELLS, _FP8_HIST, _FP8_PREQ, _Fp8WeightItem, _ST_UID, _fp8_cell, _fp8_hist_for, _fp8_rotate, _fp8_weight,
    _st_uid, fp8_amax, fp8_new_forward, fp8_quantize, fp8_quantize_act, gemm_fp8, gemm_fp8_q, linear_dgrad_fp8, linear_drop_residual_fp8, linear_fwd_fp8,
)
from .linear import (  # noqa: F401
    DGRAD_ROUTES, EPI_DROPADD, _CUS, _WgradItem, _group_split, _linear_wgrad_now, _num_cus, colsum_accum, linear_dgrad,
    linear_drop_residual, linear_fwd, linear_wgrad, wgrad_group,
)
from .kernels import (  # noqa: F401
    ATTN_QKV_INTERLEAVED, _LN_BATCH_MAX, _LnReduceItem, _attn_q, _defer_ln_reduction, _flush_ln, _ln_bwd_second_stage,
    _ln_pending, _ln_pending_task, _mask2d, attn_bwd, attn_fwd, cast, flush_ln_reductions, ln_bwd, ln_fwd,
)
from .layer import (  # noqa: F401
    BertLayerFn, _BertLayerC, _BertLayerGradC, _CF, _CI, _GemmPlanC, _P_NAMES, _U64, _VP, _ilv_wgrad_plan, _native_layer_ok,
    _plan_c, layer_params,
)
from .functions import (  # noqa: F401
    AddRowFn, AttentionFn, AttentionGeneralFn, CrossEntropyFn, DecoderCEFn, EmbedTextFn, FusedQKVFn, GatherCatRowsFn,
    GatherRowsFn, GeluFn, KLDivFn, LayerNormFn, LinearFn, MSEFn, MaskEmbedFn, OTDistFn, PadRowsFn, UnpadRowsFn, padded_rows, SelectRowsFn, TiedSubsetDecoderFn,
    TripletFn, _DEC_ROWS, _dec_chunks, _dgelu, add_rowvec, attn_general_probs_mean, attn_probs_mean,
)

if knobs.gemm_plans:
    load_plans(knobs.gemm_plans_file)
