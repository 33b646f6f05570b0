"""Every knob of the Python layer in ONE object, read from the environment ONCE (at import).

`from uc2_amd.config import cfg` -- the ops modules read `cfg.<knob>` at call time, tests and tools assign `cfg.<knob> = value`
(there is no other process-global kernel-selection state in Python; the C library keeps none at all and reads no environment
variable).  INTEGRATION.md section 4 is the user-facing table of the UC2_* variables.  `state` holds what is not configuration
but is shared between the ops modules at run time (instrumentation hooks, the fp8 task tag).
"""
import os


def _flag(env, name, default):
    """UC2_X unset -> default; "0" -> False; anything else -> True"""
    v = env.get(name)
    return default if v is None or v == "" else v != "0"


class Config:
    def __init__(self, env=None):
        env = os.environ if env is None else env
        here = os.path.dirname(os.path.abspath(__file__))
        # ---------------------------------------------------------------- library
        self.lib_path = env.get("UC2_LIB_PATH") or os.path.join(here, "libuc2_hip.so")   # another build (A/B of two builds on one box)
        # ---------------------------------------------------------------- GEMM planning
        self.autotune = _flag(env, "UC2_AUTOTUNE", True)        # time candidate kernels once per (layout, shape); 0: variable-shape runs that must never stall
        self.gemm_plans = _flag(env, "UC2_GEMM_PLANS", True)     # preload the committed (variant, split-K) table
        self.gemm_plans_file = env.get("UC2_GEMM_PLANS_FILE", os.path.join(here, "gemm_plans.json"))
        self.gemm_extra_flags = int(env.get("UC2_GEMM_EXTRA_FLAGS", "0"), 0)    # diagnostics: OR-ed into the flags of every uc2_gemm call
        self.pp_skew = {}                                        # epilogue kind -> ping-pong start skew (experiment knob, empty = off)
        if env.get("UC2_PP_SKEW"):                               # e.g. "1:2,2:2" = skew 2 for the GELU and dGELU epilogue GEMMs
            self.pp_skew = {int(k): int(v) for k, v in (kv.split(":") for kv in env["UC2_PP_SKEW"].split(","))}
        # persistent GEMMs take work items from a per-XCD queue (uc2_gemm_queued): on when N > 1 (GradSync / bench.py), where a
        # communication kernel holds CUs; at N = 1 it changed nothing (61.6-61.7 ms without, 61.8-61.9 with).  UC2_GEMM_QUEUE=0 keeps
        # it off even then, =1 forces it on
        self.gemm_queue = env.get("UC2_GEMM_QUEUE") == "1"
        self.gemm_queue_allowed = env.get("UC2_GEMM_QUEUE", "1") != "0"
        # ---------------------------------------------------------------- routes of the layer (thresholds in tokens = rows)
        self.attn_impl = 0                                       # 0 auto, 1 fp32-math kernels, 2 MFMA kernels (tests flip this)
        self.ln_fuse = int(env.get("UC2_LN_FUSE", "3"))          # dropout + residual of the Wo (bit 0) / FFN2 (bit 1) tail in the GEMM epilogue
        self.ln_fuse_min_rows = int(env.get("UC2_LN_FUSE_MIN_ROWS", "16384"))
        self.dgrad_transposed_w = _flag(env, "UC2_DGRAD_WT", True)       # bf16: dX = dY W reads a k-contiguous copy W^T (store.compute_t)
        self.dgrad_wt_min_rows = 16384                           # ... from this many tokens (at 9 984 rows the ring kernels on W win)
        self.qkv_interleaved = _flag(env, "UC2_QKV_ILV", True)   # head-interleaved q|k|v activations
        self.qkv_ilv_min_rows = 16384
        self.native_layer = _flag(env, "UC2_NATIVE_LAYER", True)         # below 16 384 tokens: one C call per layer and direction
        self.embed_bwd_seq = _flag(env, "UC2_EMBED_BWD_SEQ", True)       # per-position embedding backward (uc2_embed_bwd_seq)
        # bf16 encoder passes over a ragged token count B L run on B L rounded up to whole 256-row GEMM tiles (zero rows appended
        # once in VLXLMREncoder.forward: ops.PadRowsFn), so that every GEMM stays on its planned ping-pong kernel
        self.pad_rows = _flag(env, "UC2_PAD_ROWS", True)
        # MLM head: dz = dlogits E reads a k-contiguous copy E^T of the tied decoder table (store.table_t, 384 MB) instead of E
        # through the transposing LDS read
        self.decoder_wt = _flag(env, "UC2_DECODER_WT", True)
        # ---------------------------------------------------------------- streams
        # weight-gradient GEMMs on a second HIP stream (round 3, 1024-pair step on one box: 63.1-63.3 ms against 63.8-64.4);
        # UC2_WGRAD_SIDE = "0" | "1" | "1:<n>" (n: they leave 8 n CUs free -- measured slower, 65.5 ms at 16-24 CUs)
        self.wgrad_side_stream, self.wgrad_spare = True, 0
        if env.get("UC2_WGRAD_SIDE"):
            v = env["UC2_WGRAD_SIDE"].split(":")
            self.wgrad_side_stream = v[0] == "1"
            self.wgrad_spare = int(v[1]) if len(v) > 1 else 0
        self.wgrad_side_min_rows = 16384                         # below: close to host-bound, the side stream made the regime erratic (27.6-45.5 ms)
        self.wgrad_group = _flag(env, "UC2_WGRAD_GROUP", True)   # below that: a layer's four weight gradients as ONE grouped launch
        self.wgrad_group_side = env.get("UC2_GROUP_SIDE", "1") == "1"    # ... on the side stream (27.70-27.72 -> 27.11-27.27 ms per optimizer step)
        self.ln_reduce_side = _flag(env, "UC2_LN_REDUCE_SIDE", True)     # LayerNorm parameter-gradient reduction on the side stream
        self.ln_reduce_batch = _flag(env, "UC2_LN_REDUCE_BATCH", True)   # ... or batched into one launch per backward pass (small token counts)
        # gradient accumulation overlapped inside the top-level models (ops/streams.py accum_pass): forward i+1 beside backward i
        self.accum_overlap = _flag(env, "UC2_ACCUM_OVERLAP", True)
        # up to this many tokens per micro-batch (r06_experiments.md section 1: +3.9 % at 9 984 tokens, +7 % on the ragged 18 k-token
        # windows of the retrieval finetune, +1.7 % at 20 k, +0.8 % at 40 k, -0.6 % at 80 k, -1.8 % at 160 k)
        self.accum_overlap_max_rows = int(env.get("UC2_ACCUM_OVERLAP_MAX_ROWS", "49152"))
        # ---------------------------------------------------------------- fp8 mode
        self.fp8_delayed = _flag(env, "UC2_FP8_DELAYED", True)           # delayed (one-pass, producer-fused) activation scaling
        self.fp8_weight_batch = _flag(env, "UC2_FP8_WEIGHT_BATCH", True)  # all e4m3 weight copies of a store from one call per optimizer step
        # attention kernels write the e4m3 copies of ctx / dqkv themselves: OFF -- measured break-even on uc2-large (63.5-63.8 ms per
        # step without, 63.9-64.4 with; profiles/r05_experiments.md section 2)
        self.fp8_attn_fused = _flag(env, "UC2_FP8_ATTN_Q", False)
        # ---------------------------------------------------------------- data parallelism
        # dtype of the exposed embedding / head tail of the gradient all-reduce: fp32 (default: the library never rounds gradients
        # on its own), bf16 / auto = bf16 for stores that compute in bf16 (utils/distributed.py)
        self.allreduce_tail = env.get("UC2_ALLREDUCE_TAIL", "fp32")
        self.gloo_direct = env.get("UC2_GLOO_DIRECT", "0") == "1"        # gloo test route: let gloo handle device tensors itself
        # ---------------------------------------------------------------- checks
        self.check_hints = bool(env.get("UC2_CHECK_HINTS"))      # verify n_txt_labels / n_img_mask_tgt against the masks (host sync)


class State:
    """run-time state shared by the ops modules (not configuration)"""

    def __init__(self):
        self.gemm_timer = None     # bench.py installs an ops.GemmTimer: HIP-event timing per GEMM kernel instantiation
        self.hbm_timer = None      # ... an ops.HbmTimer: HIP-event timing of the HBM-bound kernels with their algorithmic bytes
        self.comm_timer = None     # ... a list that receives (start, end) HIP events around the EXPOSED part of the gradient all-reduce
        self.fp8_tag = None        # (task name, loss or scores) of the running model forward: part of every fp8 tensor-role key


cfg = Config()
state = State()
