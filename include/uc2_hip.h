/* uc2_hip.h -- C ABI of libuc2_hip.so: the MI355X (gfx950) kernels of the UC2 encoder hot path.
 *
 * The reference (zmykevin/UC2) is pure Python on PyTorch; it has no FFI layer of its own.  Its hot path
 * reaches the device through torch ops, apex.FusedLayerNorm and Horovod.  This library is what a binding
 * for that path binds instead: every entry point below names the reference code it replaces
 * (paths relative to the reference repository).  uc2_amd/_lib.py is the ctypes binding used by the
 * reference-shaped Python modules in uc2_amd/; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers, explicit sizes / leading dimensions, no torch types;
 *   - dtype: UC2_F32 (0) = fp32 parity mode, UC2_BF16 (1) = bf16 throughput mode (fp32 accumulate / statistics);
 *     parameters that are always fp32 (biases, LayerNorm gains, embedding tables, gradients of weights,
 *     optimizer state) are typed `float*`;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t) and returns; no call synchronises, allocates or
 *     frees; workspaces are caller-owned (sizes from the *_workspace functions);
 *   - return value: 0 = ok, < 0 = argument error, > 0 = hipError_t; uc2_last_error() describes the last failure
 *     of the calling thread; nothing throws across the boundary;
 *   - dropout masks are a pure function of (seed, element index): seed = *seed_ptr (device, may be NULL) + seed_imm,
 *     so the backward regenerates the forward's mask and a captured hipGraph draws fresh masks per replay;
 *   - not re-entrant per stream ordering only: one host thread per process drives the step (one process per GPU).
 */
#ifndef UC2_HIP_H
#define UC2_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { UC2_F32 = 0, UC2_BF16 = 1 };
enum { UC2_EPI_NONE = 0, UC2_EPI_GELU = 1, UC2_EPI_DGELU = 2, UC2_EPI_ADD = 3, UC2_EPI_TANH = 4 };

/* ---- library ---------------------------------------------------------------------------------------- */
int uc2_abi_version(void);
const char* uc2_last_error(void);
int uc2_device_info(int* cu_count, int* clock_khz, char* arch, int arch_len);

/* ---- dense contractions: nn.Linear forward / backward (model/layer.py:76-78,111,139,152; model/model.py:355,357;
 *      layer.py:257-265 tied decoder; model/model.py:1155 transposed regression weight) ------------------------
 *   C[M,N] (=|+=) epi( sum_k A(m,k) * B(n,k) + bias[n] )
 *   A(m,k) = A[m*lda+k] (trans_a=0) or A[k*lda+m] (trans_a=1);  B(n,k) = B[n*ldb+k] (trans_b=0) or B[k*ldb+n]
 *   forward Y = X W^T: (0,0); input gradient dX = dY W: (0,1); weight gradient dW += dY^T X: (1,1), accumulate,
 *   split_k > 1 reduces with fp32 atomics.  epilogue: GELU (erf form in fp32; the bf16 kernels evaluate Phi(x) by a sigmoid-polynomial fit,
 *   |GELU error| <= 3e-5, below bf16 resolution; optional pre-activation to aux_out),
 *   DGELU (multiply by gelu'(aux_in); optional aux_out = fp32 [N] += column sums of the result, i.e. the bias
 *   gradient of the layer whose pre-activation gradient this is), ADD (+ aux_in), TANH.  c_is_f32: fp32 output for bf16 inputs. */
int uc2_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
             void* C, int ldc, int c_is_f32, const float* bias, int epilogue, const void* aux_in, void* aux_out,
             int ldaux, int accumulate, int split_k, int variant, void* workspace, size_t workspace_bytes, int flags,
             void* stream);
/* Per-call plan -- the library keeps NO process-global kernel-selection state and reads no environment variables:
 *   variant    UC2_GEMM_AUTO = the library's default kernel for the shape; UC2_GEMM_GENERIC = the register-staged
 *              kernel (any shape/alignment); 0..12 = one specific kernel (0-2 LDS-DMA rings, 6/7 wave-specialised
 *              rings, 8 = persistent ping-pong 256x256 on v_mfma_f32_32x32x16_bf16, 9 = the same with 192-row tiles, 5 = with 128-row
 *              tiles (bf16 X W^T / dY W with no or a +residual epilogue: the N = 768 shapes at ~10 k tokens), 12 = the same
 *              schedule on v_mfma_f32_16x16x32_bf16 (less energy per flop: +4..8 % on X W^T and dY W; calls 12 does not cover run as 8);
 *              10, 11, 13 and 14 were experiment kernels -- rolling epilogue, two phases per k-tile, one wave per SIMD with and without
 *              the epilogue in the next item's MFMA gaps: experiments/csrc/, built only by `make EXPERIMENTS=1` -- and are
 *              argument errors in the shipped library, like every other number not listed).  A variant that does not
 *              support the shape falls back to the generic kernel.  uc2_amd/ops/gemm.py::gemm_plan picks it per shape.
 *   workspace  optional caller-owned device memory (>= split_k*M*N*4 bytes, 16-byte aligned) for split-K weight
 *              gradients: partial tiles are stored plainly and reduced in a second pass instead of fp32 atomics
 *              (bit-reproducible); NULL = atomics.
 *   flags      UC2_GEMM_DEFER_REDUCE: stop after the partial tiles, the caller runs uc2_gemm_splitk_reduce itself
 *              (lets a profiler time the two passes separately; the head-interleaved q|k|v weight gradient reduces with
 *              uc2_gemm_splitk_reduce_qkv).  Only the two-stage split-K path of variants 8 / 12 can honour it: a call that
 *              cannot take that path (shape, alignment, an operand of 4 GiB or more, no / short workspace) returns -1 and
 *              launches NOTHING -- it never falls back to a kernel that accumulates into C directly; UC2_GEMM_AUX_DERIV: the GELU epilogue saves
 *              gelu'(pre) instead of pre in aux_out and the DGELU epilogue multiplies by aux_in as it is (the
 *              derivative comes almost free beside the forward's Phi(x); the backward GEMM loses its transcendental
 *              epilogue); a forward/backward pair must agree on it; UC2_GEMM_SKEW(n): ping-pong start skew between phase
 *              groups, n * ~8k cycles; UC2_GEMM_DIAG(m): diagnostic launch modes (tools/bench_pp.py), 0 in production. */
enum { UC2_GEMM_AUTO = -2, UC2_GEMM_GENERIC = 99 };
enum { UC2_GEMM_DEFER_REDUCE = 1, UC2_GEMM_AUX_DERIV = 2 };
#define UC2_GEMM_SKEW(n) (((n) & 15) << 4)
#define UC2_GEMM_DIAG(m) (((m) & 0xFFFF) << 8)
#define UC2_GEMM_SPARE(n) (((n) & 7) << 28)          /* the persistent ping-pong kernels leave 8 n CUs without a workgroup: room for a
                                                       memory-bound kernel of another stream beside a weight-gradient GEMM */
#define UC2_GEMM_COLGROUP(n) (((n) & 15) << 24)     /* diagnostic: column tiles per L2 group of the ping-pong tile order (0 = default) */
/* uc2_gemm with a caller-owned item queue for the persistent ping-pong kernel: `queue` = 9 ints of device memory, zeroed once
 * (the kernel leaves them zeroed), one queue per stream that issues GEMMs concurrently.  Workgroups take their third and
 * later (tile, k-split) items from a per-XCD counter instead of a fixed stride, so a workgroup that is placed late because
 * another kernel (an overlapped all-reduce) holds its CU delays two items instead of its whole share.  Results are
 * identical to uc2_gemm; kernels other than the ping-pong variants ignore the queue; NULL = uc2_gemm. */
int uc2_gemm_queued(int dtype, int trans_a, int trans_b, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                    void* C, int ldc, int c_is_f32, const float* bias, int epilogue, const void* aux_in, void* aux_out,
                    int ldaux, int accumulate, int split_k, int variant, void* workspace, size_t workspace_bytes, int flags,
                    void* queue, void* stream);
int uc2_gemm_splitk_reduce(int M, int N, void* C, int ldc, int split_k, int accumulate, const void* workspace,
                           size_t workspace_bytes, void* stream);
/* C[M,N] (bf16) = dropout_p(A[M,K] W[N,K]^T + bias) + residual: the sum that the LayerNorm of BertSelfOutput / BertOutput normalises
 * (model/layer.py:111-115, :152-156: dense -> dropout -> LayerNorm(hidden + input)), produced in the GEMM's epilogue.  The mask is the
 * one uc2_ln_fwd / uc2_ln_bwd derive from (seed_imm + *seed_ptr, row * N + column), so the caller follows with uc2_ln_fwd(x = C,
 * residual = NULL, drop_p = 0) and, in the backward, uc2_ln_bwd*(x = C, residual = NULL, drop_p, drop_after = 2).  bf16 operands, the
 * ping-pong kernel only: returns -2 with nothing launched unless M % 256 == 0, N % 256 == 0, K % 128 == 0, every pointer 16-byte
 * aligned, every leading dimension a multiple of 8 and both operands below 4 GiB (the caller then keeps the unfused form). */
int uc2_gemm_drop_residual(int M, int N, int K, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                           const void* residual, int ldres, float p_drop, const uint64_t* seed_ptr, uint64_t seed_imm, int flags,
                           void* queue, void* stream);
/* Diagnostics: number of uc2_gemm / uc2_gemm_queued calls since load (or since the last reset != 0) that named a ping-pong kernel
 * (variant 5 / 8 / 9 / 12) and were run by another kernel because the shape, an alignment or the 32-bit staging-offset limit
 * (an operand of 4 GiB or more) did not qualify.  bench.py prints it as config.gemm_fallbacks; nothing selects a kernel from it. */
long long uc2_gemm_fallback_count(int reset);
/* Diagnostics only: e4m3 GEMM calls (uc2_gemm_fp8 / uc2_gemm_fp8_q) since the last reset that ran on the ring kernel (which = 0:
 * the shape is not made of whole 256 x 256 tiles, or UC2_GEMM_FP8_RING) or on the ping-pong kernel gemm_pp8.hip (which = 1). */
long long uc2_gemm_fp8_route_count(int which, int reset);
/* Grouped weight gradients: dW_i[n_out_i, n_in_i] (fp32) += dY_i[rows, n_out_i]^T X_i[rows, n_in_i] for up to four linear
 * layers that share the token axis -- the four dense layers of one BertLayer (model/layer.py:75-156; autograd issues their
 * weight gradients one GEMM at a time) -- as ONE launch of the persistent ping-pong kernel over all (tile, k-split) items
 * plus one reduction launch.  At the reference's 104-pair micro-batch (~10 k tokens, config/uc2_pretrain.json:17-19) a single
 * weight gradient has 9-36 output tiles and fills half of the 256 CUs for one round; four of them in one launch fill them.
 * bf16 only; n_out, n_in multiples of 256, rows a multiple of 128; workspace >= uc2_gemm_wgrad_group_workspace bytes of
 * caller-owned device memory.  Returns -2 (nothing launched) for shapes it does not take: issue uc2_gemm per item then. */
typedef struct {
  const void* dy; const void* x; void* dw;
  int lddy, ldx, lddw, n_out, n_in, split_k;
} Uc2WgradItem;
size_t uc2_gemm_wgrad_group_workspace(int n, const Uc2WgradItem* items);
int uc2_gemm_wgrad_group(int dtype, int n, const Uc2WgradItem* items, int rows, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- fp8 (OCP e4m3) forward / input-gradient GEMMs (BASELINE.json configs[4]; the reference's mixed precision is apex
 *      amp O2 fp16 -- pretrain.py:463-465 -- so this extends the bf16 mode behind the same layer interfaces) ----------
 *   per-tensor power-of-two scales, all on the device: amax (zero the 4-byte cell first; several tensors may share it)
 *   -> scale = 2^floor(log2(448/amax)) -> x8 = sat_e4m3(x * scale) [transpose != 0: out[c][r], the k-contiguous copy of
 *   W^T for dX = dY W] -> C = epi((A8 . B8^T) / (*scale_a * *scale_b) + bias) in bf16, fp32 accumulation on
 *   v_mfma_scale_f32_16x16x128_f8f6f4 (whole 256 x 256 tiles, K % 256 == 0: the persistent ping-pong schedule of variant 12 on
 *   e4m3 operands, gemm_pp8.hip) or v_mfma_scale_f32_32x32x64_f8f6f4 (any shape: the LDS-DMA ring kernel).  A8 [M,K], B8 [N,K],
 *   K % 128 == 0, lda/ldb % 16 == 0.  Weight gradients stay bf16 (uc2_gemm).  flags: UC2_GEMM_AUX_DERIV as for uc2_gemm;
 *   UC2_GEMM_FP8_RING forces the ring kernel. */
enum { UC2_GEMM_FP8_RING = 4 };
int uc2_fp8_amax(int dtype, size_t n, const void* x, void* amax_bits, void* stream);
int uc2_fp8_scale(const void* amax_bits, float* scale, void* stream);
int uc2_fp8_quant(int dtype, int rows, int cols, const void* x, int ldx, const float* scale, void* out, int ldo,
                  int transpose, void* stream);
/* the same with the scale derived from the amax cell inside the kernel and written to *scale_out (one launch less per tensor) */
int uc2_fp8_quant_amax(int dtype, int rows, int cols, const void* x, int ldx, const void* amax_bits, float* scale_out,
                       void* out, int ldo, int transpose, void* stream);
/* The e4m3 copies of up to any number of fp32 weight matrices in 3 launches per 32 of them: per item its maximum |w| (amax: one 4-byte
 * cell, cleared here), scale = the just-in-time power of two of uc2_fp8_quant_amax, out = sat_e4m3(w * scale) [rows, cols] and
 * out_t = its transpose [cols, rows] (either may be NULL).  Bit-identical to uc2_fp8_amax + uc2_fp8_quant_amax per weight.  rows and
 * cols multiples of 64, w 16-byte aligned: -2 (nothing launched) otherwise.  `items` is host memory. */
typedef struct Uc2Fp8WeightItem { const float* w; int rows, cols; void* out; void* out_t; void* amax; float* scale; } Uc2Fp8WeightItem;
int uc2_fp8_quant_weights_batch(int n, const Uc2Fp8WeightItem* items, void* stream);
/* delayed scaling, ONE pass over x: quantise with half the scale of the previous maximum (the maximum this tensor role had at its
 * previous use; twice that maximum stays representable, e4m3 saturates beyond) while accumulating max |x| for the next use, and clear
 * the cells of the use after that.  amax_prev / amax_next / amax_clear: three distinct groups of UC2_AMAX_CELLS (16) 4-byte cells the
 * caller rotates per tensor role; a maximum = the maximum over a group (producers spread their atomics over the cells; uc2_fp8_amax
 * writing cell 0 of a zeroed group is a valid group). */
enum { UC2_AMAX_CELLS = 16 };
int uc2_fp8_quant_delayed(int dtype, int rows, int cols, const void* x, int ldx, const void* amax_prev, void* amax_next,
                          void* amax_clear, float* scale_out, void* out, int ldo, void* stream);
/* uc2_gemm_fp8 whose epilogue also writes the e4m3 copy of its OUTPUT for the GEMM that consumes it (FFN1 -> FFN2: gelu(.); the input
 * gradients FFN2 -> FFN1: dY W x gelu') with delayed scaling as uc2_fp8_quant_delayed -- no quantisation pass over the [rows, 4H] tensor.
 * Ping-pong kernel only: returns -2 (nothing launched) for shapes it does not take and for epilogues other than GELU / DGELU with
 * UC2_GEMM_AUX_DERIV; the caller then runs uc2_gemm_fp8 and a quantisation pass. */
int uc2_gemm_fp8_q(int M, int N, int K, const void* A8, int lda, const void* B8, int ldb, const float* scale_a,
                   const float* scale_b, void* C, int ldc, const float* bias, int epilogue, const void* aux_in, void* aux_out,
                   int ldaux, int flags, void* q_out, int ldq, const void* amax_prev, void* amax_next, void* amax_clear,
                   float* q_scale_out, void* stream);
int uc2_gemm_fp8(int M, int N, int K, const void* A8, int lda, const void* B8, int ldb, const float* scale_a,
                 const float* scale_b, void* C, int ldc, const float* bias, int epilogue, const void* aux_in, void* aux_out,
                 int ldaux, int flags, void* stream);
/* The e4m3 form of uc2_gemm_drop_residual (model/layer.py:111-115, :152-156 in fp8 mode): C (bf16) = dropout_p(A8 W8^T / (*scale_a
 * *scale_w) + bias) + residual, the same counter-based mask (seed_imm + *seed_ptr, row, column) as uc2_ln_fwd / uc2_ln_bwd, so the caller
 * follows with uc2_ln_fwd_q(x = C, residual = NULL, drop_p = 0) and uc2_ln_bwd_partial_q(..., drop_after = 2) in the backward.
 * Ping-pong e4m3 kernel only: -2 (nothing launched) unless M % 256 == 0, N % 256 == 0, K % 256 == 0, 16-byte aligned pointers and
 * operands below 4 GiB; the caller then keeps uc2_gemm_fp8 + the LayerNorm kernel's own dropout / residual. */
int uc2_gemm_fp8_drop_residual(int M, int N, int K, const void* A8, int lda, const void* W8, int ldw, const float* scale_a,
                               const float* scale_w, void* C, int ldc, const float* bias, const void* residual, int ldres,
                               float p_drop, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream);

/* ---- One BertLayer per call (model/layer.py:159-170: BertAttention -> BertIntermediate -> BertOutput) ---------------------------
 * The launch sequence of uc2_amd/ops/layer.py BertLayerFn -- 7 launches forward, 8 backward -- enqueued by one C call per direction: the same
 * kernels with the same arguments in the same order as the per-kernel entry points above, hence the same bits.  It exists for the host:
 * at the reference's 104-pair micro-batches (config/uc2_pretrain.json:17-19) an optimizer step is ~700 launches of 10-170 us and the
 * per-launch Python/ctypes path is within 15 % of the device time.  The caller keeps every decision: it allocates all activations and
 * gradients (M = B L rows, row-major, leading dimension = width), chooses each GEMM's plan (variant / split_k / UC2_GEMM_* flags, as
 * for uc2_gemm: the library holds no kernel-selection state), runs the four weight gradients itself (uc2_gemm_wgrad_group on
 * (d_o2, u), (d_pre, a), (d_o1, ctx), (dqkv, x)) and the second stage of the two LayerNorm backwards (uc2_ln_bwd_reduce* on ws2 ->
 * dg2 / db2 / d(bf), ws1 -> dg1 / db1 / d(bo)).  Weights in the compute dtype ([out, in], q|k|v stacked), biases / LayerNorm fp32.
 * The plain route only: no fp8, no head-interleaved q|k|v, no fused dropout-residual tails (those stay in uc2_amd/ops/layer.py). */
typedef struct Uc2GemmPlan { int variant, split_k, flags; } Uc2GemmPlan;
typedef struct Uc2BertLayer {
  int dtype, B, L, H, nh, I, attn_impl;                 /* I: intermediate size; attn_impl as uc2_attn_fwd */
  float eps, p_hidden, p_attn;                          /* LayerNorm epsilon, hidden / attention-probability dropout */
  const uint64_t* seed;                                 /* device seed cell (NULL when both probabilities are 0) */
  uint64_t site_attn, site_ln1, site_ln2;               /* seed offsets of the layer's three dropout sites */
  const void *wqkv, *wo, *wi, *wf;                      /* [3H,H], [H,H], [I,H], [H,I] */
  const float *bqkv, *bo, *g1, *b1, *bi, *bf, *g2, *b2;
  const float* mask;                                    /* additive key mask [B, L] */
  const void* x;                                        /* layer input [M, H] */
  void *qkv, *ctx; float* lse;                          /* [M,3H], [M,H], [B,nh,L] */
  void* o1; float *mean1, *rstd1; void* a;              /* dense output, LayerNorm statistics, attention block output */
  void *pre, *u;                                        /* gelu'(pre-activation) and gelu(pre-activation), [M, I] */
  void* o2; float *mean2, *rstd2; void* y;              /* y: the layer output (forward only) */
  Uc2GemmPlan plan_qkv, plan_o, plan_i, plan_f;
  void* queue;                                          /* item queue of uc2_gemm_queued, or NULL */
} Uc2BertLayer;
typedef struct Uc2BertLayerGrad {
  const void* dy;                                       /* gradient of the layer output */
  void *d_o2, *dz2, *d_pre, *da, *d_o1, *dz1, *dctx, *dqkv;   /* dz1 / dz2 (residual-branch gradients) only read with p_hidden > 0 */
  void* dx;                                             /* gradient of the layer input, or NULL */
  void *ws1, *ws2;                                      /* uc2_ln_bwd_workspace(M, H) bytes each: partial sums for the caller's reduction */
  float *dbi, *dbqkv;                                   /* += gradients of the intermediate bias and of the fused q|k|v bias */
  int* attn_queue;                                      /* queue of uc2_attn_bwd_queued, or NULL */
  Uc2GemmPlan plan_df, plan_di, plan_do, plan_dqkv;     /* input-gradient GEMMs through W2, W1, Wo, Wqkv (weights read k-strided) */
} Uc2BertLayerGrad;
int uc2_bert_layer_fwd(const Uc2BertLayer* layer, void* stream);
int uc2_bert_layer_bwd(const Uc2BertLayer* layer, const Uc2BertLayerGrad* grad, void* stream);

/* ---- LayerNorm fused with dropout + residual (apex FusedLayerNorm, model/layer.py:25; the dense->dropout->
 *      LayerNorm(x + residual) tails at model/layer.py:111-115,152-156; embeddings model/model.py:331,358-362) -----
 *   drop_after == 0: y = LN(dropout(x) + residual) * gamma + beta     (encoder tails, model/layer.py:113-114,154-155)
 *   drop_after == 1: y = dropout(LN(x + residual) * gamma + beta)     (embedding tails, model/model.py:331-333,361-363)
 *   drop_after == 2 (backward only, residual == NULL): x is already dropout(dense) + residual (uc2_gemm_drop_residual): the mask
 *                    goes on dx only, dres is the unmasked gradient.
 *   mean/rstd [M] saved for the backward.
 *   backward: dx (grad of x), dres (grad of residual; may be NULL; equals dx when drop_p == 0), dgamma/dbeta
 *   accumulated (+=), and optionally dbias += column-sum(dx) = bias gradient of the dense layer producing x. */
int uc2_ln_fwd(int dtype, int M, int H, const void* x, const void* residual, const float* gamma, const float* beta,
               float eps, float drop_p, int drop_after, const uint64_t* seed_ptr, uint64_t seed_imm, void* y,
               float* mean, float* rstd, void* stream);
size_t uc2_ln_bwd_workspace(int M, int H);
int uc2_ln_bwd(int dtype, int M, int H, const void* dy, const void* x, const void* residual, const float* gamma,
               const float* mean, const float* rstd, float drop_p, int drop_after, const uint64_t* seed_ptr,
               uint64_t seed_imm, void* dx, void* dres, float* dgamma, float* dbeta, float* dbias, void* ws,
               void* stream);
/* uc2_ln_bwd in its two stages: _partial writes dx / dres and per-workgroup partial column sums into ws (want_dbias: those of dx
 * too); _reduce adds them into dgamma / dbeta / dbias (+=).  Nothing downstream in a backward pass reads the three vectors, so the
 * second stage may run on another stream (the caller orders it after the first): uc2_amd/ops/kernels.py puts it on the weight-gradient
 * side stream, off the input-gradient chain.  uc2_ln_bwd == both on one stream. */
int uc2_ln_bwd_partial(int dtype, int M, int H, const void* dy, const void* x, const void* residual, const float* gamma,
                       const float* mean, const float* rstd, float drop_p, int drop_after, const uint64_t* seed_ptr,
                       uint64_t seed_imm, void* dx, void* dres, int want_dbias, void* ws, void* stream);
int uc2_ln_bwd_reduce(int dtype, int M, int H, const void* ws, float* dgamma, float* dbeta, float* dbias, void* stream);
/* fp8 mode (BASELINE.json configs[4]): the LayerNorm kernels also write the e4m3 copy of their output that the next GEMM reads --
 * uc2_ln_fwd_q: q_out = sat_e4m3(y * scale) (the layer's LN outputs feed the QKV and FFN1 GEMMs); uc2_ln_bwd_partial_q: q_out =
 * sat_e4m3(dx * scale) (dx feeds the dense layer's input-gradient GEMM) -- with delayed scaling and the cell groups of
 * uc2_fp8_quant_delayed: no quantisation pass over these tensors.  bf16 only (forward: H % 8 == 0, H <= 1024, 16-byte aligned):
 * -2 and nothing launched otherwise. */
int uc2_ln_fwd_q(int dtype, int M, int H, const void* x, const void* residual, const float* gamma, const float* beta, float eps,
                 float drop_p, int drop_after, const uint64_t* seed_ptr, uint64_t seed_imm, void* y, float* mean, float* rstd,
                 void* q_out, const void* amax_prev, void* amax_next, void* amax_clear, float* q_scale_out, void* stream);
int uc2_ln_bwd_partial_q(int dtype, int M, int H, const void* dy, const void* x, const void* residual, const float* gamma,
                         const float* mean, const float* rstd, float drop_p, int drop_after, const uint64_t* seed_ptr,
                         uint64_t seed_imm, void* dx, void* dres, int want_dbias, void* ws, void* q_out, const void* amax_prev,
                         void* amax_next, void* amax_clear, float* q_scale_out, void* stream);
/* the second stage of up to 32 LayerNorm backwards of the same H in one launch (items: M and ws of the uc2_ln_bwd_partial call,
 * and where its sums go; any of dgamma / dbeta / dbias may be NULL): the micro-batch regime's 28 five-microsecond reductions per
 * backward pass become one kernel at the end of the pass */
typedef struct { int M; const void* ws; float* dgamma; float* dbeta; float* dbias; } Uc2LnReduceItem;
int uc2_ln_bwd_reduce_batch(int dtype, int n, const Uc2LnReduceItem* items, int H, void* stream);

/* ---- fused scaled-dot-product attention over the packed QKV projection (BertSelfAttention.forward,
 *      model/layer.py:75-101; additive key mask model/model.py:433-436) --------------------------------------------
 *   qkv [B*L, 3*nh*D] (q|k|v, head h at column h*D), mask [B, L] fp32 additive, ctx [B*L, nh*D], lse [B,nh,L].
 *   impl: 0 auto, 1 fp32-math kernels (any dtype), 2 MFMA kernels (bf16, L <= 160, D in {32,64}); OR-ed with
 *   UC2_ATTN_QKV_INTERLEAVED (MFMA kernels only): qkv and dqkv are [B*L, nh, 3, D] -- q|k|v of a head adjacent per token, what a
 *   QKV GEMM on row-interleaved weights (uc2_qkv_interleave_batch) writes: one 384-byte segment per (token, head) instead of
 *   three 128-byte ones.  ctx, dctx, lse and dbias_qkv (reference order q | k | v) are the same in both layouts. */
#define UC2_ATTN_QKV_INTERLEAVED 16
int uc2_attn_fwd(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, void* stream);
int uc2_attn_bwd(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx,
                 const float* lse, void* dqkv, float* dbias_qkv, void* stream);
/* uc2_attn_bwd with a caller-owned work queue for the MFMA kernel: `queue` = 2 ints of device memory, zeroed once (the kernel
 * leaves them zeroed), one per stream.  Workgroups take chunks of heads from an atomic counter instead of one fixed share each,
 * so a launch that shares the chip with an overlapped all-reduce does not wait a second round for the workgroups placed late.
 * Results identical to uc2_attn_bwd (dbias_qkv up to fp32 summation order); NULL = uc2_attn_bwd; the fp32-math kernels ignore it. */
int uc2_attn_bwd_queued(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                        float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx,
                        const float* lse, void* dqkv, float* dbias_qkv, int* queue, void* stream);
/* fp8 mode (BASELINE.json configs[4]): the MFMA attention kernels (plain q|k|v layout) also write the e4m3 copy of their output for the
 * GEMM that reads it -- uc2_attn_fwd_q: q_out = sat_e4m3(ctx * scale) [B L, nh D] (output projection); uc2_attn_bwd_q: q_out =
 * sat_e4m3(dqkv * scale) [B L, 3 nh D] (input gradient of the q|k|v projection) -- from the bf16-rounded values, with delayed scaling
 * and the cell groups of uc2_fp8_quant_delayed: no quantisation pass over ctx / dqkv.  bf16 only; -2 (nothing launched) when the MFMA
 * kernels do not take (L, D).  `queue` as uc2_attn_bwd_queued (may be NULL). */
int uc2_attn_fwd_q(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale, float drop_p,
                   const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, void* q_out, const void* amax_prev,
                   void* amax_next, void* amax_clear, float* q_scale_out, void* stream);
int uc2_attn_bwd_q(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale, float drop_p,
                   const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                   float* dbias_qkv, int* queue, void* q_out, const void* amax_prev, void* amax_next, void* amax_clear,
                   float* q_scale_out, void* stream);
int uc2_attn_mfma_supported(int L, int D);
/* head-averaged attention probabilities out[B, L, L] (MultiheadAttention need_weights, model/attention.py:255-260) */
int uc2_attn_probs_mean(int dtype, int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                        float* out, void* stream);

/* general form for the MultiheadAttention API (model/attention.py:12-264): separate q / k / v (token-major rows
 * b*L + i, head h at column h*D, leading dimensions ld*), Lq != Lk allowed, additive key mask [B, Lk] and additive
 * attn_mask [Lq, Lk] (either may be NULL); fp32 math; dropout on the normalised probabilities (the way the reference's only
 * user calls it, model/nlvr2.py:120-125,163-166) with the counter-based masks of uc2_attn_fwd: keep(q, k) is a function of
 * (seed, batch, head, q, k), regenerated by the backward.  delta: fp32 scratch [B, nh, Lq]. */
int uc2_attn_general_fwd(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k, int ldk,
                         const void* v, int ldv, const float* key_mask, const float* attn_mask, float scale, void* ctx,
                         int ldc, float* lse, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream);
int uc2_attn_general_bwd(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k, int ldk,
                         const void* v, int ldv, const float* key_mask, const float* attn_mask, float scale, const void* ctx,
                         const void* dctx, int ldc, const float* lse, float* delta, void* dq, int lddq, void* dk, int lddk,
                         void* dv, int lddv, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream);
int uc2_attn_general_probs_mean(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k, int ldk,
                                const float* key_mask, const float* attn_mask, float scale, const float* lse, float* out,
                                float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream);      /* (dropped weights, like the reference returns) */

/* ---- embeddings and sequence assembly (model/model.py:280-335, 352-364, 412-425) ---------------------------- */
int uc2_position_ids(int B, int T, const int64_t* ids, int64_t pad, int64_t* out, void* stream);
int uc2_embed_fwd(int dtype, int rows, int H, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                  int type_const, const float* word, const float* pos, const float* type, void* out, void* stream);
int uc2_embed_bwd(int dtype, int rows, int H, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                  const void* dpre, float* dword, float* dpos, float* dtype_tab, int64_t word_pad, int64_t pos_pad,
                  void* stream);
/* uc2_embed_bwd for ids laid out [B, T]: position / type gradient rows are summed in registers down the batch (their ids repeat from
 * sequence to sequence) and flushed with one atomic per column per id change, instead of one atomic per token and column; the word
 * rows are direct atomics as before.  Same sums up to fp32 order.  Returns -2 (nothing launched) unless H / 64 is one of 1, 2, 4, 8, 12, 16. */
int uc2_embed_bwd_seq(int dtype, int B, int T, int H, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                      const void* dpre, float* dword, float* dpos, float* dtype_tab, int64_t word_pad, int64_t pos_pad,
                      void* stream);     /* word_pad / pos_pad: nn.Embedding padding_idx rows get no gradient (-1 = none) */
int uc2_add_rowvec(int a_dtype, int dtype, int rows, int H, const void* a, const void* b, const float* vec,
                   const uint8_t* rowmask, void* out, void* stream);
int uc2_gather_rows_fwd(int dtype, int B, int S, int L, int H, const void* src, const int64_t* index, void* out,
                        void* stream);
int uc2_gather_rows_bwd(int dtype, int B, int S, int L, int H, const void* dout, const int64_t* index, void* dsrc,
                        void* stream);
/* the same over the concatenation [src1 [B,S1,H] | src2 [B,S2,H]] along dim 1 (torch.cat([txt_emb, img_emb], 1) followed by
 * the gather, model/model.py:412-425) without building it: forward reads both sources in place, backward writes dsrc1 and
 * dsrc2 as separate contiguous tensors */
int uc2_gather_rows2_fwd(int dtype, int B, int S1, int S2, int L, int H, const void* src1, const void* src2,
                         const int64_t* index, void* out, void* stream);
int uc2_gather_rows2_bwd(int dtype, int B, int S1, int S2, int L, int H, const void* dout, const int64_t* index,
                         void* dsrc1, void* dsrc2, void* stream);

/* ---- batch assembly on the device (collates data/itm.py:205-232, data/mrm.py:73-119, data/mlm.py:761-801; helpers
 *      data/data.py:360-384 pad_tensors / get_gather_index, data/mrm.py:36-39 _mask_img_feat; loader data/loader.py:85-140):
 *      flat ragged buffers (already on the device) -> the padded batch tensors.  row_off / txt_off: [B+1] prefix sums
 *      of num_bb / txt_len; mask_flat: one byte per region (masked regions are zero-filled) or NULL;
 *      outputs img_masks [B,maxR], img_mask_tgt [B,Lout], txt_labels [B,maxT] (-1 padded) are optional (NULL). */
int uc2_collate_regions(int out_dtype, int B, int maxR, int D, const float* flat, const int64_t* row_off,
                        const uint8_t* mask_flat, void* out, void* stream);
int uc2_collate_index(int B, int maxT, int maxR, int Lout, const int64_t* ids_flat, const int64_t* txt_off,
                      const int64_t* row_off, int64_t pad_id, const uint8_t* mask_flat, const int64_t* labels_flat,
                      int64_t* input_ids, int64_t* attn_masks, int64_t* gather_index, uint8_t* img_masks,
                      uint8_t* img_mask_tgt, int64_t* txt_labels, void* stream);

/* ---- heads and losses (model/model.py:583-596, 653-657, 668-688, 697-732, 738-775; model/itm.py:45-53) ----- */
/* scatter 0: dst[i] = src[rows[i]] (gather); 1: dst[rows[i]] = src[i]; 2: dst[rows[i]] += src[i] (rows unique) */
int uc2_select_rows(int dtype, int n, int H, const void* src, int ld_src, const int64_t* rows, void* dst, int ld_dst,
                    int scatter, void* stream);
/* retrieval recall without a top-k (eval/itm.py:6-53, itm.py:461-470): rank[q] = position of candidate target[q] in a
 * stable descending sort of s(q, c) = scores[off[q] + c*stride], c < nc.  dtype: UC2_F32, UC2_BF16 or 2 = fp16. */
int uc2_rank_of_target(int dtype, int nq, int nc, const void* scores, const int64_t* off, int64_t stride,
                       const int64_t* target, int32_t* rank, void* stream);
/* fp32 vector gather (mode 0: dst[i] = src[idx[i]]) / scatter-add (mode 1: dst[idx[i]] += src[i], idx unique):
 * the bias entries of the column subset kept by forward_mmxlm_soft (model/model.py:639-642) */
int uc2_gather_f32(int n, const float* src, const int64_t* idx, float* dst, int mode, void* stream);
int uc2_colsum_accum(int dtype, int M, int N, const void* X, int ldx, const uint8_t* rowmask, float* out,
                     void* stream);
int uc2_ce_fwd(int dtype, int n, int V, const void* logits, int ld, const int64_t* labels, int64_t ignore_index,
               float* loss, float* lse, int64_t* argmax, void* stream);
int uc2_ce_bwd(int dtype, int n, int V, void* logits_inout, int ld, const int64_t* labels, int64_t ignore_index,
               const float* lse, const float* gout, void* stream);
/* uc2_ce_bwd + the decoder-bias gradient in one pass: dbias[0..ncol_bias) += column sums of dlogits (column-strip tiling;
 * replaces the column-sum pass of model/layer.py:257-265's bias).  Returns -2 (nothing done) unless rows are 16-byte
 * aligned with ld % 8 == 0 (bf16) / % 4 (fp32). */
int uc2_ce_bwd_colsum(int dtype, int n, int V, void* logits_inout, int ld, const int64_t* labels, int64_t ignore_index,
                      const float* lse, const float* gout, float* dbias, int ncol_bias, void* stream);
int uc2_kl_fwd(int dtype, int n, int V, const void* pred, int ld, const float* target, const float* lse, float* loss,
               void* stream);
int uc2_kl_bwd(int dtype, int n, int V, const void* pred, int ld, const float* target, const float* lse,
               const float* gout, void* dpred, void* stream);
int uc2_mse(int dtype, size_t n, const void* pred, const float* target, const float* gout, float* loss, void* dpred,
            void* stream);
int uc2_triplet(int dtype, int n, int sample_size, float margin, const void* score, const float* gout, float* loss,
                void* dscore, void* stream);
int uc2_dtanh(int dtype, size_t n, const void* y, const void* dy, void* dx, void* stream);
int uc2_gelu(int dtype, size_t n, const void* x, void* y, void* stream);       /* model/layer.py:31-37, stand-alone */
int uc2_dgelu(int dtype, size_t n, const void* pre, const void* dy, void* dx, void* stream);
int uc2_cast(int from_dtype, int to_dtype, size_t n, const void* in, void* out, void* stream);
/* Batched bf16 transposes in one launch: for every item, dst_base[offset ..] as [cols][rows] = transpose of src_base[offset ..]
 * as [rows][cols] (offsets in elements, rows and cols multiples of 64).  Used to keep k-contiguous copies W^T of the layer
 * weights (nn.Linear.weight [out, in], model/layer.py:76-156) beside their bf16 compute copies, refreshed once per optimizer
 * step, so that the input-gradient GEMMs dX = dY W read both operands k-contiguously. */
/* head-interleaved copies of fused QKV projections (model/layer.py:76-78 query / key / value as one [3H, H] block): for every
 * item the bf16 weight rows and the fp32 bias entries go from row w nh D + h D + d (w = 0 q, 1 k, 2 v) to row h 3D + w D + d.
 * Offsets are element offsets from the four bases; b_*_base may both be NULL.  One launch for up to 64 items. */
typedef struct { size_t w_src, w_dst, b_src, b_dst; } Uc2IlvItem;
int uc2_qkv_interleave_batch(int n, const Uc2IlvItem* items, int nh, int D, int cols, const void* w_src_base, void* w_dst_base,
                             const float* b_src_base, float* b_dst_base, void* stream);
/* uc2_gemm_splitk_reduce for a weight gradient whose partial tiles have their ROWS in that interleaved order (dW = dqkv^T x with
 * interleaved dqkv): row h 3D + w D + d of the partials is added to row w M/3 + h D + d of C (the parameter arena's order) */
int uc2_gemm_splitk_reduce_qkv(int M, int N, void* C, int ldc, int split_k, int accumulate, const void* workspace,
                               size_t workspace_bytes, int qkv_head_dim, void* stream);
typedef struct { size_t offset; int rows, cols; } Uc2TransposeItem;
int uc2_transpose_batch(int n, const Uc2TransposeItem* items, const void* src_base, void* dst_base, void* stream);

/* ---- optimal-transport regulariser of the ITM head (model/ot.py:8-82; hooked at model/model.py:701-729) -----------
 *   seq [B, L, H] compact encoder output; scatter [B, L] = position of each row in the padded [txt(T) | img(R)] layout;
 *   txt_pad [B, T], img_pad [B, R] (1 = padding).  dist[b] = trace(cost Tm), cost = 1 - cosine, Tm = IPOT(beta, iters, k = 1)
 *   on the detached cost.  Tm [B, R, T] and the workspace (unit rows, norms, inverse scatter) are kept for the backward,
 *   which writes dseq rows (dseq zero-initialised by the caller).  T, R <= 128. */
size_t uc2_ot_workspace(int B, int T, int R, int H);
int uc2_ot_fwd(int dtype, int B, int L, int T, int R, int H, const void* seq, const int64_t* scatter, const uint8_t* txt_pad,
               const uint8_t* img_pad, float beta, int iters, float* dist, float* Tm, void* ws, void* stream);
int uc2_ot_bwd(int dtype, int B, int L, int T, int R, int H, const float* Tm, const void* ws, const float* gdist, void* dseq,
               void* stream);

/* ---- optimizer step and gradient clipping (optim/adamw.py:40-103; clip_grad_norm_ at pretrain.py:610) ------
 *   chunks: DEVICE array of uc2_adam_chunk records built once by the host; active / steps: DEVICE int32 per
 *   parameter (active == 0 means p.grad is None: skipped; steps = updates applied so far, bumped by the call).
 *   Per-group scalars are passed by value every step (the loop rewrites lr each step, pretrain.py:574-576). */
typedef struct uc2_adam_chunk {
  float* p; float* g; float* m; float* v; void* p_bf16;   /* p_bf16 may be NULL */
  uint32_t n; uint16_t group; uint16_t param;
} uc2_adam_chunk;
size_t uc2_adamw_chunk_bytes(void);
int uc2_adamw_step(const void* chunks, int n_chunks, int n_params, int n_groups, const float* lr, const float* beta1,
                   const float* beta2, const float* eps, const float* weight_decay, const int* correct_bias,
                   const int* active_dev, int* steps_dev, const float* grad_scale_dev, int zero_grad, void* stream);
/* deterministic (atomic-free) global norm: one slot of uc2_sumsq_blocks() partials per gradient span, then a
 * fixed-order final sum -- data-parallel replicas must compute bit-identical clip coefficients */
int uc2_sumsq_blocks(void);
int uc2_sumsq_partials(size_t n, const float* x, float* partials, void* stream);
int uc2_clip_coef(const float* partials, int count, float max_norm, float* coef, float* norm_out, void* stream);
int uc2_scale(size_t n, float* x, const float* scale_dev, float scale_imm, void* stream);

/* ---- data-parallel communication on RCCL over xGMI (utils/distributed.py:15-42 all_reduce_and_rescale_tensors,
 *      :99-147 broadcast_tensors; hvd.init / rank / size at pretrain.py:384-388) -------------------------------
 *   One communicator per process (one process per GPU).  The library owns a side HIP stream and three events and
 *   nothing else; librccl is opened lazily by uc2_comm_unique_id / uc2_comm_init.  A bucket's collective is ordered
 *   after `compute_stream` (its gradients are final), runs on the side stream while backward continues, and
 *   uc2_comm_wait(stream) orders `stream` behind everything issued so far (device-side wait; no call blocks the host
 *   except init / destroy).  dtype: UC2_F32 or UC2_BF16.  Return codes >= 10000 are ncclResult_t + 10000.
 *   Typical step (uc2_amd/utils/distributed.py): one uc2_comm_allreduce_bucket per encoder layer from that layer's
 *   backward hook, one for the embedding / head tail after backward, uc2_comm_wait, then clip + AdamW. */
int uc2_comm_unique_id_bytes(void);
int uc2_comm_unique_id(void* out, int bytes);                       /* rank 0; carry the bytes to the other ranks out of band */
int uc2_comm_init(int rank, int world, const void* unique_id, int bytes);
int uc2_comm_rank(void);
int uc2_comm_world(void);                                           /* ranks RCCL itself counted (ncclCommCount) at init; 0 = no communicator */
int uc2_comm_version(char* out, int bytes);                         /* "major.minor.patch" of the loaded librccl (hvd.init has no counterpart; for run records) */
int uc2_comm_allreduce_bucket(void* buf, size_t count, int dtype, int average, void* compute_stream);
/* the same, ordered after `compute_stream` AND `other_stream` (may be NULL): the caller's weight-gradient side stream.  A
 * layer's bucket is final when the main stream (dX chain, bias / LayerNorm gradients) and that stream (the dW GEMMs) have both
 * reached this point; the collective waits for the two events itself, the main stream is not joined behind the dW GEMMs
 * (pretrain.py:556-566 reduces after the whole backward; Horovod has no counterpart). */
int uc2_comm_allreduce_bucket_after(void* buf, size_t count, int dtype, int average, void* compute_stream, void* other_stream);
int uc2_comm_broadcast(void* buf, size_t count, int dtype, int root, void* compute_stream);
int uc2_comm_wait(void* stream);
int uc2_comm_destroy(void);
int uc2_comm_abort(void);                                           /* teardown on an error path: ncclCommAbort (does not wait for peers), then the handles */

#ifdef __cplusplus
}
#endif
#endif /* UC2_HIP_H */
