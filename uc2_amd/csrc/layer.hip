// One BertLayer per call: the launch sequences of uc2_amd/ops/layer.py BertLayerFn (model/layer.py:159-170 of the reference:
// BertAttention -> BertIntermediate -> BertOutput) enqueued by ONE C call per direction instead of ~10 / ~12 ctypes calls.
// Nothing new runs on the device: the same kernels with the same arguments in the same order, so the results are those of the
// Python route bit for bit.  What moves is the host: at the reference's 104-pair micro-batches (config/uc2_pretrain.json:17-19) a
// micro-step is ~700 launches of 10-170 us and the Python enqueue (23 ms per optimizer step) is within 15 % of the device time.
// The caller owns every buffer and every decision: activations, gradients, the GEMM plans (variant / split / flags per GEMM from
// its plan table -- the library holds no kernel-selection state), the weight gradients (uc2_gemm_wgrad_group on its side stream)
// and the second stage of the LayerNorm backwards (uc2_ln_bwd_reduce_batch at the end of the pass).
#include "../../include/uc2_hip.h"      // (before common.h: its UC2_AMAX_CELLS macro would rewrite the header's enum)
#include "common.h"
#include "gemm_common.h"

#define L_TRY(expr) do { const int rc__ = (expr); if (rc__ != 0) return rc__; } while (0)

static int layer_gemm(const Uc2BertLayer* a, int tb, int M, int N, int K, const void* A, const void* B, int ldb, void* C,
                      const float* bias, int epi, const void* aux_in, void* aux_out, int ldaux, const Uc2GemmPlan& pl, void* stream) {
  return uc2_gemm_queued(a->dtype, 0, tb, M, N, K, A, K, B, ldb, C, N, 0, bias, epi, aux_in, aux_out, ldaux, 0, pl.split_k, pl.variant,
                         nullptr, 0, pl.flags, a->queue, stream);
}

static int layer_check(const Uc2BertLayer* a) {
  UC2_CHECK_ARG(a != nullptr);
  UC2_CHECK_ARG(a->dtype == 0 || a->dtype == 1);
  UC2_CHECK_ARG(a->B > 0 && a->L > 0 && a->H > 0 && a->nh > 0 && a->I > 0 && a->H % a->nh == 0);
  UC2_CHECK_ARG(a->wqkv && a->bqkv && a->wo && a->bo && a->g1 && a->b1 && a->wi && a->bi && a->wf && a->bf && a->g2 && a->b2);
  UC2_CHECK_ARG(a->x && a->mask && a->qkv && a->ctx && a->lse && a->o1 && a->mean1 && a->rstd1 && a->a && a->pre && a->u && a->o2 &&
                a->mean2 && a->rstd2);
  UC2_CHECK_ARG(a->p_hidden >= 0.f && a->p_hidden < 1.f && a->p_attn >= 0.f && a->p_attn < 1.f);
  UC2_CHECK_ARG(!((a->p_hidden > 0.f || a->p_attn > 0.f) && a->seed == nullptr));
  return 0;
}

extern "C" int uc2_bert_layer_fwd(const Uc2BertLayer* a, void* stream) {
  L_TRY(layer_check(a));
  UC2_CHECK_ARG(a->y != nullptr);
  const int M = a->B * a->L, H = a->H, I = a->I, D = H / a->nh;
  const float scale = (float)(1.0 / sqrt((double)D));      // (the double the Python route passes, rounded once)
  // attention: fused q|k|v projection, softmax(QK^T / sqrt(D) + mask) V with dropout on the probabilities (model/layer.py:75-109)
  L_TRY(layer_gemm(a, 0, M, 3 * H, H, a->x, a->wqkv, H, a->qkv, a->bqkv, EPI_NONE, nullptr, nullptr, 0, a->plan_qkv, stream));
  L_TRY(uc2_attn_fwd(a->dtype, a->attn_impl, a->B, a->L, a->nh, D, a->qkv, a->mask, scale, a->p_attn, a->seed, a->site_attn, a->ctx,
                     a->lse, stream));
  // BertSelfOutput: dense -> dropout -> LayerNorm(. + x) (model/layer.py:111-115)
  L_TRY(layer_gemm(a, 0, M, H, H, a->ctx, a->wo, H, a->o1, a->bo, EPI_NONE, nullptr, nullptr, 0, a->plan_o, stream));
  L_TRY(uc2_ln_fwd(a->dtype, M, H, a->o1, a->x, a->g1, a->b1, a->eps, a->p_hidden, 0, a->seed, a->site_ln1, a->a, a->mean1, a->rstd1, stream));
  // BertIntermediate (GELU; `pre` receives gelu'(pre-activation), UC2_GEMM_AUX_DERIV) and BertOutput (model/layer.py:139-156)
  L_TRY(layer_gemm(a, 0, M, I, H, a->a, a->wi, H, a->u, a->bi, EPI_GELU, nullptr, a->pre, I, a->plan_i, stream));
  L_TRY(layer_gemm(a, 0, M, H, I, a->u, a->wf, I, a->o2, a->bf, EPI_NONE, nullptr, nullptr, 0, a->plan_f, stream));
  L_TRY(uc2_ln_fwd(a->dtype, M, H, a->o2, a->a, a->g2, a->b2, a->eps, a->p_hidden, 0, a->seed, a->site_ln2, a->y, a->mean2, a->rstd2, stream));
  return 0;
}

extern "C" int uc2_bert_layer_bwd(const Uc2BertLayer* a, const Uc2BertLayerGrad* g, void* stream) {
  L_TRY(layer_check(a));
  UC2_CHECK_ARG(g != nullptr);
  UC2_CHECK_ARG(g->dy && g->d_o2 && g->d_pre && g->da && g->d_o1 && g->dctx && g->dqkv && g->ws1 && g->ws2);
  UC2_CHECK_ARG(g->dbi && g->dbqkv);
  UC2_CHECK_ARG(!(a->p_hidden > 0.f && !(g->dz1 && g->dz2)));
  const int M = a->B * a->L, H = a->H, I = a->I, D = H / a->nh;
  const float scale = (float)(1.0 / sqrt((double)D));      // (the double the Python route passes, rounded once)
  // with dropout the gradient of the residual branch (dz, unmasked) differs from the gradient of the dense output (masked)
  void* dz2 = a->p_hidden > 0.f ? g->dz2 : g->d_o2;
  void* dz1 = a->p_hidden > 0.f ? g->dz1 : g->d_o1;
  // LayerNorm 2 (first stage: dx, dres, partial column sums in ws2; the caller reduces them into dg2 / db2 / dbf later)
  L_TRY(uc2_ln_bwd_partial(a->dtype, M, H, g->dy, a->o2, a->a, a->g2, a->mean2, a->rstd2, a->p_hidden, 0, a->seed, a->site_ln2, g->d_o2,
                           a->p_hidden > 0.f ? g->dz2 : nullptr, 1, g->ws2, stream));
  // d_pre = (d_o2 W2) * gelu'(pre), dbi += column sums; da = d_pre W1 + dz2
  L_TRY(layer_gemm(a, 1, M, I, H, g->d_o2, a->wf, I, g->d_pre, nullptr, EPI_DGELU, a->pre, g->dbi, I, g->plan_df, stream));
  L_TRY(layer_gemm(a, 1, M, H, I, g->d_pre, a->wi, H, g->da, nullptr, EPI_ADD, dz2, nullptr, H, g->plan_di, stream));
  // LayerNorm 1, output projection, attention (dbqkv += column sums of dqkv), fused q|k|v projection
  L_TRY(uc2_ln_bwd_partial(a->dtype, M, H, g->da, a->o1, a->x, a->g1, a->mean1, a->rstd1, a->p_hidden, 0, a->seed, a->site_ln1, g->d_o1,
                           a->p_hidden > 0.f ? g->dz1 : nullptr, 1, g->ws1, stream));
  L_TRY(layer_gemm(a, 1, M, H, H, g->d_o1, a->wo, H, g->dctx, nullptr, EPI_NONE, nullptr, nullptr, 0, g->plan_do, stream));
  if (g->attn_queue)
    L_TRY(uc2_attn_bwd_queued(a->dtype, a->attn_impl, a->B, a->L, a->nh, D, a->qkv, a->mask, scale, a->p_attn, a->seed, a->site_attn,
                              a->ctx, g->dctx, a->lse, g->dqkv, g->dbqkv, g->attn_queue, stream));
  else
    L_TRY(uc2_attn_bwd(a->dtype, a->attn_impl, a->B, a->L, a->nh, D, a->qkv, a->mask, scale, a->p_attn, a->seed, a->site_attn, a->ctx,
                       g->dctx, a->lse, g->dqkv, g->dbqkv, stream));
  if (g->dx)
    L_TRY(layer_gemm(a, 1, M, H, 3 * H, g->dqkv, a->wqkv, H, g->dx, nullptr, EPI_ADD, dz1, nullptr, H, g->plan_dqkv, stream));
  return 0;
}
