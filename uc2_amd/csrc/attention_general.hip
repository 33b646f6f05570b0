// General scaled-dot-product attention for the MultiheadAttention API surface (reference model/attention.py:12-264):
// separate query / key / value inputs (cross-attention, Lq != Lk), an additive key mask [B, Lk] (the boolean
// key_padding_mask as 0 / -1e30) and an optional additive attn_mask [Lq, Lk].  fp32 math, one thread per query row
// (forward, dQ) / per key row (dK, dV), operands read straight from global memory (these shapes are tiny: the module is
// used by the NLVR2 head only); the encoder's fused kernels live in attention.hip / attention_mfma.hip.
//   S = scale * Q K^T + key_mask[b, k] + attn_mask[q, k] ;  P = softmax_k(S) ;  ctx = P V
// Rows are token-major: row (b * L + i) of q / k / v / ctx, head h at column h * D, leading dimensions ld*.
#include "common.h"

template <typename T, int D>
__global__ __launch_bounds__(128) void attn_gen_fwd_kernel(int Lq, int Lk, int nh, const T* __restrict__ q, int ldq,
                                                           const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                           const float* __restrict__ kmask, const float* __restrict__ amask,
                                                           float scale, T* __restrict__ ctx, int ldc, float* __restrict__ lse,
                                                           uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr, uint64_t seed_imm) {
  const int bh = blockIdx.y, b = bh / nh, h = bh - b * nh;
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= Lq) return;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const uint32_t hq = thresh ? attn_line_hash(seed, bh, i, UC2_ATTN_SALT_Q) : 0u;
  const T* qr = q + ((size_t)b * Lq + i) * ldq + h * D;
  float qv[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { qv[d] = to_f<T>(qr[d]); o[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Lk; ++j) {
    const T* kr = k + ((size_t)b * Lk + j) * ldk + h * D;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s += qv[d] * to_f<T>(kr[d]);
    s = s * scale + (kmask ? kmask[(size_t)b * Lk + j] : 0.f) + (amask ? amask[(size_t)i * Lk + j] : 0.f);
    const float mn = fmaxf(m, s), al = __expf(m - mn), p = __expf(s - mn);
    l = l * al + p;
    // dropout acts on the normalised probabilities (model/attention.py: F.dropout(softmax(.))): the denominator sums the undropped p
    const float pd = thresh ? (attn_keep(hq, attn_line_hash(seed, bh, j, UC2_ATTN_SALT_K), thresh) ? p * keep_scale : 0.f) : p;
    const T* vr = v + ((size_t)b * Lk + j) * ldv + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = o[d] * al + pd * to_f<T>(vr[d]);
    m = mn;
  }
  const float inv = 1.0f / l;
  T* out = ctx + ((size_t)b * Lq + i) * ldc + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) out[d] = from_f<T>(o[d] * inv);
  if (lse) lse[(size_t)bh * Lq + i] = m + __logf(l);
}

// mode 0: thread per query -> dQ (and delta[bh, i] = dO . O); mode 1: thread per key -> dK, dV (needs delta)
template <typename T, int D, int MODE>
__global__ __launch_bounds__(128) void attn_gen_bwd_kernel(int Lq, int Lk, int nh, const T* __restrict__ q, int ldq,
                                                           const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                           const float* __restrict__ kmask, const float* __restrict__ amask,
                                                           float scale, const T* __restrict__ ctx, const T* __restrict__ dctx, int ldc,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           T* __restrict__ dq, int lddq, T* __restrict__ dk, int lddk,
                                                           T* __restrict__ dv, int lddv,
                                                           uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr, uint64_t seed_imm) {
  const int bh = blockIdx.y, b = bh / nh, h = bh - b * nh;
  const int r = blockIdx.x * 128 + threadIdx.x;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  if (MODE == 0) {
    const int i = r;
    if (i >= Lq) return;
    const T* qr = q + ((size_t)b * Lq + i) * ldq + h * D;
    const T* gr = dctx + ((size_t)b * Lq + i) * ldc + h * D;
    const T* orow = ctx + ((size_t)b * Lq + i) * ldc + h * D;
    float qv[D], gv[D], acc[D], dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = to_f<T>(qr[d]); gv[d] = to_f<T>(gr[d]); acc[d] = 0.f; dl += gv[d] * to_f<T>(orow[d]); }
    delta[(size_t)bh * Lq + i] = dl;
    const float lq = lse[(size_t)bh * Lq + i];
    const uint32_t hq = thresh ? attn_line_hash(seed, bh, i, UC2_ATTN_SALT_Q) : 0u;
    for (int j = 0; j < Lk; ++j) {
      const T* kr = k + ((size_t)b * Lk + j) * ldk + h * D;
      const T* vr = v + ((size_t)b * Lk + j) * ldv + h * D;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += qv[d] * to_f<T>(kr[d]); dp += gv[d] * to_f<T>(vr[d]); }
      s = s * scale + (kmask ? kmask[(size_t)b * Lk + j] : 0.f) + (amask ? amask[(size_t)i * Lk + j] : 0.f);
      if (thresh) dp = attn_keep(hq, attn_line_hash(seed, bh, j, UC2_ATTN_SALT_K), thresh) ? dp * keep_scale : 0.f;
      const float ds = __expf(s - lq) * (dp - dl) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] += ds * to_f<T>(kr[d]);
    }
    T* out = dq + ((size_t)b * Lq + i) * lddq + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = from_f<T>(acc[d]);
  } else {
    const int j = r;
    if (j >= Lk) return;
    const T* kr = k + ((size_t)b * Lk + j) * ldk + h * D;
    const T* vr = v + ((size_t)b * Lk + j) * ldv + h * D;
    float kv[D], vv[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { kv[d] = to_f<T>(kr[d]); vv[d] = to_f<T>(vr[d]); ak[d] = 0.f; av[d] = 0.f; }
    const float mk = kmask ? kmask[(size_t)b * Lk + j] : 0.f;
    const uint32_t hk = thresh ? attn_line_hash(seed, bh, j, UC2_ATTN_SALT_K) : 0u;
    for (int i = 0; i < Lq; ++i) {
      const T* qr = q + ((size_t)b * Lq + i) * ldq + h * D;
      const T* gr = dctx + ((size_t)b * Lq + i) * ldc + h * D;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += kv[d] * to_f<T>(qr[d]); dp += vv[d] * to_f<T>(gr[d]); }
      s = s * scale + mk + (amask ? amask[(size_t)i * Lk + j] : 0.f);
      const float p = __expf(s - lse[(size_t)bh * Lq + i]);
      float pd = p;
      if (thresh) {
        const bool keep = attn_keep(attn_line_hash(seed, bh, i, UC2_ATTN_SALT_Q), hk, thresh);
        pd = keep ? p * keep_scale : 0.f;
        dp = keep ? dp * keep_scale : 0.f;
      }
      const float ds = p * (dp - delta[(size_t)bh * Lq + i]) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) { av[d] += pd * to_f<T>(gr[d]); ak[d] += ds * to_f<T>(qr[d]); }
    }
    T* ok = dk + ((size_t)b * Lk + j) * lddk + h * D;
    T* ov = dv + ((size_t)b * Lk + j) * lddv + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { ok[d] = from_f<T>(ak[d]); ov[d] = from_f<T>(av[d]); }
  }
}

// head-averaged probabilities out[b, i, j] (need_weights, model/attention.py:255-260): thread per (query, key) pair
template <typename T>
__global__ __launch_bounds__(256) void attn_gen_probs_kernel(int Lq, int Lk, int nh, int D, const T* __restrict__ q, int ldq,
                                                             const T* __restrict__ k, int ldk, const float* __restrict__ kmask,
                                                             const float* __restrict__ amask, float scale,
                                                             const float* __restrict__ lse, float* __restrict__ out,
                                                             uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr, uint64_t seed_imm) {
  const int b = blockIdx.y;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Lq * Lk) return;
  const int i = p / Lk, j = p - i * Lk;
  float acc = 0.f;
  for (int h = 0; h < nh; ++h) {
    const T* qr = q + ((size_t)b * Lq + i) * ldq + h * D;
    const T* kr = k + ((size_t)b * Lk + j) * ldk + h * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += to_f<T>(qr[d]) * to_f<T>(kr[d]);
    s = s * scale + (kmask ? kmask[(size_t)b * Lk + j] : 0.f) + (amask ? amask[(size_t)i * Lk + j] : 0.f);
    const float pr = __expf(s - lse[((size_t)b * nh + h) * Lq + i]);
    const int bh = b * nh + h;                        // (the reference returns the dropped weights, model/attention.py:255-260)
    acc += thresh ? (attn_keep(attn_line_hash(seed, bh, i, UC2_ATTN_SALT_Q), attn_line_hash(seed, bh, j, UC2_ATTN_SALT_K), thresh) ? pr * keep_scale : 0.f) : pr;
  }
  out[((size_t)b * Lq + i) * Lk + j] = acc / (float)nh;
}

#define AG_ARGS_OK() UC2_CHECK_ARG((dtype == 0 || dtype == 1) && (D == 32 || D == 64) && Lq >= 1 && Lk >= 1 && nh >= 1 && B >= 0)

extern "C" int uc2_attn_general_fwd(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k, int ldk,
                                    const void* v, int ldv, const float* key_mask, const float* attn_mask, float scale,
                                    void* ctx, int ldc, float* lse, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream) {
  AG_ARGS_OK();
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  const uint32_t th = drop_thresh(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  if (B == 0) return 0;
  UC2_CHECK_ARG(q && k && v && ctx);
  dim3 grid((Lq + 127) / 128, B * nh);
  hipStream_t st = (hipStream_t)stream;
#define AG_FWD(TT, DD) hipLaunchKernelGGL((attn_gen_fwd_kernel<TT, DD>), grid, dim3(128), 0, st, Lq, Lk, nh, (const TT*)q, ldq, (const TT*)k, ldk, (const TT*)v, ldv, key_mask, attn_mask, scale, (TT*)ctx, ldc, lse, th, ks, seed_ptr, seed_imm)
  if (dtype == 0) { if (D == 32) AG_FWD(float, 32); else AG_FWD(float, 64); }
  else { if (D == 32) AG_FWD(bf16, 32); else AG_FWD(bf16, 64); }
#undef AG_FWD
  UC2_LAUNCH_CHECK();
  return 0;
}

// delta: caller-owned fp32 scratch [B, nh, Lq]
extern "C" int uc2_attn_general_bwd(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k, int ldk,
                                    const void* v, int ldv, const float* key_mask, const float* attn_mask, float scale,
                                    const void* ctx, const void* dctx, int ldc, const float* lse, float* delta,
                                    void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                                    float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream) {
  AG_ARGS_OK();
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  const uint32_t th = drop_thresh(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  if (B == 0) return 0;
  UC2_CHECK_ARG(q && k && v && ctx && dctx && lse && delta && dq && dk && dv);
  hipStream_t st = (hipStream_t)stream;
#define AG_BWD(TT, DD, MODE, LL) hipLaunchKernelGGL((attn_gen_bwd_kernel<TT, DD, MODE>), dim3((LL + 127) / 128, B * nh), dim3(128), 0, st, Lq, Lk, nh, (const TT*)q, ldq, (const TT*)k, ldk, (const TT*)v, ldv, key_mask, attn_mask, scale, (const TT*)ctx, (const TT*)dctx, ldc, lse, delta, (TT*)dq, lddq, (TT*)dk, lddk, (TT*)dv, lddv, th, ks, seed_ptr, seed_imm)
  if (dtype == 0) { if (D == 32) { AG_BWD(float, 32, 0, Lq); AG_BWD(float, 32, 1, Lk); } else { AG_BWD(float, 64, 0, Lq); AG_BWD(float, 64, 1, Lk); } }
  else { if (D == 32) { AG_BWD(bf16, 32, 0, Lq); AG_BWD(bf16, 32, 1, Lk); } else { AG_BWD(bf16, 64, 0, Lq); AG_BWD(bf16, 64, 1, Lk); } }
#undef AG_BWD
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_attn_general_probs_mean(int dtype, int B, int Lq, int Lk, int nh, int D, const void* q, int ldq, const void* k,
                                           int ldk, const float* key_mask, const float* attn_mask, float scale, const float* lse,
                                           float* out, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream) {
  UC2_CHECK_ARG((dtype == 0 || dtype == 1) && D >= 1 && Lq >= 1 && Lk >= 1 && nh >= 1 && B >= 0);
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  const uint32_t th = drop_thresh(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  if (B == 0) return 0;
  UC2_CHECK_ARG(q && k && lse && out);
  dim3 grid((Lq * Lk + 255) / 256, B);
  if (dtype == 0) hipLaunchKernelGGL(attn_gen_probs_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, Lq, Lk, nh, D, (const float*)q, ldq, (const float*)k, ldk, key_mask, attn_mask, scale, lse, out, th, ks, seed_ptr, seed_imm);
  else hipLaunchKernelGGL(attn_gen_probs_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, Lq, Lk, nh, D, (const bf16*)q, ldq, (const bf16*)k, ldk, key_mask, attn_mask, scale, lse, out, th, ks, seed_ptr, seed_imm);
  UC2_LAUNCH_CHECK();
  return 0;
}
