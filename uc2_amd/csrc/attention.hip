// Fused scaled-dot-product attention over the packed QKV projection (reference
// BertSelfAttention.forward, model/layer.py:75-101): per (batch, head)
//     S = Q K^T / sqrt(d) + mask[b, key]   (additive key mask, model/model.py:433-436)
//     P = softmax_keys(S) ; P <- dropout(P) ; ctx = P V ; heads merged in place.
// Q, K, V are read straight out of the fused-QKV GEMM output [B*L, 3*H] (row = token,
// columns q|k|v, head h at h*D) and ctx is written head-merged into [B*L, H]: no
// transpose_for_scores / permute copies, and the [B,h,L,L] score tensor never exists in HBM.
// Only the per-row log-sum-exp is kept for the backward, which recomputes P.
//
// This file holds the reference-exact fp32-math kernels ("simple": one thread per query /
// per key, K,V or Q,dO staged in LDS as fp32).  They are the parity path for both dtypes and
// the on-device checker for the MFMA kernels in attention_mfma.hip.
#include "common.h"
#define UC2_ATTN_QKV_INTERLEAVED 16        /* include/uc2_hip.h: OR-ed into `impl` */

#define AT_PAD 4

template <typename T, int D>
__device__ __forceinline__ void stage_rows(float* dst, const T* __restrict__ src, int L, int ld, int tid, int nthr) {
  // dst[l][D+AT_PAD] <- src[l*ld + 0..D)
  const int per_row = D / 4;
  for (int c = tid; c < L * per_row; c += nthr) {
    const int l = c / per_row, d4 = (c - l * per_row) * 4;
    float v[4];
    Vec4<T>::load(src + (size_t)l * ld + d4, v);
    *reinterpret_cast<float4*>(dst + l * (D + AT_PAD) + d4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_simple(int B, int L, int nh, const T* __restrict__ qkv,
                                                       const float* __restrict__ mask, float scale,
                                                       uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr, uint64_t seed_imm,
                                                       T* __restrict__ ctx, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + L * (D + AT_PAD);
  float* Ms = Vs + L * (D + AT_PAD);
  const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const int H = nh * D, ld = 3 * H;
  const T* base = qkv + (size_t)b * L * ld + h * D;
  stage_rows<T, D>(Ks, base + H, L, ld, threadIdx.x, 256);
  stage_rows<T, D>(Vs, base + 2 * H, L, ld, threadIdx.x, 256);
  for (int k = threadIdx.x; k < L; k += 256) Ms[k] = mask ? mask[(size_t)b * L + k] : 0.f;
  __syncthreads();
  for (int q = threadIdx.x; q < L; q += 256) {
    float qr[D], o[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4];
      Vec4<T>::load(base + (size_t)q * ld + d, v);
      qr[d] = v[0]; qr[d + 1] = v[1]; qr[d + 2] = v[2]; qr[d + 3] = v[3];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int k = 0; k < L; ++k) {
      const float* kr = Ks + k * (D + AT_PAD);
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(kr + d);
        s += qr[d] * kv.x + qr[d + 1] * kv.y + qr[d + 2] * kv.z + qr[d + 3] * kv.w;
      }
      s = s * scale + Ms[k];
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn);         // exp(-inf) = 0 on the first key
      const float p = __expf(s - mn);
      l = l * alpha + p;
      float pd = p;
      if (thresh) pd = attn_keep(attn_line_hash(seed, bh, q, UC2_ATTN_SALT_Q), attn_line_hash(seed, bh, k, UC2_ATTN_SALT_K), thresh) ? p * keep_scale : 0.f;
      const float* vr = Vs + k * (D + AT_PAD);
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(vr + d);
        o[d] = o[d] * alpha + pd * vv.x;
        o[d + 1] = o[d + 1] * alpha + pd * vv.y;
        o[d + 2] = o[d + 2] * alpha + pd * vv.z;
        o[d + 3] = o[d + 3] * alpha + pd * vv.w;
      }
      m = mn;
    }
    const float inv = 1.0f / l;
    T* out = ctx + ((size_t)b * L + q) * H + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4] = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
      Vec4<T>::store(out + d, v);
    }
    if (lse) lse[(size_t)bh * L + q] = m + __logf(l);
  }
}

// backward: phase A (thread per query) -> dQ ; phase B1/B2 (thread per key) -> dV, dK
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_simple(int B, int L, int nh, const T* __restrict__ qkv,
                                                       const float* __restrict__ mask, float scale,
                                                       uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr, uint64_t seed_imm,
                                                       const T* __restrict__ ctx, const T* __restrict__ dctx,
                                                       const float* __restrict__ lse, T* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* X0 = sm;                              // K  then Q
  float* X1 = sm + L * (D + AT_PAD);           // V  then dO
  float* Ms = X1 + L * (D + AT_PAD);           // mask[L]
  float* Ls = Ms + L;                          // lse[L]
  float* Ds = Ls + L;                          // delta[L]
  const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const int H = nh * D, ld = 3 * H;
  const T* base = qkv + (size_t)b * L * ld + h * D;
  const T* dob = dctx + (size_t)b * L * H + h * D;
  const T* ob = ctx + (size_t)b * L * H + h * D;
  T* dbase = dqkv + (size_t)b * L * ld + h * D;

  stage_rows<T, D>(X0, base + H, L, ld, threadIdx.x, 256);       // K
  stage_rows<T, D>(X1, base + 2 * H, L, ld, threadIdx.x, 256);   // V
  for (int k = threadIdx.x; k < L; k += 256) {
    Ms[k] = mask ? mask[(size_t)b * L + k] : 0.f;
    Ls[k] = lse[(size_t)bh * L + k];
  }
  __syncthreads();
  // ---- phase A: dQ (and delta) ----
  for (int q = threadIdx.x; q < L; q += 256) {
    float qr[D], dor[D], dq[D];
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4], g[4], o[4];
      Vec4<T>::load(base + (size_t)q * ld + d, v);
      Vec4<T>::load(dob + (size_t)q * H + d, g);
      Vec4<T>::load(ob + (size_t)q * H + d, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qr[d + e] = v[e]; dor[d + e] = g[e]; dq[d + e] = 0.f; delta += g[e] * o[e]; }
    }
    Ds[q] = delta;
    const float lq = Ls[q];
    for (int k = 0; k < L; ++k) {
      const float* kr = X0 + k * (D + AT_PAD);
      const float* vr = X1 + k * (D + AT_PAD);
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(kr + d);
        const float4 vv = *reinterpret_cast<const float4*>(vr + d);
        s += qr[d] * kv.x + qr[d + 1] * kv.y + qr[d + 2] * kv.z + qr[d + 3] * kv.w;
        dpd += dor[d] * vv.x + dor[d + 1] * vv.y + dor[d + 2] * vv.z + dor[d + 3] * vv.w;
      }
      const float p = __expf(s * scale + Ms[k] - lq);
      float dp = dpd;
      if (thresh) dp = attn_keep(attn_line_hash(seed, bh, q, UC2_ATTN_SALT_Q), attn_line_hash(seed, bh, k, UC2_ATTN_SALT_K), thresh) ? dpd * keep_scale : 0.f;
      const float ds = p * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(kr + d);
        dq[d] += ds * kv.x; dq[d + 1] += ds * kv.y; dq[d + 2] += ds * kv.z; dq[d + 3] += ds * kv.w;
      }
    }
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4] = {dq[d], dq[d + 1], dq[d + 2], dq[d + 3]};
      Vec4<T>::store(dbase + (size_t)q * ld + d, v);
    }
  }
  __syncthreads();
  stage_rows<T, D>(X0, base, L, ld, threadIdx.x, 256);            // Q
  stage_rows<T, D>(X1, dob, L, H, threadIdx.x, 256);              // dO
  __syncthreads();
  // ---- phase B1: dV[k] = sum_q Pdrop[q,k] dO[q] ----
  for (int k = threadIdx.x; k < L; k += 256) {
    float kr[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4];
      Vec4<T>::load(base + H + (size_t)k * ld + d, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kr[d + e] = v[e]; acc[d + e] = 0.f; }
    }
    const float mk = Ms[k];
    for (int q = 0; q < L; ++q) {
      const float* qr = X0 + q * (D + AT_PAD);
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qr + d);
        s += kr[d] * qv.x + kr[d + 1] * qv.y + kr[d + 2] * qv.z + kr[d + 3] * qv.w;
      }
      float p = __expf(s * scale + mk - Ls[q]);
      if (thresh) p = attn_keep(attn_line_hash(seed, bh, q, UC2_ATTN_SALT_Q), attn_line_hash(seed, bh, k, UC2_ATTN_SALT_K), thresh) ? p * keep_scale : 0.f;
      const float* gr = X1 + q * (D + AT_PAD);
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(gr + d);
        acc[d] += p * gv.x; acc[d + 1] += p * gv.y; acc[d + 2] += p * gv.z; acc[d + 3] += p * gv.w;
      }
    }
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4] = {acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
      Vec4<T>::store(dbase + 2 * H + (size_t)k * ld + d, v);
    }
  }
  // ---- phase B2: dK[k] = sum_q dS[q,k] * scale * Q[q] ----
  for (int k = threadIdx.x; k < L; k += 256) {
    float kr[D], vr[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4], u[4];
      Vec4<T>::load(base + H + (size_t)k * ld + d, v);
      Vec4<T>::load(base + 2 * H + (size_t)k * ld + d, u);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kr[d + e] = v[e]; vr[d + e] = u[e]; acc[d + e] = 0.f; }
    }
    const float mk = Ms[k];
    for (int q = 0; q < L; ++q) {
      const float* qr = X0 + q * (D + AT_PAD);
      const float* gr = X1 + q * (D + AT_PAD);
      float s = 0.f, dpd = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qr + d);
        const float4 gv = *reinterpret_cast<const float4*>(gr + d);
        s += kr[d] * qv.x + kr[d + 1] * qv.y + kr[d + 2] * qv.z + kr[d + 3] * qv.w;
        dpd += vr[d] * gv.x + vr[d + 1] * gv.y + vr[d + 2] * gv.z + vr[d + 3] * gv.w;
      }
      const float p = __expf(s * scale + mk - Ls[q]);
      float dp = dpd;
      if (thresh) dp = attn_keep(attn_line_hash(seed, bh, q, UC2_ATTN_SALT_Q), attn_line_hash(seed, bh, k, UC2_ATTN_SALT_K), thresh) ? dpd * keep_scale : 0.f;
      const float ds = p * (dp - Ds[q]) * scale;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qr + d);
        acc[d] += ds * qv.x; acc[d + 1] += ds * qv.y; acc[d + 2] += ds * qv.z; acc[d + 3] += ds * qv.w;
      }
    }
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float v[4] = {acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
      Vec4<T>::store(dbase + H + (size_t)k * ld + d, v);
    }
  }
}

template <typename T, int D>
static int launch_fwd(int B, int L, int nh, const void* qkv, const float* mask, float scale, float drop_p,
                      const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, hipStream_t st) {
  const size_t smem = (size_t)(2 * L * (D + AT_PAD) + L) * sizeof(float);
  auto kern = attn_fwd_simple<T, D>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3(B * nh), dim3(256), smem, st, B, L, nh, (const T*)qkv, mask, scale,
                     drop_thresh(drop_p), 1.0f / (1.0f - drop_p), seed_ptr, seed_imm, (T*)ctx, lse);
  UC2_LAUNCH_CHECK();
  return 0;
}

template <typename T, int D>
static int launch_bwd(int B, int L, int nh, const void* qkv, const float* mask, float scale, float drop_p,
                      const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                      hipStream_t st) {
  const size_t smem = (size_t)(2 * L * (D + AT_PAD) + 3 * L) * sizeof(float);
  auto kern = attn_bwd_simple<T, D>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3(B * nh), dim3(256), smem, st, B, L, nh, (const T*)qkv, mask, scale,
                     drop_thresh(drop_p), 1.0f / (1.0f - drop_p), seed_ptr, seed_imm, (const T*)ctx, (const T*)dctx, lse,
                     (T*)dqkv);
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_attn_fwd_simple(int dtype, int B, int L, int nh, int D, const void* qkv, const float* mask,
                                   float scale, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(D == 32 || D == 64);
  UC2_CHECK_ARG(L >= 1 && L <= 512 && B >= 0 && nh >= 1);
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG((size_t)(2 * L * (D + AT_PAD) + 3 * L) * sizeof(float) <= 160 * 1024);
  if (B == 0) return 0;
  UC2_CHECK_ARG(qkv && ctx);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    if (D == 32) return launch_fwd<float, 32>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, st);
    return launch_fwd<float, 64>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, st);
  }
  if (D == 32) return launch_fwd<bf16, 32>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, st);
  return launch_fwd<bf16, 64>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, st);
}

extern "C" int uc2_attn_bwd_simple(int dtype, int B, int L, int nh, int D, const void* qkv, const float* mask,
                                   float scale, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx,
                                   const float* lse, void* dqkv, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(D == 32 || D == 64);
  UC2_CHECK_ARG(L >= 1 && L <= 512 && B >= 0 && nh >= 1);
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG((size_t)(2 * L * (D + AT_PAD) + 3 * L) * sizeof(float) <= 160 * 1024);
  if (B == 0) return 0;
  UC2_CHECK_ARG(qkv && ctx && dctx && lse && dqkv);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    if (D == 32) return launch_bwd<float, 32>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, st);
    return launch_bwd<float, 64>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, st);
  }
  if (D == 32) return launch_bwd<bf16, 32>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, st);
  return launch_bwd<bf16, 64>(B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, st);
}

// ---- public dispatch: impl 0 = auto, 1 = simple (fp32 math), 2 = MFMA (bf16 only) ----
extern "C" int uc2_attn_fwd_mfma(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse,
                                 int ilv, void* stream);
extern "C" int uc2_attn_bwd_mfma(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx,
                                 const void* dctx, const float* lse, void* dqkv, float* dbias, int* queue, int ilv, void* stream);
extern "C" int uc2_colsum_accum(int dtype, int M, int N, const void* X, int ldx, const uint8_t* rowmask, float* out, void* stream);
extern "C" int uc2_attn_mfma_supported(int L, int D);

extern "C" int uc2_attn_fwd(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask,
                            float scale, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx,
                            float* lse, void* stream) {
  const int ilv = (impl & UC2_ATTN_QKV_INTERLEAVED) ? 1 : 0;      // q|k|v of a head interleaved per token (MFMA kernels only)
  impl &= ~UC2_ATTN_QKV_INTERLEAVED;
  UC2_CHECK_ARG(impl >= 0 && impl <= 2);
  const bool mfma = (impl == 2) || (impl == 0 && dtype == 1 && uc2_attn_mfma_supported(L, D));
  UC2_CHECK_ARG(mfma || !ilv);
  if (mfma) {
    UC2_CHECK_ARG(dtype == 1 && uc2_attn_mfma_supported(L, D));
    return uc2_attn_fwd_mfma(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, ilv, stream);
  }
  return uc2_attn_fwd_simple(dtype, B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, stream);
}
extern "C" int uc2_attn_bwd_queued(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask,
                                   float scale, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx,
                                   const void* dctx, const float* lse, void* dqkv, float* dbias_qkv, int* queue, void* stream) {
  const int ilv = (impl & UC2_ATTN_QKV_INTERLEAVED) ? 1 : 0;
  impl &= ~UC2_ATTN_QKV_INTERLEAVED;
  UC2_CHECK_ARG(impl >= 0 && impl <= 2);
  const bool mfma = (impl == 2) || (impl == 0 && dtype == 1 && uc2_attn_mfma_supported(L, D));
  UC2_CHECK_ARG(mfma || !ilv);
  if (mfma) {
    UC2_CHECK_ARG(dtype == 1 && uc2_attn_mfma_supported(L, D));
    return uc2_attn_bwd_mfma(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, dbias_qkv, queue, ilv, stream);
  }
  const int rc = uc2_attn_bwd_simple(dtype, B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv,
                                     stream);
  if (rc != 0 || !dbias_qkv || B == 0) return rc;          // fp32-math kernels: the column sums are one more pass
  return uc2_colsum_accum(dtype, B * L, 3 * nh * D, dqkv, 3 * nh * D, nullptr, dbias_qkv, stream);
}
extern "C" int uc2_attn_bwd(int dtype, int impl, int B, int L, int nh, int D, const void* qkv, const float* mask,
                            float scale, float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx,
                            const void* dctx, const float* lse, void* dqkv, float* dbias_qkv, void* stream) {
  return uc2_attn_bwd_queued(dtype, impl, B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv,
                             dbias_qkv, nullptr, stream);
}

// ---- head-averaged attention probabilities (MultiheadAttention need_weights=True, reference
//      model/attention.py:255-260): out[b, q, k] = mean_h softmax_k(scale * Q_h K_h^T + mask)[q, k].
//      fp32 math, forward only (the reference's callers discard the gradient of the weights). -------------
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_mean_kernel(int L, int nh, int D, const T* __restrict__ qkv,
                                                              const float* __restrict__ mask, float scale,
                                                              float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ float qs[256];
  const int bq = blockIdx.x, b = bq / L, q = bq - b * L;
  const int H = nh * D, ld = 3 * H, t = threadIdx.x;
  const T* base = qkv + (size_t)b * L * ld;
  for (int k0 = 0; k0 < L; k0 += 256) {
    const int k = k0 + t;
    if (k < L) out[(size_t)bq * L + k] = 0.f;
  }
  for (int h = 0; h < nh; ++h) {
    __syncthreads();
    if (t < D) qs[t] = to_f<T>(base[(size_t)q * ld + h * D + t]);
    __syncthreads();
    // pass 1: max, pass 2: sum, pass 3: write (L <= 512 -> at most 2 keys per thread, recomputed)
    float sc[2], m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = t + 256 * r;
      sc[r] = -INFINITY;
      if (k < L) {
        float s = 0.f;
        const T* kr = base + (size_t)k * ld + H + h * D;
        for (int d = 0; d < D; ++d) s += qs[d] * to_f<T>(kr[d]);
        sc[r] = s * scale + (mask ? mask[(size_t)b * L + k] : 0.f);
        m = fmaxf(m, sc[r]);
      }
    }
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float e[2], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) { e[r] = (t + 256 * r < L) ? __expf(sc[r] - m) : 0.f; sum += e[r]; }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / (sum * (float)nh);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = t + 256 * r;
      if (k < L) out[(size_t)bq * L + k] += e[r] * inv;
    }
  }
}
extern "C" int uc2_attn_probs_mean(int dtype, int B, int L, int nh, int D, const void* qkv, const float* mask,
                                   float scale, float* out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(L >= 1 && L <= 512 && D >= 1 && D <= 256 && nh >= 1);
  if (B <= 0) return 0;
  UC2_CHECK_ARG(qkv && out);
  if (dtype == 0) hipLaunchKernelGGL(attn_probs_mean_kernel<float>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, L, nh, D, (const float*)qkv, mask, scale, out);
  else hipLaunchKernelGGL(attn_probs_mean_kernel<bf16>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, L, nh, D, (const bf16*)qkv, mask, scale, out);
  UC2_LAUNCH_CHECK();
  return 0;
}
