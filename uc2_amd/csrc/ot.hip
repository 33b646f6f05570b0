// Optimal-transport regulariser of the ITM head (reference model/ot.py:8-82, hooked at model/model.py:701-729).
//
//   ctx[b, scatter[b,l], :] = seq[b, l, :]          -> txt = ctx[:, :T], img = ctx[:, T:T+R]   (zero rows where nothing lands)
//   cost[b,i,j] = 1 - <txt_i, img_j> / (max(|txt_i|, eps) max(|img_j|, eps)),  0 where either side is padding
//   Tm = IPOT(cost.detach(), beta = 0.5, 50 iterations, k = 1)                                  [B, R, T]
//   dist[b] = trace(cost Tm) = sum_ij cost[b,i,j] Tm[b,j,i]
// backward (Tm is a constant): d cost[b,i,j] = g[b] Tm[b,j,i]  ->  d x^_i = - sum_j d cost_ij y^_j  ->  through the
// row normalisation  ->  scattered back to d seq[b, l, :].
//
// Three small kernels; per batch element the whole problem (T <= 128 text rows, R <= 128 regions) lives in LDS.  All
// arithmetic is fp32 whatever the activation dtype.  Tiny next to the encoder (3.4 GFLOP per 1024 pairs): written for
// correctness and zero host round trips, not tuned.
#include "common.h"

#define OT_MAXN 128

// x^ rows (fp32) and their norms in the padded [txt | img] layout, from the compact sequence through the inverse of
// the scatter index: row p of batch b comes from position l with scatter[b,l] == p (none: zero row, model/model.py:712-716)
template <typename T>
__global__ __launch_bounds__(256) void ot_gather_norm_kernel(int L, int P, int H, const T* __restrict__ seq,
                                                             const int64_t* __restrict__ scatter, float eps,
                                                             float* __restrict__ xn, float* __restrict__ nrm,
                                                             int* __restrict__ inv) {
  __shared__ int src[2 * OT_MAXN];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int p = t; p < P; p += 256) src[p] = -1;
  __syncthreads();
  for (int l = t; l < L; l += 256) {
    const int64_t p = scatter[(size_t)b * L + l];
    if (p >= 0 && p < P) src[p] = l;
  }
  __syncthreads();
  for (int p = wv; p < P; p += 4) {
    const int l = src[p];
    if (lane == 0) inv[(size_t)b * P + p] = l;
    float ss = 0.f;
    if (l >= 0)
      for (int h = lane; h < H; h += 64) { const float v = to_f<T>(seq[((size_t)b * L + l) * H + h]); ss += v * v; }
    ss = wave_sum(ss);
    const float n = sqrtf(ss), sc = 1.0f / fmaxf(n, eps);
    if (lane == 0) nrm[(size_t)b * P + p] = n;
    for (int h = lane; h < H; h += 64)
      xn[((size_t)b * P + p) * H + h] = (l >= 0) ? to_f<T>(seq[((size_t)b * L + l) * H + h]) * sc : 0.f;
  }
}

__global__ __launch_bounds__(256) void ot_ipot_kernel(int Tn, int Rn, int H, const float* __restrict__ xn,
                                                      const uint8_t* __restrict__ txt_pad, const uint8_t* __restrict__ img_pad,
                                                      float beta, int iters, float* __restrict__ dist, float* __restrict__ Tm_out) {
  extern __shared__ float sm[];
  float* C = sm;                       // [Tn][Rn]
  float* A = C + Tn * Rn;              // [Rn][Tn]  exp(-C^T / beta), 0 at padding
  float* Tm = A + Tn * Rn;             // [Rn][Tn]
  float* sig = Tm + Tn * Rn;           // [Tn]
  float* del = sig + Tn;               // [Rn]
  float* red = del + Rn;               // [4]
  __shared__ uint8_t xp[OT_MAXN], yp[OT_MAXN];
  const int b = blockIdx.x, t = threadIdx.x, P = Tn + Rn;
  const float* X = xn + (size_t)b * P * H;
  const float* Y = X + (size_t)Tn * H;
  for (int i = t; i < Tn; i += 256) xp[i] = txt_pad[(size_t)b * Tn + i];
  for (int j = t; j < Rn; j += 256) yp[j] = img_pad[(size_t)b * Rn + j];
  __syncthreads();
  float xl = 0.f, yl = 0.f;            // lengths = number of non-padded rows (model/ot.py:73-76)
  for (int i = 0; i < Tn; ++i) xl += xp[i] ? 0.f : 1.f;
  for (int j = 0; j < Rn; ++j) yl += yp[j] ? 0.f : 1.f;
  for (int p = t; p < Tn * Rn; p += 256) {
    const int i = p / Rn, j = p - i * Rn;
    float d = 0.f;
    const float4* xi = reinterpret_cast<const float4*>(X + (size_t)i * H);
    const float4* yj = reinterpret_cast<const float4*>(Y + (size_t)j * H);
    for (int h = 0; h < H / 4; ++h) { const float4 a = xi[h], c = yj[h]; d += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w; }
    const bool pad = xp[i] || yp[j];
    const float c = pad ? 0.f : 1.0f - d;
    C[p] = c;
    A[j * Tn + i] = pad ? 0.f : __expf(-c / beta);
    Tm[j * Tn + i] = pad ? 0.f : 1.0f;
  }
  for (int i = t; i < Tn; i += 256) sig[i] = xp[i] ? 0.f : 1.0f / xl;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int j = t; j < Rn; j += 256) {                 // delta = 1 / (y_len * Q sigma + y_mask)
      float s = 0.f;
      for (int i = 0; i < Tn; ++i) s += A[j * Tn + i] * Tm[j * Tn + i] * sig[i];
      del[j] = 1.0f / (yl * s + (yp[j] ? 1e4f : 0.f));
    }
    __syncthreads();
    for (int i = t; i < Tn; i += 256) {                 // sigma = 1 / (x_len * delta Q + x_mask)
      float s = 0.f;
      for (int j = 0; j < Rn; ++j) s += del[j] * A[j * Tn + i] * Tm[j * Tn + i];
      sig[i] = 1.0f / (xl * s + (xp[i] ? 1e4f : 0.f));
    }
    __syncthreads();
    for (int p = t; p < Tn * Rn; p += 256) {            // T = delta * Q * sigma
      const int j = p / Tn, i = p - j * Tn;
      Tm[p] = del[j] * (A[p] * Tm[p]) * sig[i];
    }
    __syncthreads();
  }
  float acc = 0.f;
  for (int p = t; p < Tn * Rn; p += 256) {
    const int j = p / Tn, i = p - j * Tn;
    const float tv = (xp[i] || yp[j]) ? 0.f : Tm[p];
    Tm_out[(size_t)b * Tn * Rn + p] = tv;
    acc += C[i * Rn + j] * tv;
  }
  acc = wave_sum(acc);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t == 0) dist[b] = red[0] + red[1] + red[2] + red[3];
}

// d seq from d dist: one wave per row of the padded layout
template <typename T>
__global__ __launch_bounds__(256) void ot_bwd_kernel(int L, int Tn, int Rn, int H, const float* __restrict__ xn,
                                                     const float* __restrict__ nrm, const int* __restrict__ inv,
                                                     const float* __restrict__ Tm, const float* __restrict__ gdist,
                                                     float eps, T* __restrict__ dseq) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, P = Tn + Rn;
  const float g = gdist[b];
  const float* X = xn + (size_t)b * P * H;
  const float* TM = Tm + (size_t)b * Tn * Rn;
  for (int p = wv; p < P; p += 4) {
    const int l = inv[(size_t)b * P + p];
    if (l < 0) continue;                                  // (wave-uniform)
    const bool is_txt = p < Tn;
    const int idx = is_txt ? p : p - Tn;
    const int nother = is_txt ? Rn : Tn;
    const float* O = is_txt ? X + (size_t)Tn * H : X;     // the other side's unit rows
    float dot = 0.f;
    for (int h0 = lane * 4; h0 < H; h0 += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int o = 0; o < nother; ++o) {
        const float w = is_txt ? TM[o * Tn + idx] : TM[idx * Tn + o];     // Tm[j][i]
        const float4 v = *reinterpret_cast<const float4*>(O + (size_t)o * H + h0);
        a0 += w * v.x; a1 += w * v.y; a2 += w * v.z; a3 += w * v.w;
      }
      // d x^ = - g * sum_o Tm * y^_o ; keep it in the output buffer's place holder registers via recompute below
      const float4 me = *reinterpret_cast<const float4*>(X + (size_t)p * H + h0);
      dot += -g * (a0 * me.x + a1 * me.y + a2 * me.z + a3 * me.w);
    }
    dot = wave_sum(dot);
    const float n = nrm[(size_t)b * P + p];
    const bool big = n > eps;
    const float inv_n = 1.0f / fmaxf(n, eps);
    for (int h0 = lane * 4; h0 < H; h0 += 256) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      for (int o = 0; o < nother; ++o) {
        const float w = is_txt ? TM[o * Tn + idx] : TM[idx * Tn + o];
        const float4 v = *reinterpret_cast<const float4*>(O + (size_t)o * H + h0);
        a[0] += w * v.x; a[1] += w * v.y; a[2] += w * v.z; a[3] += w * v.w;
      }
      const float4 me = *reinterpret_cast<const float4*>(X + (size_t)p * H + h0);
      const float mev[4] = {me.x, me.y, me.z, me.w};
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dxh = -g * a[e];
        o4[e] = big ? (dxh - mev[e] * dot) * inv_n : dxh * inv_n;        // d (x / max(|x|, eps))
      }
      Vec4<T>::store(dseq + ((size_t)b * L + l) * H + h0, o4);
    }
  }
}

extern "C" size_t uc2_ot_workspace(int B, int T, int R, int H) {
  const size_t P = (size_t)T + R;
  return ((size_t)B * P * H + (size_t)B * P) * sizeof(float) + (size_t)B * P * sizeof(int) + 256;
}

// forward: dist[B] (fp32), Tm[B, R, T] (fp32, kept for the backward), ws = uc2_ot_workspace bytes (kept too)
extern "C" int uc2_ot_fwd(int dtype, int B, int L, int T, int R, int H, const void* seq, const int64_t* scatter,
                          const uint8_t* txt_pad, const uint8_t* img_pad, float beta, int iters, float* dist, float* Tm,
                          void* ws, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(T >= 1 && R >= 1 && T <= OT_MAXN && R <= OT_MAXN && (H % 4) == 0 && L >= 1 && iters >= 0 && beta > 0.f);
  if (B <= 0) return 0;
  UC2_CHECK_ARG(seq && scatter && txt_pad && img_pad && dist && Tm && ws);
  const size_t smem = ((size_t)3 * T * R + T + R + 8) * sizeof(float);
  {   // cost, transport plan and kernel matrices live in LDS for the 50 iterations: 3 T R floats must fit one workgroup's share
    int dev = 0, lim = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) lim = 64 * 1024;
    if (smem > (size_t)lim) {
      uc2_set_error(__FILE__, __LINE__, "uc2_ot_fwd: T * R too large for the LDS-resident IPOT kernel (3 * T * R * 4 bytes must fit one workgroup's LDS)");
      return -1;
    }
  }
  const int P = T + R;
  float* xn = reinterpret_cast<float*>(ws);
  float* nrm = xn + (size_t)B * P * H;
  int* inv = reinterpret_cast<int*>(nrm + (size_t)B * P);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(ot_gather_norm_kernel<float>, dim3(B), dim3(256), 0, st, L, P, H, (const float*)seq, scatter, 1e-5f, xn, nrm, inv);
  else hipLaunchKernelGGL(ot_gather_norm_kernel<bf16>, dim3(B), dim3(256), 0, st, L, P, H, (const bf16*)seq, scatter, 1e-5f, xn, nrm, inv);
  UC2_LAUNCH_CHECK();
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)ot_ipot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(ot_ipot_kernel, dim3(B), dim3(256), smem, st, T, R, H, xn, txt_pad, img_pad, beta, iters, dist, Tm);
  UC2_LAUNCH_CHECK();
  return 0;
}

// backward: dseq[B, L, H] must be zero-initialised by the caller (rows nothing scatters from get no gradient)
extern "C" int uc2_ot_bwd(int dtype, int B, int L, int T, int R, int H, const float* Tm, const void* ws, const float* gdist,
                          void* dseq, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(T >= 1 && R >= 1 && T <= OT_MAXN && R <= OT_MAXN && (H % 4) == 0);
  if (B <= 0) return 0;
  UC2_CHECK_ARG(Tm && ws && gdist && dseq);
  const int P = T + R;
  const float* xn = reinterpret_cast<const float*>(ws);
  const float* nrm = xn + (size_t)B * P * H;
  const int* inv = reinterpret_cast<const int*>(nrm + (size_t)B * P);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(ot_bwd_kernel<float>, dim3(B), dim3(256), 0, st, L, T, R, H, xn, nrm, inv, Tm, gdist, 1e-5f, (float*)dseq);
  else hipLaunchKernelGGL(ot_bwd_kernel<bf16>, dim3(B), dim3(256), 0, st, L, T, R, H, xn, nrm, inv, Tm, gdist, 1e-5f, (bf16*)dseq);
  UC2_LAUNCH_CHECK();
  return 0;
}
