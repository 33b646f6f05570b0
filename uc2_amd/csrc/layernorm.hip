// Row LayerNorm fused with dropout + residual add (replaces apex FusedLayerNorm,
// reference model/layer.py:25, and the dense->dropout->LN(x+res) tails at
// model/layer.py:111-115,152-156; embeddings LN at model/model.py:331,358-362).
//
//   z = dropout(x) + residual ;  y = (z - mean) * rstd * gamma + beta      (biased variance)
//
// One 64-lane wave per row, 4 rows per 256-thread workgroup; statistics in fp32; the row lives
// in registers (H <= 1024, H % 4 == 0), vector loads of 4 elements per lane.  HBM-bound:
// algorithmic bytes per row = (2 or 3) * H * sizeof(T) forward.
#include "common.h"

#define LN_MAXC 4     // chunks of 4 elements per lane: H <= 64*4*4 = 1024

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int M, int H, const T* __restrict__ x, const T* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr,
                                                     uint64_t seed_imm, T* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int drop_after) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = H >> 2;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const uint32_t hrow = drop_row_hash(seed, (uint32_t)row);
  // every load of the row is issued before the first use (chunks past the row end are clamped, not branched around:
  // a branch per chunk makes hipcc wait for each load before issuing the next)
  typedef typename Raw4<T>::type raw_t;
  raw_t rx[LN_MAXC], rr[LN_MAXC];
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) rx[i] = Raw4<T>::load(x + (size_t)row * H + min(lane + 64 * i, nch - 1) * 4);
  if (res) {
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) rr[i] = Raw4<T>::load(res + (size_t)row * H + min(lane + 64 * i, nch - 1) * 4);
  }
  float v[LN_MAXC][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    Raw4<T>::to_f(rx[i], v[i]);
    if (thresh && !drop_after) {
      bool kp4[4];
      drop_keep4(hrow, drop_col_hash(seed, (uint32_t)c), thresh, kp4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = kp4[e] ? v[i][e] * keep_scale : 0.f;
    }
    if (res) {
      float r[4];
      Raw4<T>::to_f(rr[i], r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] += r[e];
    }
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float g[4], b[4], o[4];
      Vec4<float>::load(gamma + c * 4, g);
      Vec4<float>::load(beta + c * 4, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (thresh && drop_after) {          // y = dropout(LN(x + res)): the embedding tails, model/model.py:331-333,361-363
        bool kp4[4];
        drop_keep4(hrow, drop_col_hash(seed, (uint32_t)c), thresh, kp4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = kp4[e] ? o[e] * keep_scale : 0.f;
      }
      Vec4<T>::store(y + (size_t)row * H + c * 4, o);
    }
  }
}

// backward.  dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//   dres = dz ;  dx = dz * keep / (1-p)
//   dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy   (per-workgroup partials -> ws, then reduced)
template <typename T, int NC, bool Q = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int M, int H, const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr,
                                                     uint64_t seed_imm, T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ ws, int want_dbias,
                                                     int drop_after, uint8_t* __restrict__ qo, const unsigned* __restrict__ q_prev,
                                                     unsigned* __restrict__ q_next, unsigned* __restrict__ q_clear, float* __restrict__ q_scale_out) {
  __shared__ float red[4][NC * 4 * 64];
  // Q: also an e4m3 copy of dx (the gradient the dense layer's input-gradient GEMM reads) -- fp8 mode, delayed scaling (common.h)
  const float qs = Q ? fp8_delayed_scale(q_prev) : 0.f;
  float qmax = 0.f;
  if (Q && blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { q_clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *q_scale_out = qs; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nch = H >> 2;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  float dg[NC][4], db[NC][4], dbx[NC][4];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; dbx[i][e] = 0.f; }

  // gamma stays in registers; per row every load is issued before the first use (column chunks past the row end
  // are clamped, not branched around: with a branch per chunk hipcc waits for each load before issuing the next,
  // nine dependent memory round trips per row)
  float gmr[NC][4];
  uint32_t hcol[NC];                                   // the column part of the dropout variate: the same for every row of the lane
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = min(lane + 64 * i, nch - 1);
    Vec4<float>::load(gamma + c * 4, gmr[i]);
    hcol[i] = drop_col_hash(seed, (uint32_t)(lane + 64 * i));
  }
  typedef typename Raw4<T>::type raw_t;
  for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
    raw_t rx[NC], rr[NC], rd[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const size_t off = (size_t)row * H + min(lane + 64 * i, nch - 1) * 4;
      rx[i] = Raw4<T>::load_nt(x + off);
      rd[i] = Raw4<T>::load_nt(dy + off);
    }
    if (res) {
#pragma unroll
      for (int i = 0; i < NC; ++i) rr[i] = Raw4<T>::load_nt(res + (size_t)row * H + min(lane + 64 * i, nch - 1) * 4);
    }
    const float mean = mean_i[row], rstd = rstd_i[row];
    const uint32_t hrow = drop_row_hash(seed, (uint32_t)row);
    // (fetching the next row into a second register set during the arithmetic changed nothing: 169 vs 172 us)
    float xh[NC][4], g[NC][4];
    bool kp[NC][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      const bool valid = c < nch;
      const size_t off = (size_t)row * H + c * 4;
      float xv[4], dyv[4];
      Raw4<T>::to_f(rx[i], xv);
      Raw4<T>::to_f(rd[i], dyv);
#pragma unroll
      for (int e = 0; e < 4; ++e) kp[i][e] = true;
      if (thresh) {
        drop_keep4(hrow, hcol[i], thresh, kp[i]);
        if (drop_after == 1) {             // mask sits on the output: dy_eff = keep ? dy / (1-p) : 0, x is not masked
#pragma unroll
          for (int e = 0; e < 4; ++e) dyv[e] = kp[i][e] ? dyv[e] * keep_scale : 0.f;
        } else if (drop_after == 0) {      // (2: x is already the normalised sum dropout(dense) + residual, written by
                                           //  uc2_gemm_drop_residual -- only dx below gets the mask)
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] = kp[i][e] ? xv[e] * keep_scale : 0.f;
        }
      }
      if (res) {
        float r[4];
        Raw4<T>::to_f(rr[i], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] += r[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dyq = valid ? dyv[e] : 0.f;
        xh[i][e] = valid ? (xv[e] - mean) * rstd : 0.f;
        g[i][e] = dyq * gmr[i][e];
        s1 += g[i][e];
        s2 += g[i][e] * xh[i][e];
        dg[i][e] += dyq * xh[i][e];
        db[i][e] += dyq;
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const size_t off = (size_t)row * H + c * 4;
        float dz[4], dxv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dz[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
          dxv[e] = (thresh && drop_after != 1) ? (kp[i][e] ? dz[e] * keep_scale : 0.f) : dz[e];
          dbx[i][e] += dxv[e];
        }
        if (dx) Vec4<T>::store(dx + off, dxv);
        if (Q) {
          qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(dxv[0]), fabsf(dxv[1])), fmaxf(fabsf(dxv[2]), fabsf(dxv[3]))));
          const unsigned r = fp8_pack4_sat(dxv[0] * qs, dxv[1] * qs, dxv[2] * qs, dxv[3] * qs);
          *reinterpret_cast<unsigned*>(qo + off) = r;
        }
        if (dres) Vec4<T>::store_nt(dres + off, dz);           // the residual-path gradient is read several kernels later (EPI_ADD of a dgrad GEMM)
      }
    }
  }
  // reduce the 4 waves of this workgroup, write one partial row per workgroup.  One accumulator kind at a time
  // (12 KiB of LDS instead of 36: the static allocation no longer caps the kernel at 4 workgroups per CU).
  const int nout = want_dbias ? 3 : 2;
  for (int which = 0; which < nout; ++which) {
    if (which) __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        red[wv][(i * 64 + lane) * 4 + e] = which == 0 ? dg[i][e] : (which == 1 ? db[i][e] : dbx[i][e]);
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256)       // red index == column (c*4 + e, c = lane + 64*i)
      ws[(size_t)blockIdx.x * 3 * H + which * H + col] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  }
  if (Q) {                                             // max |dx| of this workgroup's rows -> one atomic, spread over the cells
    __syncthreads();
    qmax = wave_max(qmax);
    if (lane == 0) red[0][wv] = qmax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_cell_raise(q_next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16 rows with H % 8 == 0, H <= 1024: HALF a wave per row and 16-byte accesses (8 elements per lane and chunk, lane l of
// the half owns chunks l, l + 32, ...: one wave-instruction covers 512 contiguous bytes of each of its two rows).  The
// 8-byte accesses of the kernels above moved 4.6-5.0 TB/s; dropout masks are the same function of the element offset.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float v) {            // sum over the 32 lanes of a lane half
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void bf8_to_f(const bf16x8& v, float (&o)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

// Q: also an e4m3 copy of y for the GEMM that consumes it (fp8 mode, delayed scaling: half the scale of the previous use's maximum,
// max |y| accumulated for the next use -- common.h; one atomic per workgroup, spread over the UC2_AMAX_CELLS cells)
struct LnQ { uint8_t* q; const unsigned* amax_prev; unsigned* amax_next; unsigned* amax_clear; float* scale_out; };
template <int NC8, bool Q = false>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(int M, int H, const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, uint32_t thresh, float keep_scale, const uint64_t* __restrict__ seed_ptr,
                                                       uint64_t seed_imm, bf16* __restrict__ y, float* __restrict__ mean_o,
                                                       float* __restrict__ rstd_o, int drop_after, LnQ lq) {
  const float qs = Q ? fp8_delayed_scale(lq.amax_prev) : 0.f;
  float qmax = 0.f;
  if (Q && blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { lq.amax_clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *lq.scale_out = qs; }
  const int l32 = threadIdx.x & 31;
  const int row_raw = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool row_ok = row_raw < M;
  const int row = row_ok ? row_raw : M - 1;            // (both halves of a wave stay in the shuffles; stores are masked)
  const int nch = H >> 3;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const uint32_t hrow = drop_row_hash(seed, (uint32_t)row);
  bf16x8 rx[NC8], rr[NC8];
#pragma unroll
  for (int i = 0; i < NC8; ++i) rx[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + (size_t)row * H + min(l32 + 32 * i, nch - 1) * 8));
  if (res) {
#pragma unroll
    for (int i = 0; i < NC8; ++i) rr[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(res + (size_t)row * H + min(l32 + 32 * i, nch - 1) * 8));
  }
  float v[NC8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC8; ++i) {
    const int c = l32 + 32 * i;
    bf8_to_f(rx[i], v[i]);
    if (thresh && !drop_after) {
      bool k0[4], k1[4];
      drop_keep4(hrow, drop_col_hash(seed, 2u * c), thresh, k0);
      drop_keep4(hrow, drop_col_hash(seed, 2u * c + 1u), thresh, k1);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = k0[e] ? v[i][e] * keep_scale : 0.f; v[i][4 + e] = k1[e] ? v[i][4 + e] * keep_scale : 0.f; }
    }
    if (res) {
      float r[8];
      bf8_to_f(rr[i], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] += r[e];
    }
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = half_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC8; ++i) {
    if (l32 + 32 * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float var = half_sum(q) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (l32 == 0 && row_ok) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NC8; ++i) {
    const int c = l32 + 32 * i;
    if (c < nch && row_ok) {
      float g[8], b[8];
      Vec4<float>::load(gamma + c * 8, *reinterpret_cast<float (*)[4]>(&g[0]));
      Vec4<float>::load(gamma + c * 8 + 4, *reinterpret_cast<float (*)[4]>(&g[4]));
      Vec4<float>::load(beta + c * 8, *reinterpret_cast<float (*)[4]>(&b[0]));
      Vec4<float>::load(beta + c * 8 + 4, *reinterpret_cast<float (*)[4]>(&b[4]));
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (thresh && drop_after) {          // y = dropout(LN(x + res)): the embedding tails, model/model.py:331-333,361-363
        bool k0[4], k1[4];
        drop_keep4(hrow, drop_col_hash(seed, 2u * c), thresh, k0);
        drop_keep4(hrow, drop_col_hash(seed, 2u * c + 1u), thresh, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = k0[e] ? o[e] * keep_scale : 0.f; o[4 + e] = k1[e] ? o[4 + e] * keep_scale : 0.f; }
      }
      bf16x8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (bf16)o[e];
      *reinterpret_cast<bf16x8*>(y + (size_t)row * H + c * 8) = ov;
      if (Q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qmax = fmaxf(qmax, fabsf(o[e]));
        const unsigned r0 = fp8_pack4_sat(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs), r1 = fp8_pack4_sat(o[4] * qs, o[5] * qs, o[6] * qs, o[7] * qs);
        *reinterpret_cast<uint2*>(lq.q + (size_t)row * H + c * 8) = make_uint2(r0, r1);
      }
    }
  }
  if (Q) {
    __shared__ float wm[4];
    qmax = wave_max(qmax);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = qmax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_cell_raise(lq.amax_next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
  }
}

// second stage: 64 columns x (a slice of the partial rows) per workgroup; the 4 waves split the slice,
// LDS combine, one atomic per column and slice (gridDim.y slices keep all CUs busy on the 1024 x 3H partials)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(int nblk, int H, int nout, const float* __restrict__ ws,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dbias) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (idx < nout * H)
    for (int b = b0 + wv; b < b1; b += 4) s += ws[(size_t)b * 3 * H + idx];
  red[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && idx < nout * H) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (idx < H) { if (dgamma) atomicAdd(dgamma + idx, s); }
    else if (idx < 2 * H) { if (dbeta) atomicAdd(dbeta + idx - H, s); }
    else { if (dbias) atomicAdd(dbias + idx - 2 * H, s); }
  }
}

static int ln_bwd_blocks(int M) {                       // (an upper bound for both kernels: the 16-byte one takes 8 rows per step)
  int nb = (M + 3) / 4;
  return nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
}

static int ln_fwd_impl(int dtype, int M, int H, const void* x, const void* residual, const float* gamma,
                       const float* beta, float eps, float drop_p, int drop_after, const uint64_t* seed_ptr, uint64_t seed_imm,
                       void* y, float* mean, float* rstd, const LnQ* lqp, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= LN_MAXC * 256);
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG(drop_after == 0 || drop_after == 1);
  if (M <= 0) return 0;
  UC2_CHECK_ARG(x && gamma && beta && y);
  const uint32_t th = drop_thresh(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool v16 = dtype == 1 && (H % 8) == 0 && H <= 1024 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  if (v16) {
    const dim3 g8((M + 7) / 8);
    const int nc8 = (H / 8 + 31) / 32;
    const LnQ lq0 = lqp ? *lqp : LnQ{nullptr, nullptr, nullptr, nullptr, nullptr};
#define LN_FWD16(NCC) do { if (lqp) hipLaunchKernelGGL((ln_fwd16_kernel<NCC, true>), g8, block, 0, st, M, H, (const bf16*)x, (const bf16*)residual, gamma, \
                                         beta, eps, th, ks, seed_ptr, seed_imm, (bf16*)y, mean, rstd, drop_after, lq0); \
                           else hipLaunchKernelGGL((ln_fwd16_kernel<NCC, false>), g8, block, 0, st, M, H, (const bf16*)x, (const bf16*)residual, gamma, \
                                         beta, eps, th, ks, seed_ptr, seed_imm, (bf16*)y, mean, rstd, drop_after, lq0); } while (0)
    if (nc8 == 1) LN_FWD16(1); else if (nc8 == 2) LN_FWD16(2); else if (nc8 == 3) LN_FWD16(3); else LN_FWD16(4);
#undef LN_FWD16
  } else if (lqp) {
    return -2;                                         // only the 16-byte bf16 kernel writes the e4m3 copy
  } else if (dtype == 0)
    hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, st, M, H, (const float*)x, (const float*)residual, gamma,
                       beta, eps, th, ks, seed_ptr, seed_imm, (float*)y, mean, rstd, drop_after);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, st, M, H, (const bf16*)x, (const bf16*)residual, gamma,
                       beta, eps, th, ks, seed_ptr, seed_imm, (bf16*)y, mean, rstd, drop_after);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_ln_fwd(int dtype, int M, int H, const void* x, const void* residual, const float* gamma,
                          const float* beta, float eps, float drop_p, int drop_after, const uint64_t* seed_ptr, uint64_t seed_imm,
                          void* y, float* mean, float* rstd, void* stream) {
  return ln_fwd_impl(dtype, M, H, x, residual, gamma, beta, eps, drop_p, drop_after, seed_ptr, seed_imm, y, mean, rstd, nullptr, stream);
}
// uc2_ln_fwd that also writes q_out = sat_e4m3(y * scale) for the fp8 GEMM that reads y (delayed scaling, cell groups as
// uc2_fp8_quant_delayed).  bf16 rows of H % 8 == 0, H <= 1024, 16-byte aligned; returns -2 (nothing launched) otherwise.
extern "C" int uc2_ln_fwd_q(int dtype, int M, int H, const void* x, const void* residual, const float* gamma,
                            const float* beta, float eps, float drop_p, int drop_after, const uint64_t* seed_ptr, uint64_t seed_imm,
                            void* y, float* mean, float* rstd, void* q_out, const void* amax_prev, void* amax_next, void* amax_clear,
                            float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(q_out && amax_prev && amax_next && amax_clear && q_scale_out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  if (!(dtype == 1 && (H % 8) == 0 && H <= 1024 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)q_out) & 15) == 0)) return -2;
  const LnQ lq{(uint8_t*)q_out, (const unsigned*)amax_prev, (unsigned*)amax_next, (unsigned*)amax_clear, q_scale_out};
  return ln_fwd_impl(dtype, M, H, x, residual, gamma, beta, eps, drop_p, drop_after, seed_ptr, seed_imm, y, mean, rstd, &lq, stream);
}

// The backward kernel is persistent (every workgroup strides over rows with its column sums in registers): launch exactly one
// full round of resident workgroups.  1024 workgroups on 768 resident slots (3 waves per SIMD at H = 768) ran a second round
// on a third of the chip.
template <typename T, int NC>
static int ln_bwd_resident(int want) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)ln_bwd_kernel<T, NC>, 256, 0) == hipSuccess && per_cu > 0)
      slots = per_cu * prop.multiProcessorCount;
    else
      slots = 1024;
  }
  return want < slots ? want : slots;
}

extern "C" size_t uc2_ln_bwd_workspace(int M, int H) { return (size_t)ln_bwd_blocks(M) * 3 * H * sizeof(float); }

// number of workgroups (= partial rows in ws) the backward kernel of (dtype, M, H) is launched with
static int ln_bwd_nblocks(int dtype, int M, int H) {
  const int nb = ln_bwd_blocks(M), nc = (H + 255) / 256;
  if (dtype == 0) return nc == 1 ? ln_bwd_resident<float, 1>(nb) : nc == 2 ? ln_bwd_resident<float, 2>(nb) : nc == 3 ? ln_bwd_resident<float, 3>(nb) : ln_bwd_resident<float, 4>(nb);
  return nc == 1 ? ln_bwd_resident<bf16, 1>(nb) : nc == 2 ? ln_bwd_resident<bf16, 2>(nb) : nc == 3 ? ln_bwd_resident<bf16, 3>(nb) : ln_bwd_resident<bf16, 4>(nb);
}

// first stage only: dx / dres and the per-workgroup partial column sums in ws (want_dbias: also those of dx)
static int ln_bwd_partial_impl(int dtype, int M, int H, const void* dy, const void* x, const void* residual,
                               const float* gamma, const float* mean, const float* rstd, float drop_p, int drop_after,
                               const uint64_t* seed_ptr, uint64_t seed_imm, void* dx, void* dres, int want_dbias, void* ws,
                               void* q_out, const void* q_prev, void* q_next, void* q_clear, float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= LN_MAXC * 256);
  UC2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG(drop_after >= 0 && drop_after <= 2 && !(drop_after == 2 && residual));
  if (M <= 0) return 0;
  UC2_CHECK_ARG(dy && x && gamma && mean && rstd && ws);
  const uint32_t th = drop_thresh(drop_p);
  const float ks = 1.0f / (1.0f - drop_p);
  const int nb = ln_bwd_nblocks(dtype, M, H);
  hipStream_t st = (hipStream_t)stream;
  const int nc = (H + 255) / 256;
#define LN_BWD_LAUNCH(TT, NCC)                                                                                      \
  do {                                                                                                              \
    if (q_out) hipLaunchKernelGGL((ln_bwd_kernel<TT, NCC, true>), dim3(nb), dim3(256), 0, st, M, H, (const TT*)dy, (const TT*)x, \
                     (const TT*)residual, gamma, mean, rstd, th, ks, seed_ptr, seed_imm, (TT*)dx, (TT*)dres,         \
                     (float*)ws, want_dbias ? 1 : 0, drop_after, (uint8_t*)q_out, (const unsigned*)q_prev, (unsigned*)q_next, (unsigned*)q_clear, q_scale_out); \
    else hipLaunchKernelGGL((ln_bwd_kernel<TT, NCC, false>), dim3(nb), dim3(256), 0, st, M, H, (const TT*)dy, (const TT*)x, \
                     (const TT*)residual, gamma, mean, rstd, th, ks, seed_ptr, seed_imm, (TT*)dx, (TT*)dres,         \
                     (float*)ws, want_dbias ? 1 : 0, drop_after, (uint8_t*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (float*)nullptr); \
  } while (0)
  if (dtype == 0) {
    if (nc == 1) LN_BWD_LAUNCH(float, 1); else if (nc == 2) LN_BWD_LAUNCH(float, 2);
    else if (nc == 3) LN_BWD_LAUNCH(float, 3); else LN_BWD_LAUNCH(float, 4);
  } else {
    if (nc == 1) LN_BWD_LAUNCH(bf16, 1); else if (nc == 2) LN_BWD_LAUNCH(bf16, 2);
    else if (nc == 3) LN_BWD_LAUNCH(bf16, 3); else LN_BWD_LAUNCH(bf16, 4);
  }
#undef LN_BWD_LAUNCH
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_ln_bwd_partial(int dtype, int M, int H, const void* dy, const void* x, const void* residual,
                                  const float* gamma, const float* mean, const float* rstd, float drop_p, int drop_after,
                                  const uint64_t* seed_ptr, uint64_t seed_imm, void* dx, void* dres, int want_dbias, void* ws,
                                  void* stream) {
  return ln_bwd_partial_impl(dtype, M, H, dy, x, residual, gamma, mean, rstd, drop_p, drop_after, seed_ptr, seed_imm, dx, dres, want_dbias,
                             ws, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// uc2_ln_bwd_partial that also writes q_out = sat_e4m3(dx * scale): the e4m3 operand of the dense layer's input-gradient GEMM in fp8
// mode (delayed scaling, cell groups as uc2_fp8_quant_delayed).  bf16 only (-2 otherwise, nothing launched); dx must be given.
extern "C" int uc2_ln_bwd_partial_q(int dtype, int M, int H, const void* dy, const void* x, const void* residual,
                                    const float* gamma, const float* mean, const float* rstd, float drop_p, int drop_after,
                                    const uint64_t* seed_ptr, uint64_t seed_imm, void* dx, void* dres, int want_dbias, void* ws,
                                    void* q_out, const void* amax_prev, void* amax_next, void* amax_clear, float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(dx && q_out && amax_prev && amax_next && amax_clear && q_scale_out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  if (dtype != 1 || ((uintptr_t)q_out & 3)) return -2;
  return ln_bwd_partial_impl(dtype, M, H, dy, x, residual, gamma, mean, rstd, drop_p, drop_after, seed_ptr, seed_imm, dx, dres, want_dbias,
                             ws, q_out, amax_prev, amax_next, amax_clear, q_scale_out, stream);
}

// second stage: dgamma / dbeta / dbias += the column sums of the partial rows a uc2_ln_bwd_partial of the same (dtype, M, H) left
// in ws.  Nothing in the backward chain reads these three vectors, so the caller may run this on another stream (ordered after
// the first stage): next to a persistent GEMM of a concurrent stream this small kernel otherwise waits ~100 us for a free CU
// with the whole input-gradient chain queued behind it (profiles/r04_2048pairs_bench_n1_kernel_stats.csv: 121 us against 5.6 us alone).
extern "C" int uc2_ln_bwd_reduce(int dtype, int M, int H, const void* ws, float* dgamma, float* dbeta, float* dbias,
                                 void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= LN_MAXC * 256);
  if (M <= 0 || !(dgamma || dbeta || dbias)) return 0;
  UC2_CHECK_ARG(ws);
  const int nb = ln_bwd_nblocks(dtype, M, H);
  const int nout = dbias ? 3 : 2;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((nout * H + 63) / 64, nb >= 256 ? 16 : 1), dim3(256), 0, (hipStream_t)stream, nb,
                     H, nout, (const float*)ws, dgamma, dbeta, dbias);
  UC2_LAUNCH_CHECK();
  return 0;
}

// The second stage of up to UC2_LN_BATCH_MAX LayerNorm backwards in ONE launch (blockIdx.z = item).  At the reference's micro-batch
// (9 984 tokens) a backward pass runs 28 of these 5.7 us reductions, each a launch of its own on the input-gradient chain
// (profiles/r04_2048pairs_regime_itm_kernel_stats.csv: 1.7 % of the step); nothing reads dgamma / dbeta / dbias before the end of the pass,
// so uc2_amd/ops/kernels.py collects the partial-sum workspaces and reduces them together at the end of the backward pass.
#define UC2_LN_BATCH_MAX 32
struct LnReduceBatch {
  const float* ws[UC2_LN_BATCH_MAX];
  float* dgamma[UC2_LN_BATCH_MAX];
  float* dbeta[UC2_LN_BATCH_MAX];
  float* dbias[UC2_LN_BATCH_MAX];
  int nblk[UC2_LN_BATCH_MAX];
};
__global__ __launch_bounds__(256) void ln_bwd_reduce_batch_kernel(LnReduceBatch b, int H) {
  __shared__ float red[4][64];
  const int it = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int nblk = b.nblk[it];
  const float* __restrict__ ws = b.ws[it];
  float* const dgamma = b.dgamma[it];
  float* const dbeta = b.dbeta[it];
  float* const dbias = b.dbias[it];
  const int nout = dbias ? 3 : 2;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (idx < nout * H)
    for (int r = b0 + wv; r < b1; r += 4) s += ws[(size_t)r * 3 * H + idx];
  red[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && idx < nout * H) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (idx < H) { if (dgamma) atomicAdd(dgamma + idx, s); }
    else if (idx < 2 * H) { if (dbeta) atomicAdd(dbeta + idx - H, s); }
    else atomicAdd(dbias + idx - 2 * H, s);
  }
}

struct Uc2LnReduceItem { int M; const void* ws; float* dgamma; float* dbeta; float* dbias; };      // mirrors include/uc2_hip.h
extern "C" int uc2_ln_bwd_reduce_batch(int dtype, int n, const Uc2LnReduceItem* items, int H, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= LN_MAXC * 256);
  UC2_CHECK_ARG(n >= 0 && n <= UC2_LN_BATCH_MAX && (n == 0 || items));
  LnReduceBatch b;
  int k = 0, nb_max = 0;
  for (int i = 0; i < n; ++i) {
    if (items[i].M <= 0 || !(items[i].dgamma || items[i].dbeta || items[i].dbias)) continue;
    UC2_CHECK_ARG(items[i].ws);
    b.ws[k] = (const float*)items[i].ws;
    b.dgamma[k] = items[i].dgamma; b.dbeta[k] = items[i].dbeta; b.dbias[k] = items[i].dbias;
    b.nblk[k] = ln_bwd_nblocks(dtype, items[i].M, H);
    nb_max = b.nblk[k] > nb_max ? b.nblk[k] : nb_max;
    ++k;
  }
  if (k == 0) return 0;
  for (int i = k; i < UC2_LN_BATCH_MAX; ++i) { b.ws[i] = nullptr; b.dgamma[i] = b.dbeta[i] = b.dbias[i] = nullptr; b.nblk[i] = 0; }
  hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((3 * H + 63) / 64, nb_max >= 256 ? 16 : 1, k), dim3(256), 0, (hipStream_t)stream, b, H);
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_ln_bwd(int dtype, int M, int H, const void* dy, const void* x, const void* residual,
                          const float* gamma, const float* mean, const float* rstd, float drop_p, int drop_after,
                          const uint64_t* seed_ptr, uint64_t seed_imm, void* dx, void* dres, float* dgamma, float* dbeta,
                          float* dbias, void* ws, void* stream) {
  const int rc = uc2_ln_bwd_partial(dtype, M, H, dy, x, residual, gamma, mean, rstd, drop_p, drop_after, seed_ptr, seed_imm, dx,
                                    dres, dbias != nullptr, ws, stream);
  if (rc != 0) return rc;
  return uc2_ln_bwd_reduce(dtype, M, H, ws, dgamma, dbeta, dbias, stream);
}
