// fp8 (OCP e4m3) GEMM inputs for the forward and input-gradient GEMMs of the encoder layer (BASELINE.json configs[4]:
// "fp8 MFMA GEMMs"; the reference has no fp8 path -- apex amp O2 is fp16 -- so this is an extension of the bf16 mode
// with the same interfaces).  Per-tensor power-of-two scaling:
//   amax   = max |x|                                   (uc2_fp8_amax, atomic max on the bit pattern)
//   scale  = 2^floor(log2(448 / amax))  (1 if amax = 0) (uc2_fp8_scale: on the device, no host round trip)
//   x8     = sat_e4m3(x * scale)                        (uc2_fp8_quant / uc2_fp8_quant_t)
//   y      = (x8 . w8) / (scale_x * scale_w)            (uc2_gemm_fp8: v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate)
// Weight gradients stay bf16 (uc2_gemm).
#include "gemm_common.h"
#include <hip/hip_fp8.h>

template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(size_t n, const T* __restrict__ x, unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float v[4];
    Vec4<T>::load(x + i * 4, v);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(to_f<T>(x[(n4 << 2) + threadIdx.x])));
  // one atomic per workgroup: with one per wave, 8192 atomics on a single address serialised into ~100 us
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    amax_cell_raise(amax_bits, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}
__global__ void fp8_scale_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scale) { *scale = fp8_scale_of(*amax_bits); }
__device__ __forceinline__ unsigned pack4_e4m3(const float (&v)[4]) {
  return fp8_pack4_sat(v[0], v[1], v[2], v[3]);
}
// rows x cols (cols % 4 == 0), out[r * ldo + c] = e4m3(x[r * ldx + c] * scale)
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(int rows, int cols, const T* __restrict__ x, int ldx,
                                                        const float* __restrict__ scale, uint8_t* __restrict__ out, int ldo,
                                                        const unsigned* __restrict__ amax_bits, float* __restrict__ scale_out) {
  const float s = amax_bits ? fp8_scale_of(*amax_bits) : *scale;            // (amax given: the scale is derived here and published)
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const int c4 = cols >> 2;
  const size_t total = (size_t)rows * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), c = (int)(i - (size_t)r * c4) * 4;
    float v[4];
    Vec4<T>::load(x + (size_t)r * ldx + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= s;
    *reinterpret_cast<unsigned*>(out + (size_t)r * ldo + c) = pack4_e4m3(v);
  }
}
// transposing variant: out[c * ldo + r] = e4m3(x[r * ldx + c] * scale)  (the k-contiguous copy of W^T for the input
// gradient dX = dY W); 64 x 64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_t_kernel(int rows, int cols, const T* __restrict__ x, int ldx,
                                                          const float* __restrict__ scale, uint8_t* __restrict__ out, int ldo,
                                                          const unsigned* __restrict__ amax_bits, float* __restrict__ scale_out) {
  __shared__ float tile[64][65];
  const float s = amax_bits ? fp8_scale_of(*amax_bits) : *scale;
  if (scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *scale_out = s;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, t = threadIdx.x;
  for (int i = t; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? to_f<T>(x[(size_t)(r0 + r) * ldx + c0 + c]) * s : 0.f;
  }
  __syncthreads();
  for (int i = t; i < 64 * 16; i += 256) {
    const int c = i >> 4, r4 = (i & 15) * 4;                     // output row c0 + c, output columns r0 + r4 .. +3
    if (c0 + c < cols && r0 + r4 < rows) {
      const float v[4] = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
      *reinterpret_cast<unsigned*>(out + (size_t)(c0 + c) * ldo + r0 + r4) = pack4_e4m3(v);
    }
  }
}

// Delayed scaling, one pass: x8 = sat_e4m3(x * scale(amax_prev) / 2) while max |x| of THIS tensor is accumulated into amax_next
// (one atomic per workgroup, spread over UC2_AMAX_CELLS cells) for the next use of the same tensor role, and a third group of cells
// is cleared for the use after that (nobody else touches it during this launch).  amax_prev / amax_next / amax_clear: UC2_AMAX_CELLS
// (16) unsigned each; a maximum = the maximum over a group's cells.  Half the just-in-time scale: values up to twice the previous maximum stay representable
// (e4m3 saturates beyond).  Replaces the amax pass + the quantisation pass (two reads of x) by one read.
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_delayed_kernel(int rows, int cols, const T* __restrict__ x, int ldx,
                                                                const unsigned* __restrict__ amax_prev, unsigned* __restrict__ amax_next,
                                                                unsigned* __restrict__ amax_clear, float* __restrict__ scale_out,
                                                                uint8_t* __restrict__ out, int ldo) {
  const float s = fp8_delayed_scale(amax_prev);
  if (blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { amax_clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *scale_out = s; }
  const int c4 = cols >> 2;
  const size_t total = (size_t)rows * c4;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), c = (int)(i - (size_t)r * c4) * 4;
    float v[4];
    Vec4<T>::load(x + (size_t)r * ldx + c, v);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= s;
    *reinterpret_cast<unsigned*>(out + (size_t)r * ldo + c) = pack4_e4m3(v);
  }
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax_cell_raise(amax_next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}
// the same pass for a dense bf16 tensor (ldx == ldo == cols, cols % 16 == 0, 16-byte aligned): 16 values per thread and step (two
// 16-byte loads, one 16-byte store) on a flat index -- the general kernel above moves 8 / 4 bytes per lane and divides a 64-bit index
// per step: 150 us for the 133 120 x 1024 activations of a uc2-large layer = 2.7 TB/s (profiles/r06_experiments.md section 5)
__global__ __launch_bounds__(256) void fp8_quant_delayed16_kernel(size_t n16, const bf16* __restrict__ x, const unsigned* __restrict__ amax_prev,
                                                                  unsigned* __restrict__ amax_next, unsigned* __restrict__ amax_clear,
                                                                  float* __restrict__ scale_out, uint8_t* __restrict__ out) {
  const float s = fp8_delayed_scale(amax_prev);
  if (blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { amax_clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *scale_out = s; }
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x) + 2 * i);
    const bf16x8 b = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x) + 2 * i + 1);
    float v[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (float)a[e]; v[8 + e] = (float)b[e]; }
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[e]));
    uint4 o;
    o.x = fp8_pack4_sat(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    o.y = fp8_pack4_sat(v[4] * s, v[5] * s, v[6] * s, v[7] * s);
    o.z = fp8_pack4_sat(v[8] * s, v[9] * s, v[10] * s, v[11] * s);
    o.w = fp8_pack4_sat(v[12] * s, v[13] * s, v[14] * s, v[15] * s);
    reinterpret_cast<uint4*>(out)[i] = o;
  }
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax_cell_raise(amax_next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}
extern "C" int uc2_fp8_quant_delayed(int dtype, int rows, int cols, const void* x, int ldx, const void* amax_prev, void* amax_next,
                                     void* amax_clear, float* scale_out, void* out, int ldo, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (rows <= 0 || cols <= 0) return 0;
  UC2_CHECK_ARG(x && amax_prev && amax_next && amax_clear && scale_out && out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  UC2_CHECK_ARG((cols & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0);
  const size_t blocks = ((size_t)rows * (cols / 4) + 255) / 256;
  const int grid = (int)(blocks > 2048 ? 2048 : blocks);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1 && ldx == cols && ldo == cols && (cols & 15) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    const size_t n16 = (size_t)rows * cols / 16, b16 = (n16 + 255) / 256;
    hipLaunchKernelGGL(fp8_quant_delayed16_kernel, dim3((int)(b16 > 2048 ? 2048 : b16)), dim3(256), 0, st, n16, (const bf16*)x, (const unsigned*)amax_prev,
                       (unsigned*)amax_next, (unsigned*)amax_clear, scale_out, (uint8_t*)out);
    UC2_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == 0) hipLaunchKernelGGL(fp8_quant_delayed_kernel<float>, dim3(grid), dim3(256), 0, st, rows, cols, (const float*)x, ldx, (const unsigned*)amax_prev, (unsigned*)amax_next, (unsigned*)amax_clear, scale_out, (uint8_t*)out, ldo);
  else hipLaunchKernelGGL(fp8_quant_delayed_kernel<bf16>, dim3(grid), dim3(256), 0, st, rows, cols, (const bf16*)x, ldx, (const unsigned*)amax_prev, (unsigned*)amax_next, (unsigned*)amax_clear, scale_out, (uint8_t*)out, ldo);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---- all e4m3 weight copies of a model in a few launches --------------------------------------------------------------------------
// After every optimizer step each layer weight W [rows, cols] (fp32 master copy) is quantised again, in both orientations (W for
// the forward GEMM, W^T k-contiguous for the input-gradient GEMM), with its own just-in-time power-of-two scale: per weight one
// maximum pass and two quantisation passes = 288 launches of ~15 us per uc2-large step when issued one by one.  Here: per batch of up
// to 32 weights one launch that clears the maxima, one that accumulates them (a workgroup per 64 x 64 tile of any of the weights)
// and one that reads each tile once and writes it twice (row-major and, through LDS, transposed).
#define UC2_FP8_WB_MAX 32
struct Fp8WBatch {
  int n; int tile0[UC2_FP8_WB_MAX + 1]; int rows[UC2_FP8_WB_MAX], cols[UC2_FP8_WB_MAX];
  const float* src[UC2_FP8_WB_MAX]; uint8_t* out[UC2_FP8_WB_MAX]; uint8_t* out_t[UC2_FP8_WB_MAX]; unsigned* amax[UC2_FP8_WB_MAX];
  float* scale[UC2_FP8_WB_MAX];
};
__device__ __forceinline__ int fp8_wb_item(const Fp8WBatch& b, int blk) {
  int i = 0;
  for (int k = 1; k < UC2_FP8_WB_MAX; ++k) if (k < b.n && blk >= b.tile0[k]) i = k;     // (uniform)
  return i;
}
__global__ void fp8_wb_clear_kernel(Fp8WBatch b) { if ((int)threadIdx.x < b.n) *b.amax[threadIdx.x] = 0u; }
__global__ __launch_bounds__(256) void fp8_wb_amax_kernel(Fp8WBatch b) {
  __shared__ float wm[4];
  const int i = fp8_wb_item(b, blockIdx.x), t = blockIdx.x - b.tile0[i];
  const int tc = b.cols[i] / 64, tr_ = t / tc, tcx = t - tr_ * tc;
  const float* s = b.src[i] + (size_t)(tr_ * 64) * b.cols[i] + tcx * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(s + (size_t)(ty + 16 * r) * b.cols[i] + tx * 4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax_cell_raise(b.amax[i], fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}
__global__ __launch_bounds__(256) void fp8_wb_quant_kernel(Fp8WBatch b) {
  __shared__ float tile[64][65];
  const int i = fp8_wb_item(b, blockIdx.x), t = blockIdx.x - b.tile0[i];
  const int rows = b.rows[i], cols = b.cols[i];
  const int tc = cols / 64, tr_ = t / tc, tcx = t - tr_ * tc;
  const float sc = fp8_scale_of(*b.amax[i]);
  if (t == 0 && threadIdx.x == 0) *b.scale[i] = sc;
  const float* s = b.src[i] + (size_t)(tr_ * 64) * cols + tcx * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ty + 16 * r;
    const float4 v4 = *reinterpret_cast<const float4*>(s + (size_t)row * cols + tx * 4);
    const float v[4] = {v4.x * sc, v4.y * sc, v4.z * sc, v4.w * sc};
    if (b.out[i]) *reinterpret_cast<unsigned*>(b.out[i] + (size_t)(tr_ * 64 + row) * cols + tcx * 64 + tx * 4) = pack4_e4m3(v);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row][tx * 4 + e] = v[e];
  }
  if (!b.out_t[i]) return;                             // (uniform)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = ty + 16 * r;                       // source column = row of W^T
    const float v[4] = {tile[tx * 4][col], tile[tx * 4 + 1][col], tile[tx * 4 + 2][col], tile[tx * 4 + 3][col]};
    *reinterpret_cast<unsigned*>(b.out_t[i] + (size_t)(tcx * 64 + col) * rows + tr_ * 64 + tx * 4) = pack4_e4m3(v);
  }
}
struct Uc2Fp8WeightItem { const float* w; int rows, cols; void* out; void* out_t; void* amax; float* scale; };   // mirrors include/uc2_hip.h
extern "C" int uc2_fp8_quant_weights_batch(int n, const Uc2Fp8WeightItem* items, void* stream) {
  UC2_CHECK_ARG(n >= 0 && (n == 0 || items));
  for (int k = 0; k < n; ++k) {
    const Uc2Fp8WeightItem& it = items[k];
    UC2_CHECK_ARG(it.w && it.amax && it.scale && (it.out || it.out_t) && it.rows > 0 && it.cols > 0);
    if ((it.rows % 64) || (it.cols % 64) || ((uintptr_t)it.w & 15) || ((uintptr_t)it.out & 3) || ((uintptr_t)it.out_t & 3)) return -2;   // nothing launched
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += UC2_FP8_WB_MAX) {
    Fp8WBatch b{};
    b.n = n - i0 < UC2_FP8_WB_MAX ? n - i0 : UC2_FP8_WB_MAX;
    int tiles = 0;
    for (int k = 0; k < b.n; ++k) {
      const Uc2Fp8WeightItem& it = items[i0 + k];
      b.tile0[k] = tiles; b.rows[k] = it.rows; b.cols[k] = it.cols; b.src[k] = it.w; b.out[k] = (uint8_t*)it.out; b.out_t[k] = (uint8_t*)it.out_t;
      b.amax[k] = (unsigned*)it.amax; b.scale[k] = it.scale;
      tiles += (it.rows / 64) * (it.cols / 64);
    }
    b.tile0[b.n] = tiles;
    hipLaunchKernelGGL(fp8_wb_clear_kernel, dim3(1), dim3(64), 0, st, b);
    hipLaunchKernelGGL(fp8_wb_amax_kernel, dim3(tiles), dim3(256), 0, st, b);
    hipLaunchKernelGGL(fp8_wb_quant_kernel, dim3(tiles), dim3(256), 0, st, b);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}

// amax_bits must be zeroed by the caller (uc2_fp8_amax accumulates a maximum, so several tensors can share one scale)
extern "C" int uc2_fp8_amax(int dtype, size_t n, const void* x, void* amax_bits, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n == 0) return 0;
  UC2_CHECK_ARG(x && amax_bits);
  const size_t blocks = (n / 4 + 255) / 256;
  const int grid = (int)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
  if (dtype == 0) hipLaunchKernelGGL(fp8_amax_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (const float*)x, (unsigned*)amax_bits);
  else hipLaunchKernelGGL(fp8_amax_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (const bf16*)x, (unsigned*)amax_bits);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_fp8_scale(const void* amax_bits, float* scale, void* stream) {
  UC2_CHECK_ARG(amax_bits && scale);
  hipLaunchKernelGGL(fp8_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)amax_bits, scale);
  UC2_LAUNCH_CHECK();
  return 0;
}
static int fp8_quant_impl(int dtype, int rows, int cols, const void* x, int ldx, const float* scale, void* out, int ldo,
                          int transpose, const unsigned* amax_bits, float* scale_out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (rows <= 0 || cols <= 0) return 0;
  UC2_CHECK_ARG(x && (scale || amax_bits) && out);
  hipStream_t st = (hipStream_t)stream;
  if (!transpose) {
    UC2_CHECK_ARG((cols & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0);
    const size_t blocks = ((size_t)rows * (cols / 4) + 255) / 256;
    const int grid = (int)(blocks > 4096 ? 4096 : blocks);
    if (dtype == 0) hipLaunchKernelGGL(fp8_quant_kernel<float>, dim3(grid), dim3(256), 0, st, rows, cols, (const float*)x, ldx, scale, (uint8_t*)out, ldo, amax_bits, scale_out);
    else hipLaunchKernelGGL(fp8_quant_kernel<bf16>, dim3(grid), dim3(256), 0, st, rows, cols, (const bf16*)x, ldx, scale, (uint8_t*)out, ldo, amax_bits, scale_out);
  } else {
    UC2_CHECK_ARG((rows & 3) == 0 && (ldo & 3) == 0);
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (dtype == 0) hipLaunchKernelGGL(fp8_quant_t_kernel<float>, grid, dim3(256), 0, st, rows, cols, (const float*)x, ldx, scale, (uint8_t*)out, ldo, amax_bits, scale_out);
    else hipLaunchKernelGGL(fp8_quant_t_kernel<bf16>, grid, dim3(256), 0, st, rows, cols, (const bf16*)x, ldx, scale, (uint8_t*)out, ldo, amax_bits, scale_out);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_fp8_quant(int dtype, int rows, int cols, const void* x, int ldx, const float* scale, void* out, int ldo,
                             int transpose, void* stream) {
  return fp8_quant_impl(dtype, rows, cols, x, ldx, scale, out, ldo, transpose, nullptr, nullptr, stream);
}
extern "C" int uc2_fp8_quant_amax(int dtype, int rows, int cols, const void* x, int ldx, const void* amax_bits, float* scale_out,
                                  void* out, int ldo, int transpose, void* stream) {
  UC2_CHECK_ARG(amax_bits && scale_out);
  return fp8_quant_impl(dtype, rows, cols, x, ldx, nullptr, out, ldo, transpose, (const unsigned*)amax_bits, scale_out, stream);
}

int uc2_gemm_fp8_launch(const GemmArgs& p8, hipStream_t st);      // gemm_fast.hip
bool uc2_gemm_pp8_supported(const GemmArgs& p);                   // gemm_pp8.hip (takes the "bf16 view": K, lda, ldb halved)
void uc2_gemm_pp8_launch(const GemmArgs& p, hipStream_t st);
extern "C" int uc2_colsum_accum(int dtype, int M, int N, const void* X, int ldx, const uint8_t* rowmask, float* out, void* stream);

// C[M,N] (bf16) = epi( (sum_k A8(m,k) B8(n,k)) / (*scale_a * *scale_b) + bias[n] ); A8 [M,K], B8 [N,K] e4m3, k-contiguous
extern "C" int uc2_gemm_fp8(int M, int N, int K, const void* A8, int lda, const void* B8, int ldb, const float* scale_a,
                            const float* scale_b, void* C, int ldc, const float* bias, int epilogue, const void* aux_in,
                            void* aux_out, int ldaux, int flags, void* stream) {
  UC2_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && (K % 128) == 0);
  UC2_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_TANH);
  UC2_CHECK_ARG(!((epilogue == EPI_DGELU || epilogue == EPI_ADD) && aux_in == nullptr));
  if (M == 0 || N == 0) return 0;
  UC2_CHECK_ARG(A8 && B8 && C && scale_a && scale_b);
  UC2_CHECK_ARG((lda % 16) == 0 && (ldb % 16) == 0 && (((uintptr_t)A8 | (uintptr_t)B8) & 15) == 0);
  GemmArgs p{};
  p.A = A8; p.B = B8; p.C = C; p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux ? ldaux : ldc;
  p.epi = epilogue; p.c_f32 = 0; p.accumulate = 0; p.split_k = 1; p.atomic = 0; p.partial = nullptr;
  p.a_vec = 1; p.b_vec = 1; p.aux_deriv = (flags >> 1) & 1; p.alpha = 1.0f;
  p.variant = (flags & 4) ? 1 : 0;                    // UC2_GEMM_FP8_RING: the LDS-DMA ring kernel even where the ping-pong kernel would take the shape
  p.alpha_dev = scale_a; p.alpha_dev2 = scale_b;
  const int route = uc2_gemm_fp8_launch(p, (hipStream_t)stream);
  UC2_LAUNCH_CHECK();
  if (epilogue == EPI_DGELU && aux_out != nullptr && route != 2) {       // (the ping-pong kernel's epilogue produced the column sums)
    return uc2_colsum_accum(1, M, N, C, ldc, nullptr, reinterpret_cast<float*>(aux_out), stream);
  }
  return 0;
}

// The e4m3 form of uc2_gemm_drop_residual: C (bf16) = dropout_p(A8 W8^T / (scale_a scale_w) + bias) + residual, the pre-LayerNorm sum
// of the encoder's dense -> dropout -> add tails, mask = the LayerNorm kernels' counter-based mask (common.h drop_keep4) for
// (seed, row, column).  Only the ping-pong kernel has this epilogue: -2 (nothing launched) for shapes it does not take -- the caller
// then runs uc2_gemm_fp8 and leaves dropout + residual to uc2_ln_fwd.  p_drop == 0: the plain residual add.
extern "C" int uc2_gemm_fp8_drop_residual(int M, int N, int K, const void* A8, int lda, const void* W8, int ldw, const float* scale_a,
                                          const float* scale_w, void* C, int ldc, const float* bias, const void* residual, int ldres,
                                          float p_drop, const uint64_t* seed_ptr, uint64_t seed_imm, void* stream) {
  UC2_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && (K % 128) == 0);
  UC2_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f);
  if (M == 0 || N == 0) return 0;
  UC2_CHECK_ARG(A8 && W8 && C && residual && scale_a && scale_w);
  UC2_CHECK_ARG((lda % 16) == 0 && (ldw % 16) == 0 && (((uintptr_t)A8 | (uintptr_t)W8) & 15) == 0);
  GemmArgs p{};
  p.A = A8; p.B = W8; p.C = C; p.bias = bias; p.aux_in = residual; p.aux_out = nullptr;
  p.M = M; p.N = N; p.K = K / 2; p.lda = lda / 2; p.ldb = ldw / 2; p.ldc = ldc; p.ldaux = ldres;
  p.epi = p_drop > 0.f ? EPI_DROPADD : EPI_ADD; p.split_k = 1; p.a_vec = 1; p.b_vec = 1; p.alpha = 1.0f;
  p.alpha_dev = scale_a; p.alpha_dev2 = scale_w;
  p.drop_thresh = drop_thresh(p_drop); p.drop_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  p.drop_seed_ptr = seed_ptr; p.drop_seed_imm = seed_imm;
  if (!uc2_gemm_pp8_supported(p)) return -2;
  uc2_gemm_pp8_launch(p, (hipStream_t)stream);
  UC2_LAUNCH_CHECK();
  return 0;
}

// uc2_gemm_fp8 whose epilogue ALSO writes the e4m3 copy of its output that the next GEMM reads (FFN1 -> FFN2 forward: gelu(.);
// FFN2 -> FFN1 input gradients: dY W x gelu') -- no separate quantisation pass over the [rows, 4H] tensor.  Delayed scaling like
// uc2_fp8_quant_delayed: q_out = sat_e4m3(out * scale(*amax_prev) / 2), max |out| -> *amax_next, *amax_clear = 0, the scale used ->
// *q_scale_out.  Only the ping-pong kernel has this epilogue: returns -2 (nothing launched) for shapes it does not take and for
// epilogues other than GELU / DGELU with UC2_GEMM_AUX_DERIV -- the caller then runs uc2_gemm_fp8 + a quantisation pass.
extern "C" int uc2_gemm_fp8_q(int M, int N, int K, const void* A8, int lda, const void* B8, int ldb, const float* scale_a,
                              const float* scale_b, void* C, int ldc, const float* bias, int epilogue, const void* aux_in,
                              void* aux_out, int ldaux, int flags, void* q_out, int ldq, const void* amax_prev, void* amax_next,
                              void* amax_clear, float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(M > 0 && N > 0 && K > 0 && (K % 128) == 0);
  UC2_CHECK_ARG(A8 && B8 && C && scale_a && scale_b && q_out && amax_prev && amax_next && amax_clear && q_scale_out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  UC2_CHECK_ARG((lda % 16) == 0 && (ldb % 16) == 0 && (((uintptr_t)A8 | (uintptr_t)B8) & 15) == 0);
  UC2_CHECK_ARG((ldq % 16) == 0 && ((uintptr_t)q_out & 15) == 0);
  if (!((epilogue == EPI_GELU || epilogue == EPI_DGELU) && (flags & 2))) return -2;
  UC2_CHECK_ARG(!(epilogue == EPI_DGELU && aux_in == nullptr) && !(epilogue == EPI_GELU && aux_out == nullptr));
  GemmArgs p{};
  p.A = A8; p.B = B8; p.C = C; p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
  p.M = M; p.N = N; p.K = K / 2; p.lda = lda / 2; p.ldb = ldb / 2; p.ldc = ldc; p.ldaux = ldaux ? ldaux : ldc;
  p.epi = epilogue; p.split_k = 1; p.a_vec = 1; p.b_vec = 1; p.aux_deriv = 1; p.alpha = 1.0f;
  p.alpha_dev = scale_a; p.alpha_dev2 = scale_b;
  p.q_out = q_out; p.ldq = ldq; p.q_amax_prev = (const unsigned*)amax_prev; p.q_amax_next = (unsigned*)amax_next;
  p.q_amax_clear = (unsigned*)amax_clear; p.q_scale_out = q_scale_out;
  if (!uc2_gemm_pp8_supported(p)) return -2;
  uc2_gemm_pp8_launch(p, (hipStream_t)stream);
  UC2_LAUNCH_CHECK();
  return 0;
}
