// Shared device helpers for the uc2 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define UC2_WAVE 64

// ---- error convention (include/uc2_hip.h): 0 ok, <0 argument error, >0 hipError_t ----
#define UC2_CHECK_ARG(cond) do { if (!(cond)) { uc2_set_error(__FILE__, __LINE__, #cond); return -1; } } while (0)
#define UC2_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e__)); return (int)e__; } } while (0)
extern "C" void uc2_set_error(const char* file, int line, const char* what);

// ---- dtype helpers ----
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// 4 consecutive elements as floats (pointer must be 4-element aligned)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Vec4<bf16> {
  static __device__ __forceinline__ void load(const bf16* p, float (&o)[4]) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
  }
  static __device__ __forceinline__ void store(bf16* p, const float (&o)[4]) {
    bf16x4 v; v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
    *reinterpret_cast<bf16x4*>(p) = v;
  }
};

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- activations ----
__device__ __forceinline__ float gelu_f(float x) {            // reference model/layer.py:31-37 (erf form)
  return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Cheap forms for the bf16 kernels' fused epilogues (results are rounded to bf16, 8 mantissa bits).  The fused
// GELU is ~130 evaluations per lane and output tile, executed with the matrix pipe idle; ocml's erff costs ~45
// VALU instructions per value (more than the whole K = 768 main loop), this form 9:
//     Phi(x) ~= sigmoid(x * (c0 + c1 x^2 + c2 x^4)),  |x| clamped to 9 inside the polynomial,
// an odd-polynomial fit of logit(Phi) (tests/fit_gelu.py): max |Phi error| 4.2e-5, max |gelu error| 2.9e-5 over all
// x, i.e. ~1/100 of a bf16 ulp at |y| ~ 1.  The fp32 (parity) kernels keep erff / tanhf.
#define UC2_PHI_C0 1.5951192f
#define UC2_PHI_C1 0.0739306293f
#define UC2_PHI_C2 (-0.000691440005f)
#define UC2_LOG2E 1.4426950408889634f
__device__ __forceinline__ float phi_bf(float x) {                      // normal CDF
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f), x2 = xc * xc;
  float q = fmaf(-UC2_PHI_C2 * UC2_LOG2E, x2, -UC2_PHI_C1 * UC2_LOG2E);
  q = fmaf(q, x2, -UC2_PHI_C0 * UC2_LOG2E);
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q * xc));
}
__device__ __forceinline__ float gelu_bf(float x) { return x * phi_bf(x); }
__device__ __forceinline__ float dgelu_bf(float x) {                    // Phi(x) + x * pdf(x)
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * (-0.5f * UC2_LOG2E));
  return fmaf(x, pdf, phi_bf(x));
}
__device__ __forceinline__ float tanh_bf(float x) {                     // 1 - 2/(1 + exp(2x)); saturates correctly at +-inf
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (2.0f * UC2_LOG2E))), 1.0f);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16>(float x) { return gelu_bf(x); }
template <typename T> __device__ __forceinline__ float dgelu_t(float x) { return dgelu_f(x); }
template <> __device__ __forceinline__ float dgelu_t<bf16>(float x) { return dgelu_bf(x); }
template <typename T> __device__ __forceinline__ float tanh_t(float x) { return tanhf(x); }
template <> __device__ __forceinline__ float tanh_t<bf16>(float x) { return tanh_bf(x); }

// ---- counter-based dropout RNG: keep(seed, idx) is a pure function, regenerated in backward ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// returns true with probability (1 - p); thresh = p * 2^32 (clamped)
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = mix32(lo ^ (uint32_t)seed);
  h = mix32(h + hi * 0x9E3779B9U + (uint32_t)(seed >> 32));
  return h >= thresh;
}
static inline uint32_t drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}
