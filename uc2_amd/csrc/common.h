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
  static __device__ __forceinline__ void store_nt(float* p, const float (&o)[4]) { store(p, o); }
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
  static __device__ __forceinline__ void store_nt(bf16* p, const float (&o)[4]) {       // streamed: consumed much later, keep it out of the caches
    bf16x4 v; v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>(p));
  }
};

// the same 4 elements kept in their storage type (all loads of a row are issued before the first use)
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  typedef float4 type;
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float4 load_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void to_f(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template <> struct Raw4<bf16> {
  typedef bf16x4 type;
  static __device__ __forceinline__ bf16x4 load(const bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ bf16x4 load_nt(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p)); }
  static __device__ __forceinline__ void to_f(const bf16x4& v, float (&o)[4]) {
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
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
// an odd-polynomial fit of logit(Phi) (tools/fit_gelu.py): max |Phi error| 4.2e-5, max |gelu error| 2.9e-5 over all
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
// GELU and its derivative from ONE evaluation of the fit: with Phi(x) ~= sigma(s(x)), s(x) = x (c0 + c1 x^2 + c2 x^4),
//     gelu(x) = x Phi ,   gelu'(x) = Phi + x Phi' = Phi + x Phi (1 - Phi) s'(x) ,   s'(x) = c0 + 3 c1 x^2 + 5 c2 x^4
// -- no transcendental beyond the forward's own exp2 + rcp (the Gaussian density would cost a second exp2).  Max
// |gelu' error| 1.1e-4 over all x (1/20 of half a bf16 ulp at |gelu'| ~ 1).  The forward epilogue that saves gelu'(pre)
// for the backward (UC2_GEMM_AUX_DERIV) uses this; the backward GEMM then only multiplies.
__device__ __forceinline__ void gelu_and_dgelu_bf(float x, float& g, float& d) {
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f), x2 = xc * xc;
  float q = fmaf(-UC2_PHI_C2 * UC2_LOG2E, x2, -UC2_PHI_C1 * UC2_LOG2E);
  q = fmaf(q, x2, -UC2_PHI_C0 * UC2_LOG2E);
  const float ph = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q * xc));
  const float sp = fmaf(fmaf(5.0f * UC2_PHI_C2, x2, 3.0f * UC2_PHI_C1), x2, UC2_PHI_C0);
  const float t = fmaf(-ph, ph, ph);                  // Phi (1 - Phi)
  g = x * ph;
  d = fmaf(xc * sp, t, ph);
}
// The same arithmetic on a PAIR of values, written on 2-vectors so that hipcc emits v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32
// (two values per full-rate issue slot) instead of v_fmaak / v_fmamk / v_mul per value: the scalar form's 12 full-rate
// instructions per value came out as 11 issue slots + one v_mov (operand packing) in the GELU GEMM's epilogue, where the matrix
// pipe is idle and the vector pipe is the bound; same operations in the same order, bit-identical results.
typedef float uc2_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_dgelu_bf2(uc2_f2v x, uc2_f2v& g, uc2_f2v& d) {
  const uc2_f2v xc = {__builtin_amdgcn_fmed3f(x.x, -9.0f, 9.0f), __builtin_amdgcn_fmed3f(x.y, -9.0f, 9.0f)};
  const uc2_f2v x2 = xc * xc;
  const uc2_f2v k2 = {-UC2_PHI_C2 * UC2_LOG2E, -UC2_PHI_C2 * UC2_LOG2E}, k1 = {-UC2_PHI_C1 * UC2_LOG2E, -UC2_PHI_C1 * UC2_LOG2E};
  const uc2_f2v k0 = {-UC2_PHI_C0 * UC2_LOG2E, -UC2_PHI_C0 * UC2_LOG2E};
  uc2_f2v q = __builtin_elementwise_fma(k2, x2, k1);
  q = __builtin_elementwise_fma(q, x2, k0);
  const uc2_f2v y = q * xc;
  const uc2_f2v e = {__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
  const uc2_f2v one = {1.0f, 1.0f};
  const uc2_f2v den = one + e;
  const uc2_f2v ph = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  const uc2_f2v s5 = {5.0f * UC2_PHI_C2, 5.0f * UC2_PHI_C2}, s3 = {3.0f * UC2_PHI_C1, 3.0f * UC2_PHI_C1}, s0 = {UC2_PHI_C0, UC2_PHI_C0};
  const uc2_f2v sp = __builtin_elementwise_fma(__builtin_elementwise_fma(s5, x2, s3), x2, s0);
  const uc2_f2v t = __builtin_elementwise_fma(-ph, ph, ph);                      // Phi (1 - Phi)
  g = x * ph;
  d = __builtin_elementwise_fma(xc * sp, t, ph);
}
__device__ __forceinline__ uc2_f2v gelu_bf2(uc2_f2v x) {                // gelu_bf on a pair (forward-only epilogues)
  const uc2_f2v xc = {__builtin_amdgcn_fmed3f(x.x, -9.0f, 9.0f), __builtin_amdgcn_fmed3f(x.y, -9.0f, 9.0f)};
  const uc2_f2v x2 = xc * xc;
  const uc2_f2v k2 = {-UC2_PHI_C2 * UC2_LOG2E, -UC2_PHI_C2 * UC2_LOG2E}, k1 = {-UC2_PHI_C1 * UC2_LOG2E, -UC2_PHI_C1 * UC2_LOG2E};
  const uc2_f2v k0 = {-UC2_PHI_C0 * UC2_LOG2E, -UC2_PHI_C0 * UC2_LOG2E};
  uc2_f2v q = __builtin_elementwise_fma(k2, x2, k1);
  q = __builtin_elementwise_fma(q, x2, k0);
  const uc2_f2v y = q * xc;
  const uc2_f2v one = {1.0f, 1.0f};
  const uc2_f2v den = one + uc2_f2v{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
  return x * uc2_f2v{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
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

// ---- running maximum of |x| for the fp8 delayed scaling: UC2_AMAX_CELLS 4-byte cells per tensor role and use (a producer's
// workgroups spread their one atomic each over the cells: thousands of atomics on ONE address serialise at ~12 ns each; readers
// take the maximum over the cells; non-negative floats order like their bit patterns)
#define UC2_AMAX_CELLS 16
__device__ __forceinline__ unsigned amax_cells_read(const unsigned* __restrict__ cells) {       // uniform address: scalar loads
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < UC2_AMAX_CELLS; ++i) m = max(m, cells[i]);
  return m;
}
// max-accumulate a workgroup's |value| maximum into a cell: the atomic only if it would raise the cell.  The 16 cells of a group share
// one 64-byte line, and same-line atomics serialise in the L2 whatever their address: the 16 640 workgroups of a LayerNorm forward over
// 133 120 rows spent 118 of their 209 us queueing on that line (profiles/r06_experiments.md section 5).  A stale (lower) read only costs
// an atomic that was not needed; the cells of a group are never lowered while the group is being accumulated (common protocol above).
__device__ __forceinline__ void amax_cell_raise(unsigned* cell, float m) {
  const unsigned b = __float_as_uint(m);                       // non-negative floats order like their bit patterns
  if (b > __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cell, b);
}
// the power-of-two scale that maps `amax` just below the e4m3 maximum (448)
__device__ __forceinline__ float fp8_scale_of(unsigned amax_bits) {
  const float a = __uint_as_float(amax_bits);
  float s = 1.0f;
  if (a > 0.f && a < INFINITY) s = exp2f(floorf(log2f(448.0f / a)));
  return fminf(fmaxf(s, 1.0f / 16777216.0f), 16777216.0f);
}
// four floats -> four e4m3 bytes, clamped to +-448 first: v_cvt_pk_fp8_f32 turns a finite value beyond the e4m3 range into NaN (0x7f) under
// the default mode, and with delayed scaling a tensor may outgrow twice its previous maximum between two uses
__device__ __forceinline__ unsigned fp8_pack4_sat(float a, float b, float c, float d) {
  unsigned r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.f, 448.f), __builtin_amdgcn_fmed3f(d, -448.f, 448.f), r, true);
  return r;
}
// delayed scaling: half the just-in-time scale of the previous use's maximum (twice that maximum stays representable)
__device__ __forceinline__ float fp8_delayed_scale(const unsigned* __restrict__ prev_cells) { return fp8_scale_of(amax_cells_read(prev_cells)) * 0.5f; }

// ---- counter-based dropout RNG: keep(seed, idx) is a pure function, regenerated in backward ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Hidden-state dropout (the dense -> dropout -> LayerNorm tails, the embedding tails): element (row, column) of a [rows, H] tensor is
// kept when a 16-bit variate of (seed, row, column) is >= thresh >> 16 (drop probability resolved to 2^-16; thresh = p * 2^32,
// see drop_thresh).  The variate is SEPARABLE: one fully mixed 32-bit hash per token row, one per group of four columns (a "piece":
// the 8 bytes of bf16 a lane of every kernel here owns), and per element the top 16 bits of a full-rate 24-bit multiply of their
// XOR with a per-position odd constant -- the scheme of the attention-probability mask below.  Rounds 1-4 hashed (seed, row * H +
// column) per piece with four quarter-rate 32-bit multiplies; that was free in the HBM-bound LayerNorm kernels but cost ~4 us per
// 256 x 256 tile in a GEMM epilogue (uc2_gemm_drop_residual), where a lane meets only 8 rows and 4 pieces per 128 elements.
// Every kernel that applies or regenerates the mask (ln_fwd*, ln_bwd, the EPI_DROPADD epilogue) goes through these three functions.
__device__ __forceinline__ uint32_t drop_row_hash(uint64_t seed, uint32_t row) {
  return mix32((row ^ (uint32_t)seed) + (uint32_t)(seed >> 32) * 0x9E3779B9U);
}
__device__ __forceinline__ uint32_t drop_col_hash(uint64_t seed, uint32_t piece) {
  return mix32(piece * 0x9E3779B9U + (uint32_t)(seed >> 32) + __builtin_rotateleft32((uint32_t)seed, 16));
}
// keep flags of the four columns of a piece
__device__ __forceinline__ void drop_keep4(uint32_t hrow, uint32_t hcol, uint32_t thresh, bool (&keep)[4]) {
  const uint32_t w = hrow ^ hcol, t = thresh & 0xffff0000U;
  keep[0] = __umul24(w, 0x9E3779U) >= t; keep[1] = __umul24(w, 0x85EBCBU) >= t;
  keep[2] = __umul24(w, 0xC2B2AFU) >= t; keep[3] = __umul24(w, 0x27D4EBU) >= t;
}
// Attention-probability dropout: keep(q, k) = light(Hq[q] ^ Hk[k]) with one fully mixed 32-bit hash per query row
// and per key column of a (batch, head) -- L + L full hashes per head instead of L*L; the per-element part is one
// multiply.  (The MFMA backward visits every element twice, once per orientation, and in one of them a lane's
// elements are not index-contiguous, so a shared group hash does not help there.)
__device__ __forceinline__ uint32_t attn_line_hash(uint64_t seed, uint32_t bh, uint32_t idx, uint32_t salt) {
  const uint32_t h = mix32(idx ^ (uint32_t)seed ^ salt);
  return mix32(h + bh * 0x9E3779B9U + (uint32_t)(seed >> 32));
}
#define UC2_ATTN_SALT_Q 0x00000000U
#define UC2_ATTN_SALT_K 0x5bd1e995U
__device__ __forceinline__ bool attn_keep(uint32_t hq, uint32_t hk, uint32_t thresh) {
  // full-rate 24-bit multiply (v_mul_u32_u24; v_mul_lo_u32 is quarter rate and this runs once per score element in
  // VALU-bound kernels): the low 24 bits of two fully mixed hashes, the top 16 bits of the 32-bit product decide
  return __umul24(hq ^ hk, 0x9E3779U) >= (thresh & 0xffff0000U);
}
static inline uint32_t drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// Column sums across the 32 lanes of a lane half, 32 partial sums per lane (index v) -> lane (c = lane & 31, h) returns
// the total of v = c.  Transposing butterfly: at the stage of lane bit `half` a lane keeps the half of its values whose
// index bit `half` matches its own lane bit and adds the partner's copies of those -- 31 exchanges instead of 5 x 32.
// No LDS crossbar (ds_bpermute) involved: bit 4 is v_permlane16_swap (one instruction exchanges both directions of a
// pair), bits 3..0 are DPP operands of the adds (row_ror:8, row_shl/shr:4, quad_perm), each computed only for the
// lanes that keep it.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float colsum_butterfly32(float (&v)[32], int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {                      // lane bit 4
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {                                                   // lane bit 3: partner = lane ^ 8 = row_ror:8 within the 16-lane row
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = v[i] + dpp_f32<0x128>(v[i]), b = v[i + 8] + dpp_f32<0x128>(v[i + 8]);
      v[i] = up ? b : a;
    }
  }
  {                                                   // lane bit 2: lower lanes read lane + 4 (row_shl:4), upper lane - 4 (row_shr:4)
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[i] + dpp_f32<0x104>(v[i]), b = v[i + 4] + dpp_f32<0x114>(v[i + 4]);
      v[i] = up ? b : a;
    }
  }
  {                                                   // lane bit 1: quad_perm [2,3,0,1]
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a = v[i] + dpp_f32<0x4E>(v[i]), b = v[i + 2] + dpp_f32<0x4E>(v[i + 2]);
      v[i] = up ? b : a;
    }
  }
  {                                                   // lane bit 0: quad_perm [1,0,3,2]
    const float a = v[0] + dpp_f32<0xB1>(v[0]), b = v[1] + dpp_f32<0xB1>(v[1]);
    v[0] = (lane & 1) ? b : a;
  }
  return v[0];
}
