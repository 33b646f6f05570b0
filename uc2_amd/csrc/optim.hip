// Multi-tensor AdamW (reference optim/adamw.py:40-103) + global-norm clipping helpers
// (torch clip_grad_norm_ as called at pretrain.py:610).  Pure HBM streaming:
// algorithmic bytes per parameter = 4 (g) + 3*8 (p, m, v read+write) [+ 2 bf16 compute copy].
//
// The host builds a device table of chunks once (one chunk = up to 64Ki contiguous elements
// of one parameter, tagged with its param-group); every step is ONE launch over the table
// with the per-group scalars (lr changes every step) passed by value.  Per element:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= step_size * m / (sqrt(v) + eps)
//   if wd > 0: p -= lr*wd*p            (decay uses the already-updated p, adamw.py:101)
// step_size = lr*sqrt(1-b2^t)/(1-b1^t) is computed on the host in double.
#include "common.h"

#define UC2_ADAM_MAX_GROUPS 8

struct AdamChunk {            // mirrored by uc2_amd/optim/adamw.py (48 bytes)
  float* p; float* g; float* m; float* v; bf16* p16;
  uint32_t n; uint16_t group; uint16_t param;
};
struct AdamGroups {
  float lr[UC2_ADAM_MAX_GROUPS], beta1[UC2_ADAM_MAX_GROUPS], beta2[UC2_ADAM_MAX_GROUPS];
  float eps[UC2_ADAM_MAX_GROUPS], wd[UC2_ADAM_MAX_GROUPS];
  int correct_bias[UC2_ADAM_MAX_GROUPS];
};

// steps[param] holds the number of updates already applied to that parameter (per-parameter, like
// state['step'] at adamw.py:74); active[param] == 0 means "p.grad is None": skipped entirely (adamw.py:52-53).
__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ chunks, AdamGroups gs,
                                                    const int* __restrict__ active, const int* __restrict__ steps,
                                                    const float* __restrict__ gscale, int zero_grad) {
  const AdamChunk c = chunks[blockIdx.x];
  if (active && !active[c.param]) return;
  const int gi = c.group;
  const float lr = gs.lr[gi], b1 = gs.beta1[gi], b2 = gs.beta2[gi], eps = gs.eps[gi], wd = gs.wd[gi];
  float ss = lr;
  if (gs.correct_bias[gi]) {
    const double t = (double)(steps[c.param] + 1);
    ss = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  const float sc = gscale ? *gscale : 1.0f;
  const bool vec = (((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0 &&
                   (((uintptr_t)c.p16 & 7) == 0);
  const uint32_t n4 = vec ? (c.n >> 2) : 0;
  for (uint32_t i = threadIdx.x; i < n4; i += 256) {
    float p[4], g[4], m[4], v[4];
    Vec4<float>::load(c.p + i * 4, p);
    Vec4<float>::load(c.g + i * 4, g);
    Vec4<float>::load(c.m + i * 4, m);
    Vec4<float>::load(c.v + i * 4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = g[e] * sc;
      m[e] = m[e] * b1 + (1.0f - b1) * ge;
      v[e] = v[e] * b2 + (1.0f - b2) * ge * ge;
      p[e] = p[e] - ss * (m[e] / (sqrtf(v[e]) + eps));
      if (wd > 0.f) p[e] = p[e] - lr * wd * p[e];
    }
    Vec4<float>::store(c.p + i * 4, p);
    Vec4<float>::store(c.m + i * 4, m);
    Vec4<float>::store(c.v + i * 4, v);
    if (c.p16) Vec4<bf16>::store(c.p16 + i * 4, p);
    if (zero_grad) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; Vec4<float>::store(c.g + i * 4, z); }
  }
  for (uint32_t i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
    const float ge = c.g[i] * sc;
    const float m = c.m[i] * b1 + (1.0f - b1) * ge;
    const float v = c.v[i] * b2 + (1.0f - b2) * ge * ge;
    float p = c.p[i] - ss * (m / (sqrtf(v) + eps));
    if (wd > 0.f) p = p - lr * wd * p;
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
    if (c.p16) c.p16[i] = (bf16)p;
    if (zero_grad) c.g[i] = 0.f;
  }
}
__global__ void adamw_bump_kernel(int n_params, const int* __restrict__ active, int* __restrict__ steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_params && (!active || active[i])) steps[i] += 1;
}

extern "C" size_t uc2_adamw_chunk_bytes(void) { return sizeof(AdamChunk); }

// chunks: DEVICE pointer to n_chunks AdamChunk records; active/steps: DEVICE int32 [n_params]
extern "C" int uc2_adamw_step(const void* chunks, int n_chunks, int n_params, int n_groups, const float* lr,
                              const float* beta1, const float* beta2, const float* eps, const float* weight_decay,
                              const int* correct_bias, const int* active_dev, int* steps_dev,
                              const float* grad_scale_dev, int zero_grad, void* stream) {
  UC2_CHECK_ARG(n_groups >= 1 && n_groups <= UC2_ADAM_MAX_GROUPS);
  if (n_chunks <= 0) return 0;
  UC2_CHECK_ARG(chunks && lr && beta1 && beta2 && eps && weight_decay && correct_bias && steps_dev);
  AdamGroups gs;
  for (int i = 0; i < UC2_ADAM_MAX_GROUPS; ++i) {
    const int j = i < n_groups ? i : 0;
    gs.lr[i] = lr[j]; gs.beta1[i] = beta1[j]; gs.beta2[i] = beta2[j]; gs.eps[i] = eps[j];
    gs.wd[i] = weight_decay[j]; gs.correct_bias[i] = correct_bias[j];
  }
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunks, gs,
                     active_dev, (const int*)steps_dev, grad_scale_dev, zero_grad);
  UC2_LAUNCH_CHECK();
  hipLaunchKernelGGL(adamw_bump_kernel, dim3((n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_params,
                     active_dev, steps_dev);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---- sum of squares (global grad norm), deterministic: every workgroup writes ONE partial (fixed order inside
// the workgroup), the clip kernel adds the partials in a fixed tree.  No atomics: data-parallel replicas that
// hold identical gradients must compute bit-identical norms or they drift apart through the clip coefficient.
#define UC2_SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_kernel(size_t n, const float* __restrict__ x, float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// partials: UC2_SUMSQ_BLOCKS floats per call (the caller provides one such slot per gradient span)
extern "C" int uc2_sumsq_partials(size_t n, const float* x, float* partials, void* stream) {
  UC2_CHECK_ARG(partials);
  UC2_CHECK_ARG(n == 0 || (x && (((uintptr_t)x & 15) == 0)));
  hipLaunchKernelGGL(sumsq_kernel, dim3(UC2_SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, n, x, partials);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_sumsq_blocks(void) { return UC2_SUMSQ_BLOCKS; }

// coef = min(1, max_norm / (sqrt(sum) + 1e-6)); norm_out = sqrt(sum); sum over `count` partials, fixed order
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int count, float max_norm,
                                                        float* coef, float* norm_out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nrm = sqrtf((float)red[0]);
    if (norm_out) *norm_out = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    *coef = c < 1.f ? c : 1.f;
  }
}
extern "C" int uc2_clip_coef(const float* partials, int count, float max_norm, float* coef, float* norm_out,
                             void* stream) {
  UC2_CHECK_ARG(partials && coef && count > 0);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, count, max_norm, coef,
                     norm_out);
  UC2_LAUNCH_CHECK();
  return 0;
}

// x *= *scale_dev * scale_imm   (in place; gradient clipping, all-reduce averaging)
__global__ __launch_bounds__(256) void scale_kernel(size_t n, float* __restrict__ x, const float* __restrict__ sdev,
                                                    float simm) {
  const float s = (sdev ? *sdev : 1.f) * simm;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = *reinterpret_cast<float4*>(x + i * 4);
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    *reinterpret_cast<float4*>(x + i * 4) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] *= s;
}
extern "C" int uc2_scale(size_t n, float* x, const float* scale_dev, float scale_imm, void* stream) {
  if (n == 0) return 0;
  UC2_CHECK_ARG(x && (((uintptr_t)x & 15) == 0));
  size_t g = (n / 4 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, n, x, scale_dev, scale_imm);
  UC2_LAUNCH_CHECK();
  return 0;
}
