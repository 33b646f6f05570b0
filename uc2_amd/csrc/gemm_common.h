// Shared pieces of the GEMM kernels: argument block, per-element epilogue, bf16 tile epilogue.
#pragma once
#include "common.h"

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_ADD = 3, EPI_TANH = 4 };

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias; const void* aux_in; void* aux_out;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int epi, c_f32, accumulate, split_k, atomic;
  int a_vec, b_vec;      // 16-byte vector loads allowed (alignment checked on the host)
};

// ------------------------------------------------------------------------------------------
// epilogue on one element
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void epi_store(const GemmArgs& p, int m, int n, float v) {
  if (m >= p.M || n >= p.N) return;
  if (p.bias) v += p.bias[n];
  const size_t ia = (size_t)m * p.ldaux + n;
  if (p.epi == EPI_GELU) {
    if (p.aux_out) reinterpret_cast<T*>(p.aux_out)[ia] = from_f<T>(v);
    v = gelu_f(v);
  } else if (p.epi == EPI_DGELU) {
    v *= dgelu_f(to_f<T>(reinterpret_cast<const T*>(p.aux_in)[ia]));
  } else if (p.epi == EPI_ADD) {
    v += to_f<T>(reinterpret_cast<const T*>(p.aux_in)[ia]);
  } else if (p.epi == EPI_TANH) {
    v = tanhf(v);
  }
  const size_t ic = (size_t)m * p.ldc + n;
  if (p.c_f32) {
    float* c = reinterpret_cast<float*>(p.C);
    if (p.atomic) atomicAdd(c + ic, v);
    else if (p.accumulate) c[ic] += v;
    else c[ic] = v;
  } else {
    T* c = reinterpret_cast<T*>(p.C);
    if (p.accumulate) v += to_f<T>(c[ic]);
    c[ic] = from_f<T>(v);
  }
}


// write one workgroup tile (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32 tiles) through the epilogue.
// TACC = true : acc holds D^T (lane owns 4 consecutive n of one m -> 8-byte bf16 stores)
// TACC = false: acc holds D   (one register = two 128-B row segments -> full-rate f32 atomics)
template <bool TACC>
__device__ __forceinline__ void bf16_tile_epilogue(const GemmArgs& p, const f32x16 (&acc)[2][2], int m0, int n0,
                                                   int wm, int wn, int lane) {
  // 32x32 C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int h = lane >> 5, c31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int mb = m0 + wm * 64 + i * 32, nb = n0 + wn * 64 + j * 32;
      if (TACC) {
        const int m = mb + c31;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int n = nb + 8 * c + 4 * h;
          float v[4] = {acc[i][j][4 * c], acc[i][j][4 * c + 1], acc[i][j][4 * c + 2], acc[i][j][4 * c + 3]};
          const bool fast = (m < p.M) && (n + 3 < p.N) && !p.c_f32 && !p.accumulate && ((p.ldc & 3) == 0) &&
                            ((p.ldaux & 3) == 0);
          if (fast) {
            if (p.bias) {
              const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
              v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            const size_t ia = (size_t)m * p.ldaux + n;
            if (p.epi == EPI_GELU) {
              if (p.aux_out) Vec4<bf16>::store(reinterpret_cast<bf16*>(p.aux_out) + ia, v);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
            } else if (p.epi == EPI_DGELU) {
              float x[4];
              Vec4<bf16>::load(reinterpret_cast<const bf16*>(p.aux_in) + ia, x);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(x[e]);
            } else if (p.epi == EPI_ADD) {
              float x[4];
              Vec4<bf16>::load(reinterpret_cast<const bf16*>(p.aux_in) + ia, x);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += x[e];
            } else if (p.epi == EPI_TANH) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            Vec4<bf16>::store(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) epi_store<bf16>(p, m, n + e, v[e]);
          }
        }
      } else {
        const int n = nb + c31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
          epi_store<bf16>(p, m, n, acc[i][j][r]);
        }
      }
    }
}
