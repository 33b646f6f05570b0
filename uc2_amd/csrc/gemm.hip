// GEMM family for the uc2 hot path (gfx950).
//
//   C[M,N] (=|+=) epi( sum_k A(m,k) * B(n,k) + bias[n] )
//   A(m,k) = A[m*lda + k]  (trans_a = 0, "k-contiguous")  or  A[k*lda + m]  (trans_a = 1)
//   B(n,k) = B[n*ldb + k]  (trans_b = 0, nn.Linear weight layout [out,in])  or  B[k*ldb + n]
//
// This one entry point covers every dense contraction on the path (reference
// model/layer.py:76-78,111,139,152; model/model.py:355,357; layer.py:257-265):
//   forward        Y  = X  * W^T          (ta=0, tb=0)
//   input grad     dX = dY * W            (ta=0, tb=1)   -- no transposed weight copy needed
//   weight grad    dW += dY^T * X         (ta=1, tb=1, accumulate, split-K over the token rows)
//
// Two implementations:
//   bf16: 128x128x64 workgroup tile, 4 waves (2x2), v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//         k-contiguous operands sit in LDS as [rows][64] with a 16-B-chunk XOR swizzle and are
//         read with ds_read_b128; k-strided operands sit as [64][128+32] and are read with
//         ds_read_b64_tr_b16 (hardware transpose), so all four layouts run on MFMA.
//         Global->LDS staging goes through registers, issued one k-tile ahead of the MFMAs.
//   f32 : 64x64x16 tile on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) -- the parity mode.
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

// ------------------------------------------------------------------------------------------
// fp32 kernel: 64x64 tile, BK=16, 4 waves (2x2), each wave 32x32 = 2x2 MFMA 16x16x4 tiles
// ------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
  const float* __restrict__ B = reinterpret_cast<const float*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kchunk = ((p.K + p.split_k - 1) / p.split_k + 15) / 16 * 16;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(p.K, kbeg + kchunk);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!TA) {
        const int m = (t >> 4) + 16 * r, k = t & 15;
        const int gm = m0 + m, gk = k0 + k;
        As[k][m] = (gm < p.M && gk < kend) ? A[(size_t)gm * p.lda + gk] : 0.f;
      } else {
        const int k = (t >> 6) + 4 * r, m = t & 63;
        const int gm = m0 + m, gk = k0 + k;
        As[k][m] = (gm < p.M && gk < kend) ? A[(size_t)gk * p.lda + gm] : 0.f;
      }
      if (!TB) {
        const int n = (t >> 4) + 16 * r, k = t & 15;
        const int gn = n0 + n, gk = k0 + k;
        Bs[k][n] = (gn < p.N && gk < kend) ? B[(size_t)gn * p.ldb + gk] : 0.f;
      } else {
        const int k = (t >> 6) + 4 * r, n = t & 63;
        const int gn = n0 + n, gk = k0 + k;
        Bs[k][n] = (gn < p.N && gk < kend) ? B[(size_t)gk * p.ldb + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kr][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kr][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D map (16x16): col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * (lane >> 4) + r;
        const int n = n0 + wn * 32 + j * 16 + (lane & 15);
        epi_store<float>(p, m, n, acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------------
// bf16 kernel
// ------------------------------------------------------------------------------------------
#define GB_BM 128
#define GB_BN 128
#define GB_BK 64
#define GB_TSTRIDE 320            // bytes per k-row of a k-strided tile: (128 + 32) * 2
#define GB_TILE_BYTES 20480       // max(128*128, 64*320)

__device__ __forceinline__ bf16x8 ld_chunk(const bf16* __restrict__ p, int valid, bool vec) {
  // 8 consecutive bf16 starting at p; only the first `valid` (0..8) may be touched
  if (vec && valid >= 8) return *reinterpret_cast<const bf16x8*>(p);
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (e < valid) ? p[e] : (bf16)0.f;
  return v;
}

// stage one 128(rows) x 64(k) operand tile global -> registers (4 chunks of 16 B per thread)
template <bool TR>
__device__ __forceinline__ void g2r(const bf16* __restrict__ X, int ld, int rows, int r0, int kend, int k0,
                                    bool vec, int t, bf16x8 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + 256 * i;
    if (!TR) {                       // X[row][k], chunk = 8 consecutive k
      const int row = c >> 3, kc = c & 7;
      const int gr = r0 + row, gk = k0 + kc * 8;
      const int valid = (gr < rows) ? max(0, min(8, kend - gk)) : 0;
      reg[i] = valid ? ld_chunk(X + (size_t)gr * ld + gk, valid, vec) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    } else {                         // X[k][row], chunk = 8 consecutive rows
      const int kr = c >> 4, rc = c & 15;
      const int gk = k0 + kr, gr = r0 + rc * 8;
      const int valid = (gk < kend) ? max(0, min(8, rows - gr)) : 0;
      reg[i] = valid ? ld_chunk(X + (size_t)gk * ld + gr, valid, vec) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
}

template <bool TR>
__device__ __forceinline__ void r2s(char* lds, int t, const bf16x8 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + 256 * i;
    int off;
    if (!TR) {
      const int row = c >> 3, kc = c & 7;
      off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);       // XOR swizzle: conflict-free ds_read_b128
    } else {
      const int kr = c >> 4, rc = c & 15;
      off = kr * GB_TSTRIDE + rc * 16;
    }
    *reinterpret_cast<bf16x8*>(lds + off) = reg[i];
  }
}

// one MFMA operand fragment (32 rows x 16 k) for k16-step s of the tile; rbase = first row
template <bool TR>
__device__ __forceinline__ bf16x8 frag(const char* lds, int rbase, int s, int lane) {
  if (!TR) {
    const int row = rbase + (lane & 31), h = lane >> 5;
    const int off = row * 128 + ((((s << 1) + h) ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds + off);
  } else {
    // ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of
    // (k-row q, 4 columns at 4p); lane i receives column i of the 4 k-rows.
    const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
    const int krow = 16 * s + 8 * h + q;
    const int col = rbase + 16 * (G & 1) + 4 * pp;
    const int off = krow * GB_TSTRIDE + col * 2;
    typedef __attribute__((address_space(3))) short4v* lds_p;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off + 4 * GB_TSTRIDE));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    return bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

// TACC = true : accumulate D^T (lane owns 4 consecutive n of one m -> 8-byte bf16 stores)
// TACC = false: accumulate D   (one register = two 128-B row segments -> full-rate f32 atomics)
template <bool TA, bool TB, bool TACC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + GB_TILE_BYTES;
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  // XCD-aware tile order: consecutive tiles of one A row-panel land on the same XCD's L2
  const int nbx = (p.N + GB_BN - 1) / GB_BN, nby = (p.M + GB_BM - 1) / GB_BM;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int m0 = (bid / nbx) * GB_BM, n0 = (bid % nbx) * GB_BN;

  const int kchunk = ((p.K + p.split_k - 1) / p.split_k + GB_BK - 1) / GB_BK * GB_BK;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(p.K, kbeg + kchunk);
  if (kbeg >= kend) return;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 ra[4], rb[4];
  g2r<TA>(A, p.lda, p.M, m0, kend, kbeg, p.a_vec, t, ra);
  g2r<TB>(B, p.ldb, p.N, n0, kend, kbeg, p.b_vec, t, rb);
  r2s<TA>(As, t, ra);
  r2s<TB>(Bs, t, rb);
  __syncthreads();

  for (int k0 = kbeg; k0 < kend; k0 += GB_BK) {
    const bool more = (k0 + GB_BK) < kend;
    if (more) {                                   // next tile's loads fly under this tile's MFMAs
      g2r<TA>(A, p.lda, p.M, m0, kend, k0 + GB_BK, p.a_vec, t, ra);
      g2r<TB>(B, p.ldb, p.N, n0, kend, k0 + GB_BK, p.b_vec, t, rb);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag<TA>(As, wm * 64 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = frag<TB>(Bs, wn * 64 + j * 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TACC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      r2s<TA>(As, t, ra);
      r2s<TB>(Bs, t, rb);
    }
    __syncthreads();
  }

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

// ------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------
template <bool TA, bool TB>
static void launch_f32(const GemmArgs& p, hipStream_t st) {
  dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.split_k);
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB>), grid, dim3(256), 0, st, p);
}
template <bool TA, bool TB>
static void launch_bf16(const GemmArgs& p, hipStream_t st) {
  const int nwg = ((p.N + GB_BN - 1) / GB_BN) * ((p.M + GB_BM - 1) / GB_BM);
  dim3 grid(nwg, 1, p.split_k);
  const bool tacc = !(p.c_f32 && p.atomic);
  if (tacc) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true>), grid, dim3(256), 2 * GB_TILE_BYTES, st, p);
  else      hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false>), grid, dim3(256), 2 * GB_TILE_BYTES, st, p);
}

extern "C" int uc2_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K,
                        const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32,
                        const float* bias, int epilogue, const void* aux_in, void* aux_out, int ldaux,
                        int accumulate, int split_k, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
  UC2_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_TANH);
  UC2_CHECK_ARG(!((epilogue == EPI_DGELU || epilogue == EPI_ADD) && aux_in == nullptr));
  UC2_CHECK_ARG(split_k >= 1);
  UC2_CHECK_ARG(!(split_k > 1 && !(accumulate && (c_is_f32 || dtype == 0))));   // split-K needs f32 accumulate
  UC2_CHECK_ARG(!(split_k > 1 && (bias != nullptr || epilogue != EPI_NONE)));
  if (M == 0 || N == 0) return 0;
  UC2_CHECK_ARG(A && B && C);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux ? ldaux : ldc;
  p.epi = epilogue; p.c_f32 = (dtype == 0) ? 1 : c_is_f32; p.accumulate = accumulate;
  p.split_k = split_k; p.atomic = (split_k > 1) ? 1 : 0;
  p.a_vec = (((uintptr_t)A & 15) == 0) && ((lda & 7) == 0);
  p.b_vec = (((uintptr_t)B & 15) == 0) && ((ldb & 7) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) { UC2_CHECK_ARG(accumulate); return 0; }
  if (dtype == 0) {
    if (!trans_a && !trans_b) launch_f32<false, false>(p, st);
    else if (!trans_a && trans_b) launch_f32<false, true>(p, st);
    else if (trans_a && !trans_b) launch_f32<true, false>(p, st);
    else launch_f32<true, true>(p, st);
  } else {
    // f32-atomic epilogue needs 4-byte aligned f32 C: always true; bf16 path: nothing else to check
    if (!trans_a && !trans_b) launch_bf16<false, false>(p, st);
    else if (!trans_a && trans_b) launch_bf16<false, true>(p, st);
    else if (trans_a && !trans_b) launch_bf16<true, false>(p, st);
    else launch_bf16<true, true>(p, st);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}
