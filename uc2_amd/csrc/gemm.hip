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
#include "gemm_common.h"
#include <stdlib.h>

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

  // (the K loop ends with a barrier: every wave is done reading the staging tiles)
  bf16_tile_epilogue<TACC>(p, acc, m0, n0, wm, wn, lane, smem + w * EPI_LDS_PER_WAVE);
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

int uc2_gemm_bf16_fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st);   // gemm_fast.hip
extern "C" int uc2_colsum_accum(int dtype, int M, int N, const void* X, int ldx, const uint8_t* rowmask, float* out, void* stream);   // elementwise.hip

// Split-K workspace (caller-owned device memory, passed with each call): with it, the ping-pong kernel's
// weight-gradient items store their fp32 partial tiles with plain stores and uc2_splitk_reduce adds them into C --
// no atomics, so the result is bit-reproducible, and 64 MB of partials cost ~25 us instead of ~50 us of fp32 atomics.

// qkv_d > 0: the partial rows are in the head-interleaved q|k|v order of the fused QKV projection (row h 3D + w D + d, D = qkv_d;
// ops.BertLayerFn writes its QKV activations / gradients that way for the attention kernels' sake) and go to row w M/3 + h D + d of C,
// the order of the parameter arena
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int M, int N, int ldc, int split, const float* __restrict__ ws,
                                                            float* __restrict__ C, int accumulate, int qkv_d) {
  const size_t mn4 = (size_t)M * N / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < mn4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(ws)[i];
    for (int z = 1; z < split; ++z) {
      const float4 b = reinterpret_cast<const float4*>(ws + (size_t)z * M * N)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const size_t e = i * 4;
    size_t m = e / N;
    const size_t n = e - m * N;
    if (qkv_d > 0) {
      const size_t hh = m / (3 * (size_t)qkv_d), r = m - hh * 3 * qkv_d, w = r / qkv_d, d = r - w * qkv_d;
      m = w * (size_t)(M / 3) + hh * qkv_d + d;
    }
    float4* c = reinterpret_cast<float4*>(C + m * ldc + n);
    if (accumulate) { const float4 o = *c; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
    *c = a;
  }
}
void uc2_splitk_reduce(const GemmArgs& p, hipStream_t st) {        // called by gemm_fast.hip after the partial launch
  const size_t mn4 = (size_t)p.M * p.N / 4;
  const int blocks = (int)((mn4 + 255) / 256 < 2048 ? (mn4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.M, p.N, p.ldc, p.split_k, p.partial,
                     reinterpret_cast<float*>(p.C), p.accumulate, 0);
}
// second stage on its own (after a uc2_gemm issued with UC2_GEMM_DEFER_REDUCE): C (=|+=) sum_z workspace[z]
extern "C" int uc2_gemm_splitk_reduce(int M, int N, void* C, int ldc, int split_k, int accumulate, const void* workspace,
                                      size_t workspace_bytes, void* stream) {
  UC2_CHECK_ARG(C && workspace && (size_t)split_k * M * N * sizeof(float) <= workspace_bytes);
  UC2_CHECK_ARG((N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
  GemmArgs p{};
  p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.split_k = split_k; p.accumulate = accumulate;
  p.partial = reinterpret_cast<float*>(const_cast<void*>(workspace));
  uc2_splitk_reduce(p, (hipStream_t)stream);
  UC2_LAUNCH_CHECK();
  return 0;
}

// the same with the rows of the partial tiles in the head-interleaved q|k|v order (see splitk_reduce_kernel); M = 3 nh qkv_head_dim
extern "C" int uc2_gemm_splitk_reduce_qkv(int M, int N, void* C, int ldc, int split_k, int accumulate, const void* workspace,
                                          size_t workspace_bytes, int qkv_head_dim, void* stream) {
  UC2_CHECK_ARG(C && workspace && (size_t)split_k * M * N * sizeof(float) <= workspace_bytes);
  UC2_CHECK_ARG((N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
  UC2_CHECK_ARG(qkv_head_dim > 0 && M % (3 * qkv_head_dim) == 0);
  const size_t mn4 = (size_t)M * N / 4;
  const int blocks = (int)((mn4 + 255) / 256 < 2048 ? (mn4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, N, ldc, split_k,
                     reinterpret_cast<const float*>(workspace), reinterpret_cast<float*>(C), accumulate, qkv_head_dim);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---- grouped weight gradients: ONE launch of the persistent ping-pong kernel over the split-K items of up to four
//      dW_i[n_out_i, n_in_i] += dY_i^T X_i that share the contraction (`rows` tokens), then one reduction launch.
//      At ~10 k tokens (the reference's 104-pair micro-batch) a layer's four weight gradients are 27 + 9 + 36 + 36 output
//      tiles: launched one by one, each runs a single round of 117-144 items on 256 CUs (50 % of the chip idle, 215 us
//      for the four); as one launch of 216 items, 136 us.
struct Uc2WgradItem {             // mirrors include/uc2_hip.h
  const void* dy; const void* x; void* dw;
  int lddy, ldx, lddw, n_out, n_in, split_k;
};
struct ReduceGroup { int n; size_t end4[UC2_GEMM_MAX_GROUP]; const float* ws[UC2_GEMM_MAX_GROUP]; float* C[UC2_GEMM_MAX_GROUP];
                     int M[UC2_GEMM_MAX_GROUP], N[UC2_GEMM_MAX_GROUP], ldc[UC2_GEMM_MAX_GROUP], split[UC2_GEMM_MAX_GROUP]; };
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroup r) {
  const size_t total = r.end4[r.n - 1];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int g = 0;
#pragma unroll
    for (int k = 1; k < UC2_GEMM_MAX_GROUP; ++k) if (k < r.n && i >= r.end4[k - 1]) g = k;
    const size_t li = i - (g ? r.end4[g - 1] : 0);
    const size_t mn = (size_t)r.M[g] * r.N[g];
    const float* ws = r.ws[g];
    float4 a = reinterpret_cast<const float4*>(ws)[li];
    for (int z = 1; z < r.split[g]; ++z) {
      const float4 b = reinterpret_cast<const float4*>(ws + (size_t)z * mn)[li];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const size_t e = li * 4, m = e / r.N[g], n = e - m * r.N[g];
    float4* c = reinterpret_cast<float4*>(r.C[g] + m * r.ldc[g] + n);
    const float4 o = *c;
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    *c = a;
  }
}
void uc2_gemm_pp_group_launch(const GemmArgs& p, hipStream_t st);       // gemm_pp.hip

extern "C" size_t uc2_gemm_wgrad_group_workspace(int n, const Uc2WgradItem* items) {
  size_t b = 0;
  for (int i = 0; i < n; ++i) b += (size_t)items[i].split_k * items[i].n_out * items[i].n_in * sizeof(float);
  return b;
}
// returns 0, an error code, or -2 when the shapes are not ones the grouped kernel takes (the caller then issues uc2_gemm per item)
extern "C" int uc2_gemm_wgrad_group(int dtype, int n, const Uc2WgradItem* items, int rows, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  UC2_CHECK_ARG(n >= 1 && items);
  if (dtype != 1 || n > UC2_GEMM_MAX_GROUP || rows < 128 || (rows & 63)) return -2;
  UC2_CHECK_ARG(workspace && uc2_gemm_wgrad_group_workspace(n, items) <= workspace_bytes && ((uintptr_t)workspace & 15) == 0);
  GemmArgs p{};
  ReduceGroup r{};
  p.epi = EPI_NONE; p.c_f32 = 1; p.accumulate = 1; p.split_k = 1; p.alpha = 1.0f; p.ngroup = n;
  const int ktiles = rows / 64;
  float* ws = reinterpret_cast<float*>(workspace);
  int item0 = 0;
  size_t end4 = 0;
  for (int i = 0; i < n; ++i) {
    const Uc2WgradItem& it = items[i];
    UC2_CHECK_ARG(it.dy && it.x && it.dw && it.split_k >= 1);
    const int per = ((ktiles + it.split_k - 1) / it.split_k + 1) & ~1;
    if ((it.n_out & 255) || (it.n_in & 255) || (it.lddy & 7) || (it.ldx & 7) || (it.lddw & 3) || ((uintptr_t)it.dy & 15) ||
        ((uintptr_t)it.x & 15) || ((uintptr_t)it.dw & 15) || per < 2 || (ktiles & 1) || (it.split_k - 1) * per >= ktiles ||
        ((ktiles - (it.split_k - 1) * per) & 1) ||
        (size_t)rows * it.lddy * 2 >= (1ull << 32) || (size_t)rows * it.ldx * 2 >= (1ull << 32))
      return -2;                                          // (32-bit operand offsets, even k-tile counts per item, whole 256 x 256 tiles)
    GemmProb& q = p.grp[i];
    q.A = it.dy; q.B = it.x; q.partial = ws; q.M = it.n_out; q.N = it.n_in; q.lda = it.lddy; q.ldb = it.ldx;
    q.nbx = it.n_in / 256; q.mt = it.n_out / 256; q.per = per; q.ktiles = ktiles; q.item0 = item0;
    { int cg = q.nbx; if (cg > 6) { cg = 1; for (int d = 6; d >= 2; --d) if (q.nbx % d == 0) { cg = d; break; } if (cg == 1) cg = 6; } q.col_group = cg; }
    item0 += q.nbx * q.mt * it.split_k;
    const size_t mn = (size_t)it.n_out * it.n_in;
    end4 += mn / 4;
    r.end4[i] = end4; r.ws[i] = ws; r.C[i] = reinterpret_cast<float*>(it.dw); r.M[i] = it.n_out; r.N[i] = it.n_in; r.ldc[i] = it.lddw;
    r.split[i] = it.split_k;
    ws += mn * it.split_k;
  }
  p.grp_items = item0;
  r.n = n;
  // (fields of the single-problem path the kernel still reads before its first item: keep them consistent with problem 0)
  p.A = p.grp[0].A; p.B = p.grp[0].B; p.M = p.grp[0].M; p.N = p.grp[0].N; p.K = rows; p.lda = p.grp[0].lda; p.ldb = p.grp[0].ldb;
  p.col_group = p.grp[0].col_group;
  hipStream_t st = (hipStream_t)stream;
  uc2_gemm_pp_group_launch(p, st);
  const int blocks = (int)((end4 + 255) / 256 < 2048 ? (end4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(blocks), dim3(256), 0, st, r);
  UC2_LAUNCH_CHECK();
  return 0;
}

// Calls that named a ping-pong kernel (variant 5 / 8 / 9 / 12) and were run by another one because the shape, an alignment or the
// 32-bit staging-offset limit (an operand of 4 GiB or more) did not qualify.  Diagnostics only: nothing reads it to select a kernel.
#include <atomic>
static std::atomic<long long> g_pp_fallbacks{0};
extern "C" long long uc2_gemm_fallback_count(int reset) {
  return reset ? g_pp_fallbacks.exchange(0, std::memory_order_relaxed) : g_pp_fallbacks.load(std::memory_order_relaxed);
}

static int gemm_impl(int dtype, int trans_a, int trans_b, int M, int N, int K,
                     const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32,
                     const float* bias, int epilogue, const void* aux_in, void* aux_out, int ldaux,
                     int accumulate, int split_k, int variant, void* workspace, size_t workspace_bytes, int flags,
                     void* queue, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
#if UC2_EXPERIMENTS
  UC2_CHECK_ARG(variant == -2 || variant == 99 || (variant >= 0 && variant <= 9) || variant == 12 || variant == 13 || variant == 14);
#else
  UC2_CHECK_ARG(variant == -2 || variant == 99 || (variant >= 0 && variant <= 9) || variant == 12);
#endif
  UC2_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
  UC2_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_TANH);
  UC2_CHECK_ARG(!((epilogue == EPI_DGELU || epilogue == EPI_ADD) && aux_in == nullptr));
  UC2_CHECK_ARG(split_k >= 1);
  UC2_CHECK_ARG(!(split_k > 1 && !(accumulate && (c_is_f32 || dtype == 0))));   // split-K needs f32 accumulate
  UC2_CHECK_ARG(!(split_k > 1 && (bias != nullptr || epilogue != EPI_NONE)));
  if (M == 0 || N == 0) return 0;
  if (K == 0 && accumulate) return 0;                 // empty contraction (a weight gradient over zero rows): C += 0
  UC2_CHECK_ARG(C && (K == 0 || (A && B)));
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux_in = aux_in; p.aux_out = aux_out;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux ? ldaux : ldc;
  p.epi = epilogue; p.c_f32 = (dtype == 0) ? 1 : c_is_f32; p.accumulate = accumulate;
  p.split_k = split_k; p.atomic = (split_k > 1) ? 1 : 0; p.partial = nullptr; p.alpha = 1.0f; p.alpha_dev = nullptr; p.alpha_dev2 = nullptr;
  p.variant = variant; p.ws = reinterpret_cast<float*>(workspace); p.ws_bytes = workspace ? workspace_bytes : 0;
  p.queue = reinterpret_cast<int*>(queue);
  {
    const int nbx = N / 256 > 0 ? N / 256 : 1;
    int cg = nbx;
    if (nbx > 6) { cg = 1; for (int d = 6; d >= 2; --d) if (nbx % d == 0) { cg = d; break; } if (cg == 1) cg = 6; }
    const int want = (flags >> 24) & 15;                 // diagnostic override: UC2_GEMM_COLGROUP(n)
    p.col_group = want ? (want > nbx ? nbx : want) : cg;
  }
  p.defer = flags & 1; p.aux_deriv = (flags >> 1) & 1; p.skew = (flags >> 4) & 15; p.diag = (flags >> 8) & 0xFFFF;
  p.spare_cus = 8 * ((flags >> 28) & 7);                /* UC2_GEMM_SPARE(n): the persistent kernels leave 8 n CUs without a workgroup */
  p.a_vec = (((uintptr_t)A & 15) == 0) && ((lda & 7) == 0);
  p.b_vec = (((uintptr_t)B & 15) == 0) && ((ldb & 7) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) { UC2_CHECK_ARG(accumulate); return 0; }
  UC2_CHECK_ARG(!(p.defer && (dtype == 0 || variant == 99)));    // the fp32 and generic kernels have no two-stage split-K path
  if (dtype == 0) {
    if (!trans_a && !trans_b) launch_f32<false, false>(p, st);
    else if (!trans_a && trans_b) launch_f32<false, true>(p, st);
    else if (trans_a && !trans_b) launch_f32<true, false>(p, st);
    else launch_f32<true, true>(p, st);
  } else {
    const bool want_colsum = (epilogue == EPI_DGELU && aux_out != nullptr);
    const int fast = variant == 99 ? 0 : uc2_gemm_bf16_fast_try(p, trans_a, trans_b, st);
    if (fast < 0 || (p.defer && fast != 2)) {
      uc2_set_error(__FILE__, __LINE__, "UC2_GEMM_DEFER_REDUCE: this call cannot leave split-K partial tiles in the workspace (shape / "
                    "variant / 4 GiB operand limit / workspace); nothing was launched");
      return -1;
    }
    // diagnostic counter (uc2_gemm_fallback_count): an explicit ping-pong plan (variants 5 / 8 / 9 / 12) that another kernel ran
    if ((variant == 5 || variant == 8 || variant == 9 || variant == 12 || variant == 13 || variant == 14) && fast != 2) g_pp_fallbacks.fetch_add(1, std::memory_order_relaxed);
    if (fast) {
      UC2_LAUNCH_CHECK();
      if (want_colsum && fast != 2) return uc2_colsum_accum(1, M, N, C, ldc, nullptr, reinterpret_cast<float*>(aux_out), stream);
      return 0;
    }
    if (!trans_a && !trans_b) launch_bf16<false, false>(p, st);
    else if (!trans_a && trans_b) launch_bf16<false, true>(p, st);
    else if (trans_a && !trans_b) launch_bf16<true, false>(p, st);
    else launch_bf16<true, true>(p, st);
  }
  UC2_LAUNCH_CHECK();
  if (epilogue == EPI_DGELU && aux_out != nullptr)      // kernels without the fused column sums: one more pass over C
    return uc2_colsum_accum(dtype, M, N, C, ldc, nullptr, reinterpret_cast<float*>(aux_out), stream);
  return 0;
}

// C[M,N] (bf16) = dropout_p(A[M,K] W[N,K]^T + bias) + residual: the sum the LayerNorm of a BertSelfOutput / BertOutput normalises
// (model/layer.py:78-81, :112-115 of the reference: dense -> dropout -> LayerNorm(hidden + input)), produced by the GEMM's epilogue so
// that the LayerNorm kernels read one tensor instead of two and generate no mask.  The mask is the one uc2_ln_fwd / uc2_ln_bwd_partial
// derive from (seed, row * N + column): a layer may mix the fused and the unfused form between forward and backward.
// Runs on the ping-pong kernel only (variant 12); returns -2 with nothing launched when the shape does not qualify (M % 256, N % 256,
// K % 128, 16-byte alignment, operands below 4 GiB) -- the caller then takes the unfused route.
bool uc2_gemm_pp16_supported(const GemmArgs& p, int trans_a, int trans_b);                       // gemm_pp16.hip
void uc2_gemm_pp16_launch(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st);
extern "C" int uc2_gemm_drop_residual(int M, int N, int K, const void* A, int lda, const void* W, int ldw, void* C, int ldc,
                                      const float* bias, const void* residual, int ldres, float p_drop, const uint64_t* seed_ptr,
                                      uint64_t seed_imm, int flags, void* queue, void* stream) {
  UC2_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
  UC2_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f);
  if (M == 0 || N == 0) return 0;
  UC2_CHECK_ARG(A && W && C && residual);
  const int ktiles = K / 64;
  const unsigned long long ea = 2ull * M * lda, eb = 2ull * N * ldw;
  const bool ok = K >= 128 && (K % 64) == 0 && !(ktiles & 1) && (M % 256) == 0 && (N % 256) == 0 &&
                  ((uintptr_t)A & 15) == 0 && (lda & 7) == 0 && ((uintptr_t)W & 15) == 0 && (ldw & 7) == 0 &&
                  ((uintptr_t)C & 15) == 0 && (ldc & 7) == 0 && ((uintptr_t)residual & 15) == 0 && (ldres & 7) == 0 &&
                  ((uintptr_t)bias & 15) == 0 && ea < (1ull << 32) && eb < (1ull << 32);
  if (!ok) return -2;
  GemmArgs p{};
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.aux_in = residual; p.aux_out = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc; p.ldaux = ldres;
  // (no dropout: the plain residual-add epilogue)
  p.epi = p_drop > 0.f ? EPI_DROPADD : EPI_ADD; p.c_f32 = 0; p.accumulate = 0; p.split_k = 1; p.atomic = 0; p.alpha = 1.0f;
  p.variant = 12; p.queue = reinterpret_cast<int*>(queue);
  {
    const int nbx = N / 256;
    int cg = nbx;
    if (nbx > 6) { cg = 1; for (int d = 6; d >= 2; --d) if (nbx % d == 0) { cg = d; break; } if (cg == 1) cg = 6; }
    p.col_group = cg;
  }
  p.skew = (flags >> 4) & 15; p.diag = (flags >> 8) & 0xFFFF; p.spare_cus = 8 * ((flags >> 28) & 7);
  if (p.diag) p.atomic |= (p.diag << 8);
  p.a_vec = 1; p.b_vec = 1;
  p.drop_thresh = drop_thresh(p_drop); p.drop_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  p.drop_seed_ptr = seed_ptr; p.drop_seed_imm = seed_imm;
  if (!uc2_gemm_pp16_supported(p, 0, 0)) return -2;
  uc2_gemm_pp16_launch(p, 0, 0, (hipStream_t)stream);
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K,
                        const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32,
                        const float* bias, int epilogue, const void* aux_in, void* aux_out, int ldaux,
                        int accumulate, int split_k, int variant, void* workspace, size_t workspace_bytes, int flags,
                        void* stream) {
  return gemm_impl(dtype, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, bias, epilogue, aux_in, aux_out, ldaux,
                   accumulate, split_k, variant, workspace, workspace_bytes, flags, nullptr, stream);
}
// The same with a caller-owned item queue for the persistent ping-pong kernel (9 ints, zeroed once; the kernel leaves them
// zeroed; one queue per stream that launches GEMMs concurrently): workgroups take their third and later work items from a
// per-XCD counter instead of a fixed stride, so a workgroup that starts late (CUs held by a communication kernel) simply
// takes fewer items.  Other kernels ignore it.
extern "C" int uc2_gemm_queued(int dtype, int trans_a, int trans_b, int M, int N, int K,
                               const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32,
                               const float* bias, int epilogue, const void* aux_in, void* aux_out, int ldaux,
                               int accumulate, int split_k, int variant, void* workspace, size_t workspace_bytes, int flags,
                               void* queue, void* stream) {
  return gemm_impl(dtype, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, c_is_f32, bias, epilogue, aux_in, aux_out, ldaux,
                   accumulate, split_k, variant, workspace, workspace_bytes, flags, queue, stream);
}
