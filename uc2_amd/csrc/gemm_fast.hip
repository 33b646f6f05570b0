// Pipelined bf16 GEMM for the aligned shapes of the hot path (every encoder GEMM, forward and backward).
//
// Same tile and MFMA as gemm.hip's generic kernel (128x128x64, 4 waves, 32x32x16), but the staging is
// asynchronous LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging VGPRs):
//   * two LDS stages of 32 KiB; tile t+1 is in flight while tile t feeds the MFMAs; ONE barrier per k-tile;
//   * an LDS-DMA write is lane-linear, so both bank-conflict swizzles are applied to the per-lane SOURCE
//     address and mirrored on the fragment reads:
//       k-contiguous operand  [128 rows][128 B]: 16-B chunk c of row r sits at chunk c ^ ((r>>1)&7)
//                              -> conflict-free ds_read_b128 fragments;
//       k-strided operand     [64 k-rows][256 B]: chunk c of k-row k sits at chunk c ^ ((k&3)<<2)
//                              -> conflict-free ds_read_b64_tr_b16 (hardware-transposed) fragments;
//   * 64 KiB LDS and <= 128 VGPRs per workgroup -> two workgroups per CU that overlap each other's waits.
// Requirements (checked on the host, else gemm.hip's generic kernel runs): contraction length % 64 == 0,
// 16-byte aligned operands with leading dimensions % 8 == 0, and extents % 8 == 0 for k-strided operands.
// Row/column edges are handled by clamping the source row (the clamped lanes only feed outputs that the
// epilogue guards away).
#include "gemm_common.h"

#define GF_BM 128
#define GF_BN 128
#define GF_BK 64
#define GF_OPERAND_BYTES 16384
#define GF_STAGE_BYTES 32768

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* glb_void_p;

// per-lane source pointer of wave-instruction i (0..3) of one operand tile, at the first k-tile
template <bool TR>
__device__ __forceinline__ const bf16* gf_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int i,
                                              int w, int l) {
  if (!TR) {
    const int row = (i * 4 + w) * 8 + (l >> 3), cp = l & 7;
    const int c = cp ^ ((row >> 1) & 7);
    const int gr = min(r0 + row, rows - 1);
    return X + (size_t)gr * ld + kbeg + c * 8;
  } else {
    const int krow = (i * 4 + w) * 4 + (l >> 4), cp = l & 15;
    const int c = cp ^ ((krow & 3) << 2);
    const int col = min(r0 + c * 8, rows - 8);
    return X + (size_t)(kbeg + krow) * ld + col;
  }
}

template <bool TR>
__device__ __forceinline__ bf16x8 gf_frag(const char* lds, int rbase, int s, int lane) {
  if (!TR) {
    const int row = rbase + (lane & 31), h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((((s << 1) + h) ^ ((row >> 1) & 7)) << 4));
  } else {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
    const int krow = 16 * s + 8 * h + q;                 // krow & 3 == q for both reads
    const int col = rbase + 16 * (G & 1) + 4 * pp;
    const int off = krow * 256 + ((((col >> 3) ^ (q << 2))) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) short4v* lds_p;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off + 4 * 256));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    return bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

template <bool TA, bool TB, bool TACC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_fast_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  const int nbx = (p.N + GF_BN - 1) / GF_BN, nby = (p.M + GF_BM - 1) / GF_BM;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {   // XCD-aware, bijective: consecutive tiles (sharing an A row panel) stay on one XCD's L2
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int m0 = (bid / nbx) * GF_BM, n0 = (bid % nbx) * GF_BN;

  const int ktiles = p.K / GF_BK;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int tbeg = blockIdx.z * per, tend = min(ktiles, tbeg + per);
  if (tbeg >= tend) return;
  const int nt = tend - tbeg, kbeg = tbeg * GF_BK;

  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pa[i] = gf_src<TA>(A, p.lda, p.M, m0, kbeg, i, w, lane);
    pb[i] = gf_src<TB>(B, p.ldb, p.N, n0, kbeg, i, w, lane);
  }
  const size_t stepa = TA ? (size_t)GF_BK * p.lda : (size_t)GF_BK;
  const size_t stepb = TB ? (size_t)GF_BK * p.ldb : (size_t)GF_BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define GF_ISSUE(STAGE)                                                                              \
  do {                                                                                               \
    char* sa_ = smem + (STAGE) * GF_STAGE_BYTES + w * 1024;                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
      __builtin_amdgcn_global_load_lds((glb_void_p)pa[i], (lds_void_p)(sa_ + i * 4096), 16, 0, 0);   \
      __builtin_amdgcn_global_load_lds((glb_void_p)pb[i], (lds_void_p)(sa_ + GF_OPERAND_BYTES + i * 4096), 16, 0, 0); \
      pa[i] += stepa;                                                                                \
      pb[i] += stepb;                                                                                \
    }                                                                                                \
  } while (0)

  GF_ISSUE(0);
  for (int kt = 0; kt < nt; ++kt) {
    __syncthreads();                 // tile kt has landed for every wave; stage (kt+1)&1 is free again
    if (kt + 1 < nt) GF_ISSUE((kt + 1) & 1);
    const char* As = smem + (kt & 1) * GF_STAGE_BYTES;
    const char* Bs = As + GF_OPERAND_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = gf_frag<TA>(As, wm * 64 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = gf_frag<TB>(Bs, wn * 64 + j * 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TACC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
  }
#undef GF_ISSUE
  bf16_tile_epilogue<TACC>(p, acc, m0, n0, wm, wn, lane);
}

template <bool TA, bool TB>
static void gf_launch(const GemmArgs& p, hipStream_t st) {
  const int nwg = ((p.N + GF_BN - 1) / GF_BN) * ((p.M + GF_BM - 1) / GF_BM);
  dim3 grid(nwg, 1, p.split_k);
  const bool tacc = !(p.c_f32 && p.atomic);
  if (tacc) hipLaunchKernelGGL((gemm_bf16_fast_kernel<TA, TB, true>), grid, dim3(256), 2 * GF_STAGE_BYTES, st, p);
  else      hipLaunchKernelGGL((gemm_bf16_fast_kernel<TA, TB, false>), grid, dim3(256), 2 * GF_STAGE_BYTES, st, p);
}

// returns 1 if the shape qualifies and the kernel was launched, 0 if the caller must use the generic kernel
int uc2_gemm_bf16_fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st) {
  if (p.K < GF_BK || (p.K % GF_BK) != 0) return 0;
  if (!p.a_vec || !p.b_vec) return 0;
  if (trans_a ? ((p.M & 7) != 0 || p.M < 8) : (p.M < 1)) return 0;
  if (trans_b ? ((p.N & 7) != 0 || p.N < 8) : (p.N < 1)) return 0;
  static bool attr_done = false;
  if (!attr_done) {
    const void* ks[8] = {(const void*)gemm_bf16_fast_kernel<false, false, true>, (const void*)gemm_bf16_fast_kernel<false, false, false>,
                         (const void*)gemm_bf16_fast_kernel<false, true, true>,  (const void*)gemm_bf16_fast_kernel<false, true, false>,
                         (const void*)gemm_bf16_fast_kernel<true, false, true>,  (const void*)gemm_bf16_fast_kernel<true, false, false>,
                         (const void*)gemm_bf16_fast_kernel<true, true, true>,   (const void*)gemm_bf16_fast_kernel<true, true, false>};
    for (int i = 0; i < 8; ++i) hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GF_STAGE_BYTES);
    attr_done = true;
  }
  if (!trans_a && !trans_b) gf_launch<false, false>(p, st);
  else if (!trans_a && trans_b) gf_launch<false, true>(p, st);
  else if (trans_a && !trans_b) gf_launch<true, false>(p, st);
  else gf_launch<true, true>(p, st);
  return 1;
}
