// Pipelined bf16 GEMM for the aligned shapes of the hot path (every encoder GEMM, forward and backward).
//
// Workgroup tile BM x 128 x BK (BM = 256 with 8 waves, or 128 with 4 waves), each wave a 64x64 block of
// v_mfma_f32_32x32x16_bf16 tiles, fp32 accumulate.  Staging is asynchronous LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging VGPRs) into an NSTAGE-deep LDS ring:
//   * tiles kt+1 .. kt+NSTAGE-2 are in flight while tile kt feeds the MFMAs; ONE raw s_barrier per k-tile,
//     counted `s_waitcnt vmcnt(N)` (never 0 in steady state) so the DMA stays in flight across barriers.
//     Bytes in flight per CU, not MFMA issue, bound these short-K GEMMs (K = 768): at ~2 us HBM latency a
//     CU needs ~100 KiB in flight to feed its matrix cores, hence the 144 KiB ring and the 256-row tile
//     (11.4 KiB of operand traffic per MFLOP instead of 15.2 for 128x128);
//   * an LDS-DMA write is lane-linear, so both bank-conflict swizzles are applied to the per-lane SOURCE
//     address and mirrored on the fragment reads:
//       k-contiguous operand  [rows][BK*2 B]   : 16-B chunk c of row r sits at c ^ ((r>>1)&7) (128-B rows) or c ^ ((r>>2)&3) (64-B rows)
//                              -> conflict-free ds_read_b128 fragments;
//       k-strided operand     [BK k-rows][W*2 B]: chunk c of k-row k sits at c ^ ((k&3)<<2)
//                              -> conflict-free ds_read_b64_tr_b16 (hardware-transposed) fragments.
// Requirements (checked on the host, else gemm.hip's generic kernel runs): contraction length % 64 == 0,
// 16-byte aligned operands with leading dimensions % 8 == 0, and extents % 8 == 0 for k-strided operands.
// Row/column edges are handled by clamping the source row (the clamped lanes only feed outputs that the
// epilogue guards away).
#include "gemm_common.h"
#include <stdlib.h>

#define GF_BN 128

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* glb_void_p;

// per-lane source pointer for wave-instruction `wi` (1 KiB of the operand tile) at the first k-tile.
//   !TR: tile [W rows][BK] , row = BK*2 bytes ;  TR: tile [BK k-rows][W], k-row = W*2 bytes
template <bool TR, int W, int BK>
__device__ __forceinline__ const bf16* gf_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int wi,
                                              int l) {
  if (!TR) {
    constexpr int CPR = BK / 8;                 // 16-B chunks per row (8 or 4)
    constexpr int RPI = 64 / CPR;               // rows per wave-instruction
    const int row = wi * RPI + l / CPR, cp = l % CPR;
    const int c = cp ^ ((row >> (CPR == 8 ? 1 : 2)) & (CPR - 1));
    const int gr = min(r0 + row, rows - 1);
    return X + (size_t)gr * ld + kbeg + c * 8;
  } else {
    constexpr int CPR = W / 8;                  // chunks per k-row (16 or 32)
    constexpr int RPI = 64 / CPR;               // k-rows per wave-instruction (4 or 2)
    const int krow = wi * RPI + l / CPR, cp = l % CPR;
    const int c = cp ^ ((krow & 3) << 2);
    const int col = min(r0 + c * 8, rows - 8);
    return X + (size_t)(kbeg + krow) * ld + col;
  }
}

// one MFMA operand fragment (32 rows x 16 k) for k16-step s of the tile; rbase = first row of the fragment
template <bool TR, int W, int BK>
__device__ __forceinline__ bf16x8 gf_frag(const char* lds, int rbase, int s, int lane) {
  if (!TR) {
    constexpr int CPR = BK / 8;
    const int row = rbase + (lane & 31), h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(lds + row * (BK * 2) +
                                            ((((s << 1) + h) ^ ((row >> (CPR == 8 ? 1 : 2)) & (CPR - 1))) << 4));
  } else {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
    const int krow = 16 * s + 8 * h + q;                 // krow & 3 == q for both reads
    const int col = rbase + 16 * (G & 1) + 4 * pp;
    const int off = krow * (W * 2) + ((((col >> 3) ^ (q << 2))) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) short4v* lds_p;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off + 4 * (W * 2)));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    return bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<9>() { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<18>() { asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }

template <bool TA, bool TB, bool TACC, int BM, int BK, int NSTAGE>
__global__ __launch_bounds__(BM * 2) void gemm_bf16_fast_kernel(GemmArgs p) {
  constexpr int NW = BM / 32;                         // waves: 8 (BM 256) or 4 (BM 128); wave grid (BM/64) x 2
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = GF_BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NW;           // wave-instructions per wave per k-tile
  constexpr int B_PW = B_BYTES / 1024 / NW;
  constexpr int LPT = A_PW + B_PW;
  static_assert(B_PW >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  const int nbx = (p.N + GF_BN - 1) / GF_BN, nby = (p.M + BM - 1) / BM;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {   // XCD-aware, bijective: consecutive tiles (sharing an A row panel) stay on one XCD's L2
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int m0 = (bid / nbx) * BM, n0 = (bid % nbx) * GF_BN;

  const int ktiles = p.K / BK;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int tbeg = blockIdx.z * per, tend = min(ktiles, tbeg + per);
  if (tbeg >= tend) return;
  const int nt = tend - tbeg, kbeg = tbeg * BK;

  const bf16* pa[A_PW];
  const bf16* pb[B_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) pa[i] = gf_src<TA, BM, BK>(A, p.lda, p.M, m0, kbeg, i * NW + w, lane);
#pragma unroll
  for (int i = 0; i < B_PW; ++i) pb[i] = gf_src<TB, GF_BN, BK>(B, p.ldb, p.N, n0, kbeg, i * NW + w, lane);
  const size_t stepa = TA ? (size_t)BK * p.lda : (size_t)BK;
  const size_t stepb = TB ? (size_t)BK * p.ldb : (size_t)BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define GF_ISSUE(STAGE_IDX)                                                                                   \
  do {                                                                                                        \
    char* sa_ = smem + (STAGE_IDX) * STAGE + w * 1024;                                                        \
    _Pragma("unroll") for (int i = 0; i < A_PW; ++i) {                                                        \
      __builtin_amdgcn_global_load_lds((glb_void_p)pa[i], (lds_void_p)(sa_ + i * NW * 1024), 16, 0, 0);       \
      pa[i] += stepa;                                                                                         \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < B_PW; ++i) {                                                        \
      __builtin_amdgcn_global_load_lds((glb_void_p)pb[i], (lds_void_p)(sa_ + A_BYTES + i * NW * 1024), 16, 0, 0); \
      pb[i] += stepb;                                                                                         \
    }                                                                                                         \
  } while (0)

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nt) GF_ISSUE(s);

  int cur = 0, nxt = NSTAGE - 1;                       // stage holding tile kt / stage to refill
  for (int kt = 0; kt < nt; ++kt) {
    // wait for tile kt only: the tiles issued after it (up to NSTAGE-2) stay in flight across the barrier
    const int ahead = min(NSTAGE - 2, nt - 1 - kt);
    if (NSTAGE >= 4 && ahead == 2) wait_vmcnt<2 * LPT>();
    else if (NSTAGE >= 3 && ahead >= 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                      // every wave's part of tile kt landed; stage nxt is free
    if (kt + NSTAGE - 1 < nt) GF_ISSUE(nxt);
    const char* As = smem + cur * STAGE;
    const char* Bs = As + A_BYTES;
    if (p.atomic & 0x100) {          // diagnostic: fetch-only (no fragment reads, no MFMAs): the staging rate alone
      cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
      nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
      continue;
    }
    // all fragment reads of the k-tile are issued up front; the MFMAs then wait with counted lgkmcnt, so
    // the LDS latency of step s+1.. hides under the MFMAs of step s even at 1-2 waves per SIMD
    bf16x8 a[BK / 16][2], b[BK / 16][2];
    if (p.atomic & 0x400) {          // diagnostic: no LDS reads (constant fragments), MFMAs only
#pragma unroll
      for (int s = 0; s < BK / 16; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[s][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; b[s][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; }
    } else {
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s][i] = gf_frag<TA, BM, BK>(As, wm * 64 + i * 32, s, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[s][j] = gf_frag<TB, GF_BN, BK>(Bs, wn * 64 + j * 32, s, lane);
      }
    }
    if (p.atomic & 0x200) {          // diagnostic: LDS reads but no MFMAs (fragments folded by cheap VALU)
#pragma unroll
      for (int s = 0; s < BK / 16; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0][s] += (float)a[s][i][0] + (float)b[s][i][7];
      cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
      nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
      continue;
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TACC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[s][j], a[s][i], acc[i][j], 0, 0, 0);
          else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
    }
    // pin the interleave (LLVM sched_group_barrier masks: MFMA 0x8, DS_READ 0x100): the first two k16-steps'
    // fragments, then one ds_read behind every MFMA while fragments remain, then the rest of the MFMAs
    {
      constexpr int NF = (BK / 16) * 4, NM = (BK / 16) * 4;
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int q = 0; q < NF - 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x8, NM - (NF - 8), 0);
    }
    cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
    nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
  }
#undef GF_ISSUE
  if ((p.atomic & 0x100) && !(p.atomic & 0x1000)) return;
  if (p.atomic & 0x800) {            // diagnostic: full main loop, no epilogue (one dummy store keeps acc alive)
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) z += acc[i][j][r];
    if (z == 123.456f) reinterpret_cast<bf16*>(p.C)[0] = (bf16)z;
    return;
  }
  __syncthreads();                     // all waves are done reading the last stage: LDS is reused below
  bf16_tile_epilogue<TACC>(p, acc, m0, n0, wm, wn, lane, smem + w * EPI_LDS_PER_WAVE);
}

// ------------------------------------------------------------------------------------------------------
// host side: variant selection
// ------------------------------------------------------------------------------------------------------
static int g_fetch_only = 0;
extern "C" int uc2_gemm_set_fetch_only(int v) { g_fetch_only = v; return 0; }
static int g_variant = -1;          // -1 = read UC2_GEMM_VARIANT; -2 = per-shape heuristic (default); 0..5 = fixed; 99 = generic kernel
extern "C" int uc2_gemm_set_variant(int v) { g_variant = v; return 0; }

template <bool TA, bool TB, bool TACC, int BM, int BK, int NSTAGE>
static void gf_launch1(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = NSTAGE * (BM * BK * 2 + GF_BN * BK * 2);
  auto kern = gemm_bf16_fast_kernel<TA, TB, TACC, BM, BK, NSTAGE>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nwg = ((p.N + GF_BN - 1) / GF_BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg, 1, p.split_k), dim3(BM * 2), smem, st, p);
}
template <bool TA, bool TB, bool TACC>
static void gf_launch2(const GemmArgs& p, int variant, hipStream_t st) {
  switch (variant) {
    case 0: gf_launch1<TA, TB, TACC, 128, 64, 2>(p, st); break;     // 64 KiB LDS, 2 workgroups / CU
    case 2: gf_launch1<TA, TB, TACC, 256, 32, 3>(p, st); break;     // 72 KiB LDS, 2 workgroups / CU
    case 3: gf_launch1<TA, TB, TACC, 256, 32, 4>(p, st); break;     // 96 KiB LDS, 1 workgroup / CU
    case 4: gf_launch1<TA, TB, TACC, 128, 64, 3>(p, st); break;     // 96 KiB LDS, 1 workgroup / CU
    case 5: gf_launch1<TA, TB, TACC, 128, 32, 4>(p, st); break;     // 64 KiB LDS, 2 workgroups / CU
    default: gf_launch1<TA, TB, TACC, 256, 64, 3>(p, st); break;    // 144 KiB LDS, 1 workgroup / CU
  }
}

// returns 1 if the shape qualifies and the kernel was launched, 0 if the caller must use the generic kernel
int uc2_gemm_bf16_fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st) {
  if (p.K < 64 || (p.K % 64) != 0) return 0;
  if (!p.a_vec || !p.b_vec) return 0;
  if (trans_a ? ((p.M & 7) != 0 || p.M < 8) : (p.M < 1)) return 0;
  if (trans_b ? ((p.N & 7) != 0 || p.N < 8) : (p.N < 1)) return 0;
  if (g_variant == -1) { const char* e = getenv("UC2_GEMM_VARIANT"); g_variant = e ? atoi(e) : -2; }
  int variant = g_variant;
  if (variant == 99) return 0;                       // caller asked for the generic kernel
  if (variant == -2) {
    // measured on MI355X (tests/bench_gemm.py, profiles/): forward X*W^T -> 256x128x32 ring of 3, 2 WG/CU;
    // input-gradient dY*W with a narrow output -> the generic register-staged kernel (3 WG/CU) wins;
    // weight-gradient (long contraction, split-K) -> 256x128x64 ring of 3.
    if (!trans_a && !trans_b) variant = 2;
    else if (!trans_a && trans_b) { if (p.N >= 2048) variant = 2; else return 0; }
    else variant = 1;
  }
  const bool tacc = !(p.c_f32 && p.atomic);
  GemmArgs pd = p;
  if (g_fetch_only) pd.atomic |= (g_fetch_only << 8);
#define GF_GO(TA_, TB_) do { if (tacc) gf_launch2<TA_, TB_, true>(pd, variant, st); else gf_launch2<TA_, TB_, false>(pd, variant, st); } while (0)
  if (!trans_a && !trans_b) GF_GO(false, false);
  else if (!trans_a && trans_b) GF_GO(false, true);
  else if (trans_a && !trans_b) GF_GO(true, false);
  else GF_GO(true, true);
#undef GF_GO
  return 1;
}
