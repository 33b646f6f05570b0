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
#include "gemm_tile.h"

// (second launch-bound argument = waves per SIMD: rings of <= 80 KiB are meant to run two workgroups per CU)
// F8 = true: the same ring with OCP fp8 (e4m3) operands, both k-contiguous (TA = TB = false).  A k-tile is still
// 128 bytes per row (BK = 64 "bf16 columns" = 128 fp8 values), so staging, swizzles and the LDS image are byte for
// byte those of the bf16 kernel; a k-tile is two steps of v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales: the
// per-tensor scales are folded into GemmArgs::alpha) instead of four of v_mfma_f32_32x32x16_bf16 -- the same matrix-pipe
// cycles for twice the contraction length.  A lane's 32 operand bytes of a step are the two 16-byte chunks
// 4s + 2h, 4s + 2h + 1 of its row for BOTH operands (any assignment of k to operand bytes is valid as long as A and B
// use the same one; checked on the device against an fp64 reference).
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int BK>
__device__ __forceinline__ i32x8 gf_frag_f8(const char* lds, int rbase, int s, int lane) {
  const int row = rbase + (lane & 31), h = lane >> 5, sw = (row >> 1) & 7;
  const i32x4 lo = *reinterpret_cast<const i32x4*>(lds + row * (BK * 2) + (((4 * s + 2 * h) ^ sw) << 4));
  const i32x4 hi = *reinterpret_cast<const i32x4*>(lds + row * (BK * 2) + (((4 * s + 2 * h + 1) ^ sw) << 4));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <bool TA, bool TB, bool TACC, int BM, int BK, int NSTAGE, bool F8 = false>
__global__ __launch_bounds__(BM * 2)
void gemm_bf16_fast_kernel(GemmArgs p) {
  static_assert(!F8 || (!TA && !TB && TACC && BK == 64), "fp8: k-contiguous operands, bf16 output, 128-byte k-tiles");
  if (F8 && p.alpha_dev) p.alpha = 1.0f / (p.alpha_dev[0] * p.alpha_dev2[0]);
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

  // Issue wave-instruction q (1 KiB) of a tile into ring stage DST: q < A_PW -> A, else B.  ADV: advance the
  // source to the next k-tile afterwards; when the ring is only being kept full (no tile left) the last valid
  // tile is re-fetched instead (BACK = one k-tile), so every iteration issues the same number of loads.
#define GF_ISSUE1(DST, Q, ADVA, ADVB, BACKA, BACKB)                                                            \
  do {                                                                                                        \
    if ((Q) < A_PW) {                                                                                         \
      __builtin_amdgcn_global_load_lds((glb_void_p)(pa[(Q)] - (BACKA)), (lds_void_p)((DST) + (Q) * NW * 1024), \
                                       16, 0, 0);                                                             \
      pa[(Q)] += (ADVA);                                                                                      \
    } else {                                                                                                  \
      __builtin_amdgcn_global_load_lds((glb_void_p)(pb[(Q) - A_PW] - (BACKB)),                                \
                                       (lds_void_p)((DST) + A_BYTES + ((Q) - A_PW) * NW * 1024), 16, 0, 0);   \
      pb[(Q) - A_PW] += (ADVB);                                                                               \
    }                                                                                                         \
  } while (0)

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) {               // prologue: the ring always holds NSTAGE-1 tiles
    const bool real = s < nt;
    char* dst = smem + s * STAGE + w * 1024;
#pragma unroll
    for (int q = 0; q < LPT; ++q)
      GF_ISSUE1(dst, q, real ? stepa : 0, real ? stepb : 0, real ? 0 : stepa, real ? 0 : stepb);
  }

  constexpr int KS = BK / 16;
  int cur = 0, nxt = NSTAGE - 1;                       // stage holding tile kt / stage to refill
  for (int kt = 0; kt < nt; ++kt) {
    // Wait for tile kt only: the NSTAGE-2 tiles issued after it stay in flight across the barrier (the count is
    // a compile-time constant because the ring is kept full to the end; the loop body has no branches).
    wait_vmcnt<(NSTAGE - 2) * LPT>();
    __builtin_amdgcn_s_barrier();                      // every wave's part of tile kt landed; stage nxt is free
    const bool refill = (kt + NSTAGE - 1 < nt);
    const size_t adva = refill ? stepa : 0, advb = refill ? stepb : 0;
    const size_t backa = refill ? 0 : stepa, backb = refill ? 0 : stepb;
    const char* As = smem + cur * STAGE;
    const char* Bs = As + A_BYTES;
    char* dst = smem + nxt * STAGE + w * 1024;
    cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
    nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    if (p.atomic & 0x100) {          // diagnostic: fetch-only (no fragment reads, no MFMAs): the staging rate alone
#pragma unroll
      for (int q = 0; q < LPT; ++q) GF_ISSUE1(dst, q, adva, advb, backa, backb);
      continue;
    }
    // Software pipeline inside the k-tile: behind each MFMA of k16-step s the wave issues a fragment read of step
    // s+1, and its share of the next tile's LDS-DMA is spread over the steps.  Spreading the DMA issue matters: the
    // CU's address unit accepts one 1-KiB wave-instruction per ~30 cycles, and a wave that issues all its DMA up
    // front sits in the issue queue instead of feeding the MFMA pipe (measured: fetch alone 50 us + MFMA alone
    // 35 us ran as 97 us when issued back to back).
    if constexpr (F8) {
      i32x8 a8[2][2], b8[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a8[0][i] = gf_frag_f8<BK>(As, wm * 64 + i * 32, 0, lane);
        b8[0][i] = gf_frag_f8<BK>(Bs, wn * 64 + i * 32, 0, lane);
      }
#define GF8_STEP(S)                                                                                            \
      {                                                                                                        \
        constexpr int c = (S) & 1, n = c ^ 1;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[c][j], a8[c][i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
        if constexpr ((S) + 1 < 2) {                                                                           \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            a8[n][i] = gf_frag_f8<BK>(As, wm * 64 + i * 32, (S) + 1, lane);                                    \
            b8[n][i] = gf_frag_f8<BK>(Bs, wn * 64 + i * 32, (S) + 1, lane);                                    \
          }                                                                                                    \
        }                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < LPT; ++q)                                                        \
          if (q % 2 == (S)) GF_ISSUE1(dst, q, adva, advb, backa, backb);                                       \
      }
      GF8_STEP(0)
      GF8_STEP(1)
#undef GF8_STEP
      continue;
    }
    bf16x8 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[0][i] = gf_frag<TA, BM, BK>(As, wm * 64 + i * 32, 0, lane);
      b[0][i] = gf_frag<TB, GF_BN, BK>(Bs, wn * 64 + i * 32, 0, lane);
    }
    // one k16-step with literal index S (the sched_group_barrier arguments must be integer constants)
    constexpr int RD = (TA ? 4 : 2) + (TB ? 4 : 2);            // ds_read instructions per step (tr fragment = 2 reads)
    constexpr int R0 = RD / 4 + (RD % 4 > 0), R1 = RD / 4 + (RD % 4 > 1), R2 = RD / 4 + (RD % 4 > 2), R3 = RD / 4;
#define GF_STEP(S)                                                                                             \
    if constexpr ((S) < KS) {                                                                                  \
      constexpr int c = (S) & 1, n = c ^ 1;                                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
          if (TACC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[c][i], acc[i][j], 0, 0, 0); \
          else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][i], b[c][j], acc[i][j], 0, 0, 0); \
        }                                                                                                      \
      if constexpr ((S) + 1 < KS) {                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
          a[n][i] = gf_frag<TA, BM, BK>(As, wm * 64 + i * 32, (S) + 1, lane);                                  \
          b[n][i] = gf_frag<TB, GF_BN, BK>(Bs, wn * 64 + i * 32, (S) + 1, lane);                               \
        }                                                                                                      \
      }                                                                                                        \
      _Pragma("unroll") for (int q = 0; q < LPT; ++q)                                                          \
        if (q % KS == (S)) GF_ISSUE1(dst, q, adva, advb, backa, backb);                                        \
      /* pin: (MFMA, reads) x 4, then this step's DMA issues; LLVM masks: MFMA 0x8, DS_READ 0x100, VMEM 0x10 */ \
      if constexpr ((S) + 1 < KS) {                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R2, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                       \
        if constexpr (R3 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R3, 0);                              \
      } else {                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                       \
      }                                                                                                        \
      if constexpr ((LPT + KS - 1 - (S)) / KS > 0)                                                             \
        __builtin_amdgcn_sched_group_barrier(0x10, (LPT + KS - 1 - (S)) / KS, 0);                              \
    }
    __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);        // step-0 fragments first
    GF_STEP(0)
    GF_STEP(1)
    GF_STEP(2)
    GF_STEP(3)
#undef GF_STEP
  }
#undef GF_ISSUE1
  if ((p.atomic & 0x100) && !(p.atomic & 0x1000)) return;      // diagnostic: staging only
  __syncthreads();                     // all waves are done reading the last stage: LDS is reused below
  bf16_tile_epilogue<TACC>(p, acc, m0, n0, wm, wn, lane, smem + w * EPI_LDS_PER_WAVE);
}

// ------------------------------------------------------------------------------------------------------
// Wave-specialised variant: 8 MFMA waves (4x2, 64x64 each, tile 256x128) + 4 loader waves per workgroup.
// The loader waves own ALL the LDS-DMA of the workgroup (one per SIMD, 12 (BK=64) wave-instructions per k-tile
// each), so the MFMA waves never queue behind the CU's address unit: their instruction stream is only
// ds_read + MFMA.  Same ring, same single barrier per k-tile (all 12 waves take it): a loader waits for its
// share of tile kt (counted vmcnt), takes the barrier, then refills the stage the MFMA waves have just left.
// ------------------------------------------------------------------------------------------------------
template <bool TA, bool TB, bool TACC, int BK, int NSTAGE, int NWL>
__global__ __launch_bounds__(512 + NWL * 64) void gemm_bf16_ws_kernel(GemmArgs p) {
  constexpr int BM = 256, NWC = 8;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = GF_BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PL = A_BYTES / 1024 / NWL, B_PL = B_BYTES / 1024 / NWL, LPT = A_PL + B_PL;   // per loader wave
  constexpr int KS = BK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;

  const int nbx = (p.N + GF_BN - 1) / GF_BN, nby = (p.M + BM - 1) / BM;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int m0 = (bid / nbx) * BM, n0 = (bid % nbx) * GF_BN;
  const int ktiles = p.K / BK;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int tbeg = blockIdx.z * per, tend = min(ktiles, tbeg + per);
  if (tbeg >= tend) return;
  const int nt = tend - tbeg, kbeg = tbeg * BK;

  if (w >= NWC) {
    // ------------------------------- loader wave -------------------------------
    const int lw = w - NWC;
    const bf16* pa[A_PL];
    const bf16* pb[B_PL];
#pragma unroll
    for (int i = 0; i < A_PL; ++i) pa[i] = gf_src<TA, BM, BK>(A, p.lda, p.M, m0, kbeg, i * NWL + lw, lane);
#pragma unroll
    for (int i = 0; i < B_PL; ++i) pb[i] = gf_src<TB, GF_BN, BK>(B, p.ldb, p.N, n0, kbeg, i * NWL + lw, lane);
    const size_t stepa = TA ? (size_t)BK * p.lda : (size_t)BK;
    const size_t stepb = TB ? (size_t)BK * p.ldb : (size_t)BK;
#define WS_ISSUE(DST, ADVA, ADVB, BACKA, BACKB)                                                                 \
    do {                                                                                                       \
      _Pragma("unroll") for (int q = 0; q < A_PL; ++q) {                                                       \
        __builtin_amdgcn_global_load_lds((glb_void_p)(pa[q] - (BACKA)), (lds_void_p)((DST) + q * NWL * 1024), 16, 0, 0); \
        pa[q] += (ADVA);                                                                                       \
      }                                                                                                        \
      _Pragma("unroll") for (int q = 0; q < B_PL; ++q) {                                                       \
        __builtin_amdgcn_global_load_lds((glb_void_p)(pb[q] - (BACKB)), (lds_void_p)((DST) + A_BYTES + q * NWL * 1024), 16, 0, 0); \
        pb[q] += (ADVB);                                                                                       \
      }                                                                                                        \
    } while (0)
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
      const bool real = s < nt;
      char* dst = smem + s * STAGE + lw * 1024;
      WS_ISSUE(dst, real ? stepa : 0, real ? stepb : 0, real ? 0 : stepa, real ? 0 : stepb);
    }
    int nxt = NSTAGE - 1;
    const bool no_fetch = (p.atomic & 0x2000) != 0;    // diagnostic: compute side only (stale LDS contents)
    for (int kt = 0; kt < nt; ++kt) {
      wait_vmcnt<(NSTAGE - 2) * LPT>();                // this wave's share of tile kt has landed
      __builtin_amdgcn_s_barrier();                    // ... publish it; the MFMA waves have left stage nxt
      if (no_fetch) continue;
      const bool refill = (kt + NSTAGE - 1 < nt);
      char* dst = smem + nxt * STAGE + lw * 1024;
      WS_ISSUE(dst, refill ? stepa : 0, refill ? stepb : 0, refill ? 0 : stepa, refill ? 0 : stepb);
      nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
    }
#undef WS_ISSUE
    __syncthreads();                                   // (drains the ring) matches the MFMA waves' pre-epilogue barrier
    return;
  }

  // --------------------------------- MFMA wave ---------------------------------
  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int cur = 0;
  for (int kt = 0; kt < nt; ++kt) {
    __builtin_amdgcn_s_barrier();                      // tile kt is in LDS (every loader waited for its share first)
    const char* As = smem + cur * STAGE;
    const char* Bs = As + A_BYTES;
    cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
    bf16x8 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[0][i] = gf_frag<TA, BM, BK>(As, wm * 64 + i * 32, 0, lane);
      b[0][i] = gf_frag<TB, GF_BN, BK>(Bs, wn * 64 + i * 32, 0, lane);
    }
    constexpr int RD = (TA ? 4 : 2) + (TB ? 4 : 2);
    constexpr int R0 = RD / 4 + (RD % 4 > 0), R1 = RD / 4 + (RD % 4 > 1), R2 = RD / 4 + (RD % 4 > 2), R3 = RD / 4;
#define WS_STEP(S)                                                                                             \
    if constexpr ((S) < KS) {                                                                                  \
      constexpr int c = (S) & 1, n = c ^ 1;                                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
          if (TACC) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][j], a[c][i], acc[i][j], 0, 0, 0); \
          else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][i], b[c][j], acc[i][j], 0, 0, 0); \
        }                                                                                                      \
      if constexpr ((S) + 1 < KS) {                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
          a[n][i] = gf_frag<TA, BM, BK>(As, wm * 64 + i * 32, (S) + 1, lane);                                  \
          b[n][i] = gf_frag<TB, GF_BN, BK>(Bs, wn * 64 + i * 32, (S) + 1, lane);                               \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, R2, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                       \
        if constexpr (R3 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R3, 0);                              \
      } else {                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                       \
      }                                                                                                        \
    }
    __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
    WS_STEP(0)
    WS_STEP(1)
    WS_STEP(2)
    WS_STEP(3)
#undef WS_STEP
  }
  if ((p.atomic & 0x100) && !(p.atomic & 0x1000)) { __syncthreads(); return; }
  __syncthreads();
  bf16_tile_epilogue<TACC>(p, acc, m0, n0, wm, wn, lane, smem + w * EPI_LDS_PER_WAVE);
}

template <bool TA, bool TB, bool TACC, int BK, int NSTAGE, int NWL>
static void ws_launch1(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = NSTAGE * (256 * BK * 2 + GF_BN * BK * 2);
  auto kern = gemm_bf16_ws_kernel<TA, TB, TACC, BK, NSTAGE, NWL>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nwg = ((p.N + GF_BN - 1) / GF_BN) * ((p.M + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(nwg, 1, p.split_k), dim3(512 + NWL * 64), smem, st, p);
}

bool uc2_gemm_pp_supported(int trans_a, int trans_b, int c_f32, int epi, int tile_rows);        // gemm_pp.hip
void uc2_gemm_pp_launch(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st, int tile_rows);
bool uc2_gemm_pp16_supported(const GemmArgs& p, int trans_a, int trans_b);                       // gemm_pp16.hip
void uc2_gemm_pp16_launch(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st);
#if UC2_EXPERIMENTS        // `make EXPERIMENTS=1` only: variants 13 / 14 (experiments/csrc/gemm_p1.hip, gemm_p2.hip), measured slower than 12
bool uc2_gemm_p1_supported(const GemmArgs& p, int trans_a, int trans_b);
void uc2_gemm_p1_launch(const GemmArgs& p, hipStream_t st);
bool uc2_gemm_p2_supported(const GemmArgs& p, int trans_a, int trans_b);
void uc2_gemm_p2_launch(const GemmArgs& p, hipStream_t st);
#else
static inline bool uc2_gemm_p1_supported(const GemmArgs&, int, int) { return false; }
static inline void uc2_gemm_p1_launch(const GemmArgs&, hipStream_t) {}
static inline bool uc2_gemm_p2_supported(const GemmArgs&, int, int) { return false; }
static inline void uc2_gemm_p2_launch(const GemmArgs&, hipStream_t) {}
#endif
void uc2_splitk_reduce(const GemmArgs& p, hipStream_t st);                                        // gemm.hip

// ------------------------------------------------------------------------------------------------------
// host side: variant selection
// ------------------------------------------------------------------------------------------------------

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
    case 6: ws_launch1<TA, TB, TACC, 64, 3, 4>(p, st); break;        // wave-specialised: 8 MFMA + 4 loader waves
    case 7: ws_launch1<TA, TB, TACC, 64, 3, 8>(p, st); break;        // wave-specialised: 8 MFMA + 8 loader waves
    default: gf_launch1<TA, TB, TACC, 256, 64, 3>(p, st); break;    // 144 KiB LDS, 1 workgroup / CU
  }
}

// fp8 (e4m3) x fp8 -> bf16, both operands k-contiguous: the operands are handed to the ring kernel as "bf16" matrices of
// half the width (same bytes).  M, N arbitrary (row clamping + guarded epilogue), K % 128 == 0.
bool uc2_gemm_pp8_supported(const GemmArgs& p);                                                     // gemm_pp8.hip
void uc2_gemm_pp8_launch(const GemmArgs& p, hipStream_t st);
// Diagnostics only (like uc2_gemm_fallback_count): e4m3 GEMM calls that the ring kernel ran because the ping-pong kernel does not
// take the shape (M not a multiple of 256, ...).  Tests use it to prove that a model-level fp8 run exercised gemm_pp8.hip.
#include <atomic>
static std::atomic<long long> g_fp8_ring{0}, g_fp8_pp{0};
extern "C" long long uc2_gemm_fp8_route_count(int which, int reset) {
  std::atomic<long long>& c = which ? g_fp8_pp : g_fp8_ring;
  return reset ? c.exchange(0, std::memory_order_relaxed) : c.load(std::memory_order_relaxed);
}
int uc2_gemm_fp8_launch(const GemmArgs& p8, hipStream_t st) {
  GemmArgs p = p8;
  p.K = p8.K / 2; p.lda = p8.lda / 2; p.ldb = p8.ldb / 2;
  // whole 256 x 256 tiles and an even number of k-tiles: the persistent ping-pong schedule on v_mfma_scale_f32_16x16x128_f8f6f4
  // (p.variant == 1 forces the ring kernel below: tests, A/B)
  if (p8.variant != 1 && uc2_gemm_pp8_supported(p)) { g_fp8_pp.fetch_add(1, std::memory_order_relaxed); uc2_gemm_pp8_launch(p, st); return 2; }
  g_fp8_ring.fetch_add(1, std::memory_order_relaxed);
  constexpr int BM = 256, BK = 64, NSTAGE = 3;
  constexpr int smem = NSTAGE * (BM * BK * 2 + GF_BN * BK * 2);
  auto kern = gemm_bf16_fast_kernel<false, false, true, BM, BK, NSTAGE, true>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nwg = ((p.N + GF_BN - 1) / GF_BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg, 1, 1), dim3(BM * 2), smem, st, p);
  return 0;
}

// returns 1 (ring kernel) / 2 (ping-pong family) if the shape qualifies and the kernel was launched, 0 if the caller must use the
// generic kernel, -1 (nothing launched) if UC2_GEMM_DEFER_REDUCE was asked for and the two-stage split-K path cannot take the call
static int fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st);
int uc2_gemm_bf16_fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st) {
  if (p.defer) {
    // The caller will run the reduction pass itself (uc2_gemm_splitk_reduce / _qkv) on the partial tiles in its workspace: that is
    // only meaningful when THIS call leaves them there.  Anything else -- a shape the ping-pong kernels do not take, the 32-bit
    // staging-offset limit, a missing or short workspace -- used to fall through to a kernel that accumulates into C directly, and
    // the caller's reduction then added stale workspace contents on top.  Refuse instead of computing something else.
    const int v = p.variant;
    const bool pp = v == 8 || v == 12 || v == 13 || v == 14;
    const int ktiles = p.K / 64, per = ((ktiles + p.split_k - 1) / p.split_k + 1) & ~1;
    const unsigned long long ea = 2ull * (trans_a ? p.K : p.M) * p.lda, eb = 2ull * (trans_b ? p.K : p.N) * p.ldb;
    const size_t need = (size_t)p.split_k * p.M * p.N * sizeof(float);
    const bool ok = pp && p.K >= 64 && (p.K % 64) == 0 && p.a_vec && p.b_vec && !(ktiles & 1) && ktiles - (p.split_k - 1) * per >= 2 &&
                    (p.M % 256) == 0 && (p.N & 255) == 0 && ea < (1ull << 32) && eb < (1ull << 32) &&
                    (trans_a ? (p.M & 7) == 0 : true) && (trans_b ? (p.N & 7) == 0 : true) &&
                    p.c_f32 && p.split_k > 1 && p.ws && need <= p.ws_bytes && (p.N & 3) == 0 && (p.ldc & 3) == 0 &&
                    ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.ws & 15) == 0 &&
                    (uc2_gemm_pp_supported(trans_a, trans_b, p.c_f32, p.epi, 256) || (v != 8 && uc2_gemm_pp16_supported(p, trans_a, trans_b)));
    if (!ok) return -1;
  }
  return fast_try(p, trans_a, trans_b, st);
}
static int fast_try(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st) {
  if (p.K < 64 || (p.K % 64) != 0) return 0;
  if (!p.a_vec || !p.b_vec) return 0;
  if (trans_a ? ((p.M & 7) != 0 || p.M < 8) : (p.M < 1)) return 0;
  if (trans_b ? ((p.N & 7) != 0 || p.N < 8) : (p.N < 1)) return 0;
  int variant = p.variant;                           // per call (uc2_gemm's `variant` argument), never process state
  if (variant == 99) return 0;                       // caller asked for the generic kernel
  // (variants 10 = rolling epilogue and 11 = two phases per k-tile were measured in round 3, never selected by a plan, and
  //  live under experiments/csrc/ with their result table in profiles/HISTORY.md; uc2_gemm rejects the numbers)
  const bool want_pp16 = variant == 12;              // ping-pong kernel on the 16x16x32 MFMA (gemm_pp16.hip)
  const bool want_p1 = variant == 13;                // experiments build only (uc2_gemm rejects 13 / 14 otherwise): one wave per SIMD, falls back to 12
  const bool want_p2 = variant == 14;                // ... with the epilogue in the next item's MFMA gaps: falls back to 12
  if (want_pp16 || want_p1 || want_p2) variant = 8;
  if (variant == 8 || variant == 9 || variant == 5) {
    // ping-pong kernel: whole 256x256 (variant 9: 192x256, variant 5: 128x256) tiles, >= 2 k-tiles per split, and (bf16 output) an
    // epilogue made of whole 16-byte accesses; anything else runs on the ring kernels above / the generic kernel
    const int rows = variant == 9 ? 192 : (variant == 5 ? 128 : 256);
    const int ktiles = p.K / 64, per = ((ktiles + p.split_k - 1) / p.split_k + 1) & ~1;   // k-tiles per split: even, >= 2
    if ((ktiles & 1) || ktiles - (p.split_k - 1) * per < 2) return 0;
    if ((p.M % rows) || (p.N & 255)) return 0;
    {   // staging sources are 32-bit byte offsets from the operand bases
      const unsigned long long ea = 2ull * (trans_a ? p.K : p.M) * p.lda, eb = 2ull * (trans_b ? p.K : p.N) * p.ldb;
      if (ea >= (1ull << 32) || eb >= (1ull << 32)) return 0;
    }
    if (!uc2_gemm_pp_supported(trans_a, trans_b, p.c_f32, p.epi, rows) &&
        !((want_pp16 || want_p1 || want_p2) && !p.c_f32 && uc2_gemm_pp16_supported(p, trans_a, trans_b))) return 0;
    if (!p.c_f32 && (p.accumulate || (p.ldc & 7) || (p.ldaux & 7) || ((uintptr_t)p.C & 15) ||
                     ((uintptr_t)p.aux_in & 15) || ((uintptr_t)p.aux_out & 15) || ((uintptr_t)p.bias & 15)))
      return 0;
  }
  const bool tacc = !(p.c_f32 && p.atomic);
  GemmArgs pd = p;
  if (p.diag) pd.atomic |= (p.diag << 8);
  // skew is off unless the call asks for it: back-to-back launches of the double-store GELU GEMM gained 16 % from
  // de-phasing, inside the training step no kernel moved
  if (variant == 9) { uc2_gemm_pp_launch(pd, trans_a, trans_b, st, 192); return 2; }
  if (variant == 5) { uc2_gemm_pp_launch(pd, trans_a, trans_b, st, 128); return 2; }
  if (variant == 8) {
    const size_t need = (size_t)p.split_k * p.M * p.N * sizeof(float);
    if (p.c_f32 && p.split_k > 1 && p.ws && need <= p.ws_bytes && (p.N & 3) == 0 && (p.ldc & 3) == 0 &&
        ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.ws & 15) == 0) {
      pd.partial = p.ws;                             // two-stage: plain partial stores, then one reduction pass
      if ((want_pp16 || want_p1 || want_p2) && uc2_gemm_pp16_supported(pd, trans_a, trans_b)) uc2_gemm_pp16_launch(pd, trans_a, trans_b, st);
      else uc2_gemm_pp_launch(pd, trans_a, trans_b, st, 256);
      if (!p.defer) uc2_splitk_reduce(pd, st);
    } else {
      if (want_p1 && uc2_gemm_p1_supported(pd, trans_a, trans_b)) uc2_gemm_p1_launch(pd, st);
      else if (want_p2 && uc2_gemm_p2_supported(pd, trans_a, trans_b)) uc2_gemm_p2_launch(pd, st);
      else if ((want_pp16 || want_p1 || want_p2) && uc2_gemm_pp16_supported(pd, trans_a, trans_b)) uc2_gemm_pp16_launch(pd, trans_a, trans_b, st);
      else uc2_gemm_pp_launch(pd, trans_a, trans_b, st, 256);
    }
    return 2;                                         // (2: the kernel also produced the EPI_DGELU column sums)
  }
#define GF_GO(TA_, TB_) do { if (tacc) gf_launch2<TA_, TB_, true>(pd, variant, st); else gf_launch2<TA_, TB_, false>(pd, variant, st); } while (0)
  if (!trans_a && !trans_b) GF_GO(false, false);
  else if (!trans_a && trans_b) GF_GO(false, true);
  else if (trans_a && !trans_b) GF_GO(true, false);
  else GF_GO(true, true);
#undef GF_GO
  return 1;
}
