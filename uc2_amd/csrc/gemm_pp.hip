#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// Ping-pong variant: tile 256x256x64, 8 waves = 2 (M) x 4 (N), 128x64 of output per wave (128 accumulator
// registers), one workgroup per CU, 128 KiB of LDS = 2 k-tile buffers x 4 staging units of 16 KiB.
//
// A k-tile is consumed in four phases, one 64x32 quadrant of the wave's output each (8 MFMAs):
//     phase 0: read A0 (rows 0-63 of the wave's 128) and B0 (columns 0-31 of its 64)   -> A0 x B0
//     phase 1: read B1                                                                 -> A0 x B1
//     phase 2: read A1 (overwrites A0)                                                 -> A1 x B1
//     phase 3: (no reads)                                                              -> A1 x B0
// Every phase is  [L: fragment reads + 2 LDS-DMA issues + counted vmcnt] barrier [C: 8 MFMAs] barrier.  The two
// wave rows run one barrier apart (wave row 1 takes one extra barrier at the start), so in every barrier
// interval one wave of each SIMD is in its C section while the other is in its L section: the matrix pipe
// always has a wave feeding it and the other wave's LDS reads / DMA issues cost it nothing.
//
// Staging: the k-tile is cut into four units of 128 rows x 64 k in the order the phases consume them
// (A0-rows of both wave rows, B0-columns of all four wave columns, B1, A1).  Phase f issues unit f+6 (each wave
// two 1-KiB LDS-DMA instructions), so a unit is issued 5-6 phases before its first read and at least two
// barrier intervals after the last read of the unit it overwrites; `s_waitcnt vmcnt(8)` at the end of every
// L section retires exactly the units the next phase reads (all but the 4 youngest units), and the barrier
// that follows publishes them to the other waves.
// ------------------------------------------------------------------------------------------------------
#define PP_UNIT 16384

template <int J> __device__ __forceinline__ int pp_map(int ur) {     // unit row -> row/column of the 256-wide tile
  if (J == 0 || J == 3) return (ur >> 6) * 128 + (J == 3 ? 64 : 0) + (ur & 63);
  else return (ur >> 5) * 64 + (J == 2 ? 32 : 0) + (ur & 31);
}

// per-lane source of wave-instruction wi (0..15) of a unit; same LDS images and swizzles as gf_src<TR,128,64>
template <bool TR, int J>
__device__ __forceinline__ const bf16* pp_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int wi,
                                              int l) {
  if (!TR) {
    const int row = wi * 8 + (l >> 3), cp = l & 7;
    const int c = cp ^ ((row >> 1) & 7);
    const int gr = min(r0 + pp_map<J>(row), rows - 1);
    return X + (size_t)gr * ld + kbeg + c * 8;
  } else {
    const int krow = wi * 4 + (l >> 4), cp = l & 15;
    const int c = cp ^ ((krow & 3) << 2);
    const int col = min(r0 + pp_map<J>(c * 8), rows - 8);
    return X + (size_t)(kbeg + krow) * ld + col;
  }
}

// byte offset (inside a unit) of lane's part of the fragment at rows rbase.., k16-step 0; the step-s fragment
// is at  off ^ (s << 5)  (k-contiguous image) or  off + s * 4096  (k-strided image, second read + 1024).
template <bool TR> __device__ __forceinline__ unsigned pp_frag_off(int rbase, int lane) {
  if (!TR) {
    const int row = rbase + (lane & 31), h = lane >> 5;
    return row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
  } else {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
    const int krow = 8 * h + q;
    const int col = rbase + 16 * (G & 1) + 4 * pp;
    return krow * 256 + ((((col >> 3) ^ (q << 2))) << 4) + (col & 7) * 2;
  }
}

// The reads are inline asm: the compiler then neither waits for the pending LDS-DMA (it orders the tr-read
// builtin after every outstanding vmcnt) nor places its own lgkmcnt waits; the kernel waits by hand after the
// barrier (s_waitcnt lgkmcnt(0) + sched_barrier, cdna_hip_programming.md rule 18).
template <bool TR, int S> __device__ __forceinline__ void pp_read(bf16x8& dst, unsigned addr) {
  if (!TR) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr ^ (unsigned)(S << 5)));
  } else {
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(S * 4096));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(S * 4096 + 1024));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    dst = bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

template <bool TA, bool TB, bool TACC>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;

  const int nbx = (p.N + 255) / 256, nby = (p.M + 255) / 256;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int m0 = (bid / nbx) * 256, n0 = (bid % nbx) * 256;
  const int ktiles = p.K / 64;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int tbeg = blockIdx.z * per, tend = min(ktiles, tbeg + per);
  if (tbeg >= tend) return;
  const int nt = tend - tbeg, kbeg = tbeg * 64;
  const int nunits = 4 * nt;
  // diagnostic time stamps (p.atomic & 0x10000, p.aux_out = uint32 [8 waves][32]): workgroup 0, k-tile 3.
  // s_memtime returns through lgkmcnt; the values are only read after the kernel's own lgkmcnt(0) waits.
  const bool dbg = (p.atomic & 0x10000) && blockIdx.x == 0 && blockIdx.z == 0;
  unsigned ts[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) ts[i] = 0;
#define PP_STAMP(I) do { if (dbg_t) { unsigned long long t64_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64_) :: "memory"); ts[I] = (unsigned)t64_; } } while (0)
  { const bool dbg_t = dbg; PP_STAMP(19); }

  // staging sources: unit type J (0 A0, 1 B0, 2 B1, 3 A1), wave-instructions w and w + 8 of each
  const bf16* src[4][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    src[0][q] = pp_src<TA, 0>(A, p.lda, p.M, m0, kbeg, w + 8 * q, lane);
    src[1][q] = pp_src<TB, 1>(B, p.ldb, p.N, n0, kbeg, w + 8 * q, lane);
    src[2][q] = pp_src<TB, 2>(B, p.ldb, p.N, n0, kbeg, w + 8 * q, lane);
    src[3][q] = pp_src<TA, 3>(A, p.lda, p.M, m0, kbeg, w + 8 * q, lane);
  }
  const size_t stepa = TA ? (size_t)64 * p.lda : (size_t)64;
  const size_t stepb = TB ? (size_t)64 * p.ldb : (size_t)64;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void_p)smem);

  // issue unit type J of the next k-tile that type has not fetched yet into buffer BUF (0/1)
#define PP_ISSUE(J, BUF)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                            \
      __builtin_amdgcn_global_load_lds((glb_void_p)src[J][q],                                                  \
                                       (lds_void_p)(smem + (BUF) * 65536 + (J) * PP_UNIT + (w + 8 * q) * 1024), 16, 0, 0); \
      src[J][q] += ((J) == 0 || (J) == 3) ? stepa : stepb;                                                     \
    }                                                                                                          \
  } while (0)

  f32x16 acc[2][2][2];                                 // [A half][i][j]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  // fragment addresses (buffer 0); the other buffer is + 65536
  unsigned fa[2], fb;
  fa[0] = lds0 + pp_frag_off<TA>(wr * 64, lane);
  fa[1] = lds0 + pp_frag_off<TA>(wr * 64 + 32, lane);
  fb = lds0 + pp_frag_off<TB>(wc * 32, lane);

  // prologue: units 0..5 (k-tile 0 whole, A0 and B0 of k-tile 1); the host guarantees nt >= 2
  PP_ISSUE(0, 0); PP_ISSUE(1, 0); PP_ISSUE(2, 0); PP_ISSUE(3, 0); PP_ISSUE(0, 1); PP_ISSUE(1, 1);
  wait_vmcnt<8>();                                     // units 0 and 1
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();           // wave row 1 runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 a[2][4], b0[4], b1[4];
  { const bool dbg_t = dbg; PP_STAMP(20); }

#define PP_MFMA(H, JB, BREG)                                                                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
        if (TACC) acc[H][i][JB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BREG[s], a[i][s], acc[H][i][JB], 0, 0, 0); \
        else      acc[H][i][JB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][s], BREG[s], acc[H][i][JB], 0, 0, 0); \
      }                                                                                                        \
  } while (0)
#define PP_READ_A(BASEOFF)                                                                                     \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
      pp_read<TA, 0>(a[i][0], fa[i] + (BASEOFF)); pp_read<TA, 1>(a[i][1], fa[i] + (BASEOFF));                  \
      pp_read<TA, 2>(a[i][2], fa[i] + (BASEOFF)); pp_read<TA, 3>(a[i][3], fa[i] + (BASEOFF));                  \
    }                                                                                                          \
  } while (0)
#define PP_READ_B(BREG, BASEOFF)                                                                               \
  do {                                                                                                         \
    pp_read<TB, 0>(BREG[0], fb + (BASEOFF)); pp_read<TB, 1>(BREG[1], fb + (BASEOFF));                          \
    pp_read<TB, 2>(BREG[2], fb + (BASEOFF)); pp_read<TB, 3>(BREG[3], fb + (BASEOFF));                          \
  } while (0)
  // end of an L section: retire the units the next phase reads, publish, then wait for this phase's own reads
#define PP_SYNC_L(ALLOW, P)                                                                                    \
  do {                                                                                                         \
    if ((ALLOW) >= 4) wait_vmcnt<8>(); else if ((ALLOW) == 3) wait_vmcnt<6>(); else if ((ALLOW) == 2) wait_vmcnt<4>(); \
    else if ((ALLOW) == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
  } while (0)
#define PP_SYNC_C(P)                                                                                           \
  do {                                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  const bool no_dma = (p.atomic & 0x2000) != 0, no_rd = (p.atomic & 0x8000) != 0;   // diagnostics
  auto body = [&](auto tail_c, int kt) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const unsigned cb = (kt & 1) * 65536u;             // this k-tile's buffer
    const int nb = (kt & 1) ^ 1;                       // buffer of k-tile kt+1 (kt+2 shares this tile's)
    const int f0 = 4 * kt;                             // first phase; phase f issues unit f+6, may leave min(4, nunits-3-f) units in flight
    // ---- phase 0
    if (!no_rd) { PP_READ_A(cb + 0 * PP_UNIT); PP_READ_B(b0, cb + 1 * PP_UNIT); }
    if ((!TAIL || f0 + 6 < nunits) && !no_dma) PP_ISSUE(2, nb);
    PP_SYNC_L(TAIL ? nunits - 3 - f0 : 4, 0);
    PP_MFMA(0, 0, b0);
    PP_SYNC_C(0);
    // ---- phase 1
    if (!no_rd) PP_READ_B(b1, cb + 2 * PP_UNIT);
    if ((!TAIL || f0 + 7 < nunits) && !no_dma) PP_ISSUE(3, nb);
    PP_SYNC_L(TAIL ? nunits - 4 - f0 : 4, 1);
    PP_MFMA(0, 1, b1);
    PP_SYNC_C(1);
    // ---- phase 2
    if (!no_rd) PP_READ_A(cb + 3 * PP_UNIT);
    if ((!TAIL || f0 + 8 < nunits) && !no_dma) PP_ISSUE(0, nb ^ 1);
    PP_SYNC_L(TAIL ? nunits - 5 - f0 : 4, 2);
    PP_MFMA(1, 1, b1);
    PP_SYNC_C(2);
    // ---- phase 3
    if ((!TAIL || f0 + 9 < nunits) && !no_dma) PP_ISSUE(1, nb ^ 1);
    PP_SYNC_L(TAIL ? nunits - 6 - f0 : 4, 3);
    PP_MFMA(1, 0, b0);
    if (TAIL && kt == nt - 1) {                        // wave row 1 has no partner barrier left after its last C section
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (wr == 0) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      PP_SYNC_C(3);
    }
  };
  if (p.atomic & 0x4000) {                             // diagnostic: epilogue only
    wait_vmcnt<0>();
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();
  } else {
    int kt = 0;
    for (; kt + 2 < nt; ++kt) body(std::false_type{}, kt);
    for (; kt < nt; ++kt) body(std::true_type{}, kt);
  }
#undef PP_ISSUE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B
#undef PP_SYNC_L
#undef PP_SYNC_C
  // (the epilogue's LDS staging is wave-private and no wave reads the ring after its last L section, so wave
  //  row 0 starts storing while wave row 1 is still in its last C section)
  { const bool dbg_t = dbg; PP_STAMP(21); }
  if (p.atomic & 0x800) return;                        // diagnostic: main loop only
  if (TACC) {                                          // bf16 output, aligned (host-checked): straight from the registers
    bf16_tile_epilogue_direct(p, acc[0], m0 + wr * 128, n0 + wc * 64, lane);
    bf16_tile_epilogue_direct(p, acc[1], m0 + wr * 128 + 64, n0 + wc * 64, lane);
  } else {                                             // fp32 output (accumulate / split-K atomics): row segments per register
    bf16_tile_epilogue<false>(p, acc[0], m0 + wr * 128, n0 + wc * 64, 0, 0, lane, smem);
    bf16_tile_epilogue<false>(p, acc[1], m0 + wr * 128 + 64, n0 + wc * 64, 0, 0, lane, smem);
  }
  if (dbg) {
    { const bool dbg_t = true; PP_STAMP(22); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
      unsigned* o = reinterpret_cast<unsigned*>(p.aux_out) + w * 32;
#pragma unroll
      for (int i = 0; i < 32; ++i) o[i] = ts[i];
    }
  }
#undef PP_STAMP
}

template <bool TA, bool TB, bool TACC>
void pp_launch1(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 131072;
  auto kern = gemm_bf16_pp_kernel<TA, TB, TACC>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nwg = ((p.N + 255) / 256) * ((p.M + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(nwg, 1, p.split_k), dim3(512), smem, st, p);
}

template void pp_launch1<false, false, false>(const GemmArgs&, hipStream_t);
template void pp_launch1<false, false, true>(const GemmArgs&, hipStream_t);
template void pp_launch1<false, true, false>(const GemmArgs&, hipStream_t);
template void pp_launch1<false, true, true>(const GemmArgs&, hipStream_t);
template void pp_launch1<true, false, false>(const GemmArgs&, hipStream_t);
template void pp_launch1<true, false, true>(const GemmArgs&, hipStream_t);
template void pp_launch1<true, true, false>(const GemmArgs&, hipStream_t);
template void pp_launch1<true, true, true>(const GemmArgs&, hipStream_t);
