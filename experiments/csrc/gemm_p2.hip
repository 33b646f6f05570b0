#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// Variant 14: one wave per SIMD, and the epilogue of work item i runs in the MFMA gaps of item i + 1.
//
// The ping-pong kernels (variants 8 / 12) hide fragment reads and LDS-DMA issue behind a partner wave, but not the epilogue: both
// waves of a SIMD finish a tile together and the matrix pipe idles while they convert, transpose and store -- 57 % of the time of
// the K = 768 GEMM that applies GELU and saves gelu' (profiles/r04_pmc_summary.json: MFMA-busy 43 %).  Two waves per SIMD leave a
// wave 256 registers, 128 of them accumulators: no room to keep a finished tile while the next one accumulates.  Here a wave is
// alone on its SIMD and has 512:
//   AGPRs  128 accumulators (128 x 64 per wave, 2 x 2 waves -> tile 256 x 128) + 96 fragment registers (ds_read_b128 loads AGPRs
//          directly, MFMA reads its A / B operands from them);
//   VGPRs  128 `P` = the PREVIOUS item's finished accumulators, copied out during that item's last k-tile (v_accvgpr_read in the
//          MFMA gaps, one phase behind each block's last MFMA), + addresses + the epilogue's temporaries.
// The item body is fully unrolled (K = 768: 12 k-tiles x 64 MFMAs) and GENERATED: tools/gen_p2_body.py places every fragment read,
// LDS-DMA piece, epilogue step, LDS transposition and global store after a specific MFMA and derives every counted vmcnt from that
// order (gemm_p2_body{1,2}.inc).  An MFMA 16x16x32 blocks the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md,
// instruction-issue table): two full-rate vector instructions per gap are free, what exceeds that stretches the gap -- the GELU +
// gelu' epilogue is ~4.2 issue slots per gap over 576 of the 768 gaps.
//   Stream: as gemm_p1.hip -- k-tile t stages k-tile t + 2 (A0, B0 behind barrier X_t; B1, A1 behind Y_t) and runs on into the
//   next item; two 48 KiB k-tile buffers; LDS image and swizzles of gemm_pp16.h.
//   First item of a workgroup: no pending epilogue (its steps run on garbage, the stores are skipped, the vmcnt counts without the
//   stores are generated too).  After the last item the body runs once more on the same item (a "ghost": its own result is never
//   stored) to carry the last real epilogue -- one tile per workgroup and launch (~0.5 % at the sizes this kernel is for).
//   Bias: added by the epilogue (the accumulators start at zero: the first MFMA of a block in k-tile 0 has C = 0), so results differ
//   from variants 8 / 12 -- which start the fp32 sums at the bias -- by fp32 rounding: equal to them up to one bf16 ulp on a few
//   elements, not bit for bit.
// k-contiguous operands, K = 768, M % 256 == 0, N % 128 == 0, bf16 output; epilogues: none, GELU + gelu' (UC2_GEMM_AUX_DERIV).
// ------------------------------------------------------------------------------------------------------
#include "gemm_pp16.h"

#define P2_BUF 49152
#define P2_TP0 98304             /* transposition buffers behind the ring: 8 KiB per wave (output stream, second stream) */

struct P2Ep {                    // temporaries of the piece in flight (all in registers: every step is inlined)
  float x, xc, x2, q, ph, sp;
  float g[4], d[4];
};

// step S of piece PZ of the pending item's epilogue.  PZ = 8 b + 4 mbl + nb: 32-row block b = 2 hh + i, 16-row block mb = 2 i + mbl of
// A half hh, 16-column block nb; the lane's four values are columns 16 nb + 4 g .. + 3 of row 16 mbl + (lane & 15) of the 32-row block.
// The arithmetic is gelu_and_dgelu_bf's (common.h), operation for operation; here x = accumulator + bias.
template <int EPI, int PZ, int S>
__device__ __forceinline__ void p2_ep_step(P2Ep& c, const float (&P)[2][4][4][4], const f32x4 (&biasv)[4], const unsigned (&qa)[4]) {
  constexpr int b = PZ / 8, k = PZ % 8, hh = b >> 1, i = b & 1, mbl = k >> 2, nb = k & 3, mb = 2 * i + mbl;
  if constexpr (S < 16) {
    constexpr int e = S >> 2, st = S & 3;
    if constexpr (EPI == EPI_GELU_D) {
      if constexpr (st == 0) {
        c.x = P[hh][mb][nb][e] + biasv[nb][e];
        c.xc = __builtin_amdgcn_fmed3f(c.x, -9.0f, 9.0f);
        c.x2 = c.xc * c.xc;
        c.q = fmaf(-UC2_PHI_C2 * UC2_LOG2E, c.x2, -UC2_PHI_C1 * UC2_LOG2E);
      } else if constexpr (st == 1) {
        c.q = fmaf(c.q, c.x2, -UC2_PHI_C0 * UC2_LOG2E);
        c.q = __builtin_amdgcn_exp2f(c.q * c.xc);
      } else if constexpr (st == 2) {
        c.ph = __builtin_amdgcn_rcpf(1.0f + c.q);
        c.sp = fmaf(fmaf(5.0f * UC2_PHI_C2, c.x2, 3.0f * UC2_PHI_C1), c.x2, UC2_PHI_C0);
      } else {
        const float t = fmaf(-c.ph, c.ph, c.ph);
        c.g[e] = c.x * c.ph;
        c.d[e] = fmaf(c.xc * c.sp, t, c.ph);
      }
    } else {
      if constexpr (st == 0) c.g[e] = P[hh][mb][nb][e] + biasv[nb][e];
    }
  } else if constexpr (S == 16) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)c.g[e];
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(qa[nb]), "v"(o), "n"(mbl * 2048) : "memory");
  } else if constexpr (S == 17 && EPI == EPI_GELU_D) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)c.d[e];
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(qa[nb]), "v"(o), "n"(mbl * 2048 + 4096) : "memory");
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_p2_kernel(GemmArgs p) {
  constexpr bool TWO = EPI == EPI_GELU_D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* const A = reinterpret_cast<const bf16*>(p.A);
  const bf16* const B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 1, wc = w & 1;

  // ---- work items: 256 x 128 tiles, XCD-contiguous, column tiles in groups (two per 256-column group of the other kernels)
  const int nbx = p.N / 128, mt = p.M / 256, nitems = nbx * mt;
  int item, item_end, item_step;
  {
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nx = min(G, 8);
    const int q = nitems / nx, r = nitems % nx;
    const int beg = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    item_end = beg + q + (xcd < r ? 1 : 0);
    item_step = (G - xcd + 7) >> 3;
    item = beg + slot;
  }
  if (item >= item_end) return;

  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void_p)smem);
  unsigned fa[2][2], fb[2][2];                         // fragment bases [k-tile buffer][k-step]; block and unit are immediate offsets
  fa[0][0] = lds0 + pp16_frag_off<false>(wr * 64, lane); fa[0][1] = fa[0][0] ^ 64u;
  fb[0][0] = lds0 + pp16_frag_off<false>(wc * 32, lane); fb[0][1] = fb[0][0] ^ 64u;
  fa[1][0] = fa[0][0] + P2_BUF; fa[1][1] = fa[0][1] + P2_BUF;
  fb[1][0] = fb[0][0] + P2_BUF; fb[1][1] = fb[0][1] + P2_BUF;

  int m0, n0;                   // the item being computed
  int pm0 = 0, pn0 = 0;         // the pending item (its epilogue runs now)
  int m0x = 0, n0x = 0;         // the item being staged
  unsigned srcA0[4], srcA1[4], srcB0[2], srcB1[2];     // staging sources of the staged item at k = 0 (byte offsets from A / B)
  const int cg = min(2 * p.col_group, nbx);
  auto setup = [&](int it) __attribute__((always_inline)) {
    const int per_group = mt * cg;
    const int g = it / per_group, r = it - g * per_group;
    const int cw = min(cg, nbx - g * cg);
    const int tm = r / cw, tc = r - tm * cw;
    m0x = tm * 256; n0x = (g * cg + tc) * 128;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      srcA0[q] = (unsigned)((const char*)pp_src<false, 0, 2>(A, p.lda, p.M, m0x, 0, w + 4 * q, ln) - (const char*)A);
      srcA1[q] = (unsigned)((const char*)pp_src<false, 3, 2>(A, p.lda, p.M, m0x, 0, w + 4 * q, ln) - (const char*)A);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      srcB0[q] = (unsigned)((const char*)pp_src<false, 1, 2>(B, p.ldb, p.N, n0x, 0, w + 4 * q, ln) - (const char*)B);
      srcB1[q] = (unsigned)((const char*)pp_src<false, 2, 2>(B, p.ldb, p.N, n0x, 0, w + 4 * q, ln) - (const char*)B);
    }
  };
  const unsigned wlds = lds0 + (unsigned)w * 1024u;
  // one LDS-DMA piece (1 KiB): per-lane offset and the wave's LDS base through opaque copies (see gemm_p1.hip)
#define P2_DMA(BASE, SRC, UOFF, Q, BUFI, KTS)                                                                   \
  do {                                                                                                         \
    unsigned so_ = SRC[Q], wl_ = wlds;                                                                         \
    const char* ub_ = (const char*)(BASE) + (KTS) * 128;            /* uniform base + per-lane 32-bit offset: the saddr form */ \
    asm volatile("" : "+v"(so_), "+s"(wl_), "+s"(ub_));                                                        \
    __builtin_amdgcn_global_load_lds((glb_void_p)(ub_ + so_),                                                  \
                                     (lds_void_p)(uintptr_t)(wl_ + (unsigned)((BUFI) * P2_BUF + (UOFF) + 4096 * (Q))), 16, 0, 0); \
  } while (0)
  const bool dg_nodma = (p.atomic & 0x100) != 0, dg_nord = (p.atomic & 0x200) != 0;     // timing diagnostics (with 0x800): no LDS-DMA / no fragment reads
#define P2_DMA_A0(Q, BUFI, KTS, X) do { if (!dg_nodma) P2_DMA(A, srcA0, 0, Q, BUFI, KTS); } while (0)
#define P2_DMA_B0(Q, BUFI, KTS, X) do { if (!dg_nodma) P2_DMA(B, srcB0, 16384, Q, BUFI, KTS); } while (0)
#define P2_DMA_B1(Q, BUFI, KTS, X) do { if (!dg_nodma) P2_DMA(B, srcB1, 24576, Q, BUFI, KTS); } while (0)
#define P2_DMA_A1(Q, BUFI, KTS, X) do { if (!dg_nodma) P2_DMA(A, srcA1, 32768, Q, BUFI, KTS); } while (0)

  f32x4 accM[2][4][4];                                 // [A half][16-row block][16-column block], AGPRs
  bf16x8 a0[4][2], a1[4][2], bx[2][2], by[2][2];       // fragment sets [block][k-step], AGPRs; bx / by swap roles (B0 / B1) every k-tile
  float P[2][4][4][4];                                 // the pending item's accumulators, VGPRs
  f32x4 biasv[4];                                      // bias of the lane's columns 16 nb + 4 g .. + 3 (pending item from MFMA gap 32 on)
  P2Ep epc;
  bf16x8 ot[4], pt[4];                                 // a 32-row block on its way from the transposition buffer to memory

  // epilogue addressing (fixed per lane)
  const int g4 = lane >> 4, r15 = lane & 15, lr = lane >> 3, lc = lane & 7;
  const unsigned tb = lds0 + P2_TP0 + (unsigned)w * 8192u;
  unsigned qa[4];
  {
    const unsigned q0 = tb + (unsigned)r15 * 128u + ((unsigned)((g4 >> 1) ^ (r15 & 7)) << 4) + 8u * (unsigned)(g4 & 1);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) qa[nb] = q0 ^ (unsigned)(nb << 5);
  }
  const unsigned tline = tb + (unsigned)(lr * 128 + ((lc ^ lr) << 4));
  const unsigned vst_c = (unsigned)((lr * p.ldc + 8 * lc) * 2), vst_x = (unsigned)((lr * p.ldaux + 8 * lc) * 2);
  const unsigned vbias = (unsigned)(16 * g4);
  const bool has_bias = p.bias != nullptr;
  const float* const bias_base = has_bias ? p.bias : reinterpret_cast<const float*>(p.A);      // (no bias: load anything valid, zeroed below)

#define P2_MF(AH, MB, NB, BREG, AREG)  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accM[AH][MB][NB]) : "a"(BREG), "a"(AREG))
#define P2_MF0(AH, MB, NB, BREG, AREG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(accM[AH][MB][NB]) : "a"(BREG), "a"(AREG))
#define P2_RDA(DST, KS, BUFI, OFF) do { if (!dg_nord) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+a"(DST) : "v"(fa[BUFI][KS]), "n"(OFF)); } while (0)
#define P2_RDB(DST, KS, BUFI, OFF) do { if (!dg_nord) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+a"(DST) : "v"(fb[BUFI][KS]), "n"(OFF)); } while (0)
#define P2_GAP() __builtin_amdgcn_sched_barrier(0)
#define P2_PHASE_BEGIN() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P2_WAITB(NP, NN)                                                                                        \
  do {                                                                                                         \
    if ((NP) == (NN) || pending) wait_vmcnt<NP>(); else wait_vmcnt<NN>();                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define P2_EP(PZ, S) p2_ep_step<EPI, PZ, S>(epc, P, biasv, qa)
#define P2_CP(AH, MB, NB, E0)                                                                                   \
  do {                                                                                                         \
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(P[AH][MB][NB][E0]) : "a"(accM[AH][MB][NB][E0]));           \
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(P[AH][MB][NB][(E0) + 1]) : "a"(accM[AH][MB][NB][(E0) + 1])); \
  } while (0)
  // transposition buffer -> registers, whole 128-byte lines (stream 0: output, 1: second stream)
#define P2_TPR(BLK, WHICH)                                                                                      \
  do {                                                                                                         \
    if ((WHICH) == 0) { tp_read_o<0>(ot[0], tline); tp_read_o<1024>(ot[1], tline); tp_read_o<2048>(ot[2], tline); tp_read_o<3072>(ot[3], tline); } \
    else { tp_read_o<4096>(pt[0], tline); tp_read_o<5120>(pt[1], tline); tp_read_o<6144>(pt[2], tline); tp_read_o<7168>(pt[3], tline); } \
  } while (0)
  // rows 8 it + (lane >> 3) of 32-row block BLK of the pending item: scalar base + per-lane offset, no vector arithmetic.
  // (s_nop: a vector instruction that writes the data registers of a store of more than 8 bytes right behind it corrupts the store
  //  -- hipcc pads its own stores, it cannot see into inline asm; found as 1.2 % wrong elements, always dword 0 of a 16-byte piece)
#define P2_ST(BLK, IT, WHICH)                                                                                   \
  do {                                                                                                         \
    if (pending) {                                                                                             \
      const int row_ = pm0 + wr * 128 + ((BLK) >> 1) * 64 + ((BLK) & 1) * 32 + 8 * (IT);                        \
      if ((WHICH) == 0) {                                                                                      \
        const char* u_ = reinterpret_cast<const char*>(p.C) + ((size_t)row_ * p.ldc + pn0 + wc * 64) * 2;       \
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(vst_c), "v"(ot[IT]), "s"(u_) : "memory");      \
      } else {                                                                                                 \
        const char* u_ = reinterpret_cast<const char*>(p.aux_out) + ((size_t)row_ * p.ldaux + pn0 + wc * 64) * 2; \
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(vst_x), "v"(pt[IT]), "s"(u_) : "memory");      \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
#define P2_BIASLD(NB)                                                                                           \
  do {                                                                                                         \
    const char* u_ = reinterpret_cast<const char*>(bias_base + (has_bias ? n0 + wc * 64 : 0));                  \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(biasv[NB]) : "v"(vbias), "s"(u_), "n"((NB) * 64) : "memory"); \
  } while (0)

  // no bias: the dummy loads have landed behind barrier Y of k-tile 0 (they are older than the B0 pieces it waits for); zero them there
#define P2_BIASFIX() do { if (!has_bias) { biasv[0] = f32x4{0.f, 0.f, 0.f, 0.f}; biasv[1] = biasv[0]; biasv[2] = biasv[0]; biasv[3] = biasv[0]; } } while (0)
  bool pending = false, ghost = false, will_ghost = false;
  int nxt = 0;
#define P2_SETUP_NEXT()                                                                                         \
  do {                                                                                                         \
    const int cand_ = item + item_step;                                                                        \
    if (!ghost && cand_ < item_end) { nxt = cand_; will_ghost = false; } else { nxt = item; will_ghost = !ghost; } \
    setup(nxt);                                                                                                \
  } while (0)

  // ---- prologue: the vector-memory operations of phases 40 .. 47 of the body, in their order (the counted waits of k-tiles 0 and 1
  //      look back at them): k-tile 0 (A0 B0 B1, the bias loads, A1), then k-tile 1
  setup(item);
  m0 = m0x; n0 = n0x;
#define P2_PRO(BASE, SRC, UOFF, NQ, BUFI, KTS) _Pragma("unroll") for (int q = 0; q < (NQ); ++q) P2_DMA(BASE, SRC, UOFF, q, BUFI, KTS)
  P2_PRO(A, srcA0, 0, 4, 0, 0);
  P2_PRO(B, srcB0, 16384, 2, 0, 0);
  P2_PRO(B, srcB1, 24576, 2, 0, 0);
  P2_BIASLD(0); P2_BIASLD(1); P2_BIASLD(2); P2_BIASLD(3);
  P2_PRO(A, srcA1, 32768, 4, 0, 0);
  P2_PRO(A, srcA0, 0, 4, 1, 1);
  P2_PRO(B, srcB0, 16384, 2, 1, 1);
  P2_PRO(B, srcB1, 24576, 2, 1, 1);
  P2_PRO(A, srcA1, 32768, 4, 1, 1);
#undef P2_PRO
  wait_vmcnt<22>();                                    // A0 and B0 of k-tile 0 have landed (22 operations are younger)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // fragments phase 0 of k-tile 0 starts with: A0 -> a0, B0 -> bx
  P2_RDA(a0[0][0], 0, 0, 0); P2_RDA(a0[1][0], 0, 0, 2048); P2_RDA(a0[2][0], 0, 0, 4096); P2_RDA(a0[3][0], 0, 0, 6144);
  P2_RDA(a0[0][1], 1, 0, 0); P2_RDA(a0[1][1], 1, 0, 2048); P2_RDA(a0[2][1], 1, 0, 4096); P2_RDA(a0[3][1], 1, 0, 6144);
  P2_RDB(bx[0][0], 0, 0, 16384); P2_RDB(bx[1][0], 0, 0, 16384 + 2048); P2_RDB(bx[0][1], 1, 0, 16384); P2_RDB(bx[1][1], 1, 0, 16384 + 2048);
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
    if constexpr (TWO) {
#include "gemm_p2_body2.inc"
    } else {
#include "gemm_p2_body1.inc"
    }
    if (ghost) break;
    pm0 = m0; pn0 = n0;
    m0 = m0x; n0 = n0x; item = nxt;
    pending = !(p.atomic & 0x800);                     // (diagnostic 0x800: main loop only, nothing is stored)
    ghost = will_ghost;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ghost item staged two more k-tiles: nothing may land in LDS after the wave has left
#undef P2_DMA
}

static int p2_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI>
static void p2_launch0(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = P2_TP0 + 4 * 8192;            // two 48 KiB k-tile buffers + two 4 KiB transposition buffers per wave = 128 KiB
  auto kern = gemm_bf16_p2_kernel<EPI>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = (p.N / 128) * (p.M / 256);
  int cus = p2_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, p);
}

// what variant 14 takes (the caller has checked whole 256-row tiles, N % 256 == 0, the 32-bit staging offsets, 16-byte alignments)
bool uc2_gemm_p2_supported(const GemmArgs& p, int trans_a, int trans_b) {
  if (trans_a || trans_b || p.c_f32 || p.split_k != 1 || p.queue || p.K != 768) return false;
  if (p.epi == EPI_NONE) return true;
  return p.epi == EPI_GELU && p.aux_deriv && p.aux_out != nullptr;
}

void uc2_gemm_p2_launch(const GemmArgs& p, hipStream_t st) {
  if (p.epi == EPI_GELU) p2_launch0<EPI_GELU_D>(p, st);
  else p2_launch0<EPI_NONE>(p, st);
}
