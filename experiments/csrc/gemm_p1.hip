#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// Variant 13: the 256 x 256 x 64 tile of the ping-pong kernels on ONE wave per SIMD (4 waves, 512 registers each).
// A wave owns 128 x 128 of the output (64 accumulator blocks of v_mfma_f32_16x16x32_bf16 = 256 AGPRs), so a k-tile costs it
// 32 KiB of fragment reads for 128 MFMAs -- two thirds of the LDS bytes per flop of the 128 x 64 waves of variants 8 / 12 (the
// GEMMs are power-limited: DESIGN.md section 4.1) -- and there is no partner wave: fragment reads, LDS-DMA issue and the two
// barriers of a k-tile sit in the wave's own MFMA stream, placed by hand between the MFMAs.
//   LDS image, swizzles, staging sources: gemm_pp.h / gemm_pp16.h (units A0, B0, B1, A1 of 16 KiB per k-tile buffer, two buffers);
//   here B0 / B1 hold 64 columns per wave column (the A mapping), not 32.
//   Stream: k-tile t issues the units of k-tile t + 2 into its own buffer -- A0, B0 (the first half F) in phases 0 / 1, behind
//   barrier X_t, B1, A1 (the second half S) in phases 2 / 3, behind barrier Y_t -- and runs straight on into the next work item.
//   X_t: every wave has consumed F_t and its S_t pieces have landed (vmcnt 16); Y_t: S_t consumed, F_{t+1} landed.
//   Phases: (A0,B0) (A0,B1) (A1,B1) (A1,B0), 32 MFMAs each; phase p reads the fragment set phase p + 1 starts with.
// k-contiguous operands only (Y = X W^T and the input gradients on W^T), bf16 output, whole tiles, K % 128 == 0, no split-K.
// ------------------------------------------------------------------------------------------------------
#include "gemm_pp16.h"

template <> __device__ __forceinline__ void wait_vmcnt<32>() { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<48>() { asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<63>() { asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); }
// all but the n youngest vector-memory operations done; rounded DOWN to a step of the ladder (stricter, never laxer)
__device__ __forceinline__ void p1_wait(int n) {
  if (n >= 63) wait_vmcnt<63>(); else if (n >= 48) wait_vmcnt<48>(); else if (n >= 32) wait_vmcnt<32>(); else if (n >= 24) wait_vmcnt<24>();
  else if (n >= 16) wait_vmcnt<16>(); else if (n >= 8) wait_vmcnt<8>(); else wait_vmcnt<0>();
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_p1_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* const A = reinterpret_cast<const bf16*>(p.A);
  const bf16* const B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 1, wc = w & 1;
  constexpr int NSTORE = (EPI == EPI_GELU || EPI == EPI_GELU_D) ? 64 : 32;      // epilogue stores per wave and item

  // ---- work items: tiles, XCD-contiguous, column tiles in groups (gemm_pp16.hip)
  const int nbx = p.N / 256, mt = p.M / 256, nitems = nbx * mt;
  const int nt = p.K / 64;                             // k-tiles per item: even (host-checked)
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
  fb[0][0] = lds0 + pp16_frag_off<false>(wc * 64, lane); fb[0][1] = fb[0][0] ^ 64u;
  fa[1][0] = fa[0][0] + 65536u; fa[1][1] = fa[0][1] + 65536u;
  fb[1][0] = fb[0][0] + 65536u; fb[1][1] = fb[0][1] + 65536u;

  int m0, n0;                   // the item being computed
  int m0x = 0, n0x = 0;         // the item being staged
  unsigned src[4][4];           // staging sources of the staged item at k = 0: byte offsets from A / B; unit (0 A0, 1 B0, 2 B1, 3 A1), piece q
  auto setup = [&](int it) __attribute__((always_inline)) {
    const int cg = p.col_group;
    const int per_group = mt * cg;
    const int g = it / per_group, r = it - g * per_group;
    const int cw = min(cg, nbx - g * cg);
    const int tm = r / cw, tc = r - tm * cw;
    m0x = tm * 256; n0x = (g * cg + tc) * 256;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      src[0][q] = (unsigned)((const char*)pp_src<false, 0, 2>(A, p.lda, p.M, m0x, 0, w + 4 * q, ln) - (const char*)A);
      src[3][q] = (unsigned)((const char*)pp_src<false, 3, 2>(A, p.lda, p.M, m0x, 0, w + 4 * q, ln) - (const char*)A);
      src[1][q] = (unsigned)((const char*)pp_src<false, 0, 2>(B, p.ldb, p.N, n0x, 0, w + 4 * q, ln) - (const char*)B);
      src[2][q] = (unsigned)((const char*)pp_src<false, 3, 2>(B, p.ldb, p.N, n0x, 0, w + 4 * q, ln) - (const char*)B);
    }
  };
  // one piece (1 KiB) of unit U of the staged k-tile (byte offset KOFF along k) into buffer BUF.  The per-lane offset and the
  // wave's LDS base go through opaque copies: hoisted out of the loop, hipcc keeps 16 zero-extended 64-bit offsets and 32 M0
  // values alive (and spills them) instead of one add per piece
  unsigned wlds = lds0 + (unsigned)w * 1024u;
#define P1_DMA(U, Q, BUF, KOFF)                                                                                 \
  do {                                                                                                         \
    unsigned so_ = src[U][Q], wl_ = wlds;                                                                      \
    asm volatile("" : "+v"(so_), "+s"(wl_));                                                                   \
    __builtin_amdgcn_global_load_lds((glb_void_p)((const char*)(((U) == 0 || (U) == 3) ? (const void*)A : (const void*)B) + (KOFF) + so_), \
                                     (lds_void_p)(uintptr_t)(wl_ + (unsigned)((BUF) * 65536 + (U) * PP_UNIT + 4096 * (Q))), 16, 0, 0); \
  } while (0)

  f32x4 acc[2][2][4][4];                               // [B half][A half][16-row block][16-column block]
  bf16x8 a0[4][2], a1[4][2], bx[4][2], by[4][2];       // fragment sets [block][k-step]; the B sets swap roles (B0 / B1) every k-tile

  // inline asm: the accumulators stay where they are (as builtins hipcc moved accumulator blocks between registers, read them
  // back into VGPRs and spilled them inside the loop); the two places where the hardware does not interlock an MFMA result
  // against vector instructions -- accumulator initialisation before the loop, the epilogue's reads after it -- get s_nops
#define P1_MF(AH, BH, AREG, BREG, I)                                                                            \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[BH][AH][((I) >> 2) & 3][(I) & 3])           \
               : "v"(BREG[(I) & 3][(I) >> 4]), "v"(AREG[((I) >> 2) & 3][(I) >> 4]))
  // four MFMAs, one fragment read after the second, (odd groups) one LDS-DMA piece after the fourth
#define P1_GROUP(G, AH, BH, AREG, BREG, RD, DM)                                                                 \
  do {                                                                                                         \
    P1_MF(AH, BH, AREG, BREG, 4 * (G)); P1_MF(AH, BH, AREG, BREG, 4 * (G) + 1);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    RD;                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    P1_MF(AH, BH, AREG, BREG, 4 * (G) + 2); P1_MF(AH, BH, AREG, BREG, 4 * (G) + 3);                            \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    DM;                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
  // read G (0..7) of a fragment set: block G & 3, k-step G >> 2
#define P1_RD(DST, FBASE, BUF, UNIT, G, ON)                                                                     \
  do { if (ON) { if (((G) >> 2) == 0) pp16_read<false, (G) & 3, 0, (UNIT) * PP_UNIT>(DST[(G) & 3][0], FBASE[BUF][0]);          \
                 else pp16_read<false, (G) & 3, 1, (UNIT) * PP_UNIT>(DST[(G) & 3][1], FBASE[BUF][1]); } } while (0)
#ifndef P1_READS_EARLY
#define P1_READS_EARLY 1
#endif
  // two MFMAs then one fragment read (first half of a phase: the reads have 16 MFMAs to come back before the next phase's wait)
#define P1_PAIR(I, AH, BH, AREG, BREG, RD)                                                                      \
  do {                                                                                                         \
    P1_MF(AH, BH, AREG, BREG, I); P1_MF(AH, BH, AREG, BREG, (I) + 1);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    RD;                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
  // four MFMAs then one LDS-DMA piece (second half)
#define P1_QUAD(I, AH, BH, AREG, BREG, DM)                                                                      \
  do {                                                                                                         \
    P1_MF(AH, BH, AREG, BREG, I); P1_MF(AH, BH, AREG, BREG, (I) + 1); P1_MF(AH, BH, AREG, BREG, (I) + 2); P1_MF(AH, BH, AREG, BREG, (I) + 3); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    DM;                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#if P1_READS_EARLY
#define P1_PHASE(AH, BH, AREG, BREG, RDST, RBASE, RBUF, RUNIT, RON, DU, DBUF, DKOFF, DON)                         \
  do {                                                                                                         \
    P1_PAIR(0, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 0, RON));                                    \
    P1_PAIR(2, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 1, RON));                                    \
    P1_PAIR(4, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 2, RON));                                    \
    P1_PAIR(6, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 3, RON));                                    \
    P1_PAIR(8, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 4, RON));                                    \
    P1_PAIR(10, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 5, RON));                                   \
    P1_PAIR(12, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 6, RON));                                   \
    P1_PAIR(14, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 7, RON));                                   \
    P1_QUAD(16, AH, BH, AREG, BREG, do { if (DON) P1_DMA(DU, 0, DBUF, DKOFF); } while (0));                      \
    P1_QUAD(20, AH, BH, AREG, BREG, do { if (DON) P1_DMA(DU, 1, DBUF, DKOFF); } while (0));                      \
    P1_QUAD(24, AH, BH, AREG, BREG, do { if (DON) P1_DMA(DU, 2, DBUF, DKOFF); } while (0));                      \
    P1_QUAD(28, AH, BH, AREG, BREG, do { if (DON) P1_DMA(DU, 3, DBUF, DKOFF); } while (0));                      \
  } while (0)
#else
#define P1_PHASE(AH, BH, AREG, BREG, RDST, RBASE, RBUF, RUNIT, RON, DU, DBUF, DKOFF, DON)                         \
  do {                                                                                                         \
    P1_GROUP(0, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 0, RON), (void)0);                           \
    P1_GROUP(1, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 1, RON), do { if (DON) P1_DMA(DU, 0, DBUF, DKOFF); } while (0)); \
    P1_GROUP(2, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 2, RON), (void)0);                           \
    P1_GROUP(3, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 3, RON), do { if (DON) P1_DMA(DU, 1, DBUF, DKOFF); } while (0)); \
    P1_GROUP(4, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 4, RON), (void)0);                           \
    P1_GROUP(5, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 5, RON), do { if (DON) P1_DMA(DU, 2, DBUF, DKOFF); } while (0)); \
    P1_GROUP(6, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 6, RON), (void)0);                           \
    P1_GROUP(7, AH, BH, AREG, BREG, P1_RD(RDST, RBASE, RBUF, RUNIT, 7, RON), do { if (DON) P1_DMA(DU, 3, DBUF, DKOFF); } while (0)); \
  } while (0)
#endif
#define P1_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P1_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  bool more = false;            // another item follows the current one
  int nxt = 0;
  int est = 0;                  // epilogue stores of the previous item that are younger than the units the first waits of an item need
  // one k-tile.  SW: k-tile parity = its buffer; GEN: the general form (the two k-tiles behind an epilogue, the two at an item's
  // end); the steady-state form has every count a constant
  auto body = [&](auto sw_c, auto gen_c, int kt) __attribute__((always_inline)) {
    constexpr int SW = decltype(sw_c)::value;
    constexpr bool GEN = decltype(gen_c)::value;
    bf16x8 (&b0)[4][2] = SW ? by : bx;
    bf16x8 (&b1)[4][2] = SW ? bx : by;
    constexpr int CB = SW, NB = SW ^ 1;
    const int ts = kt + 2;
    const bool cross = GEN && ts >= nt;                 // the staged k-tile belongs to the next item
    const bool stage = !GEN || !cross || more;
    const bool last = GEN && kt == nt - 1;
    const bool rd_next = !last || more;                 // k-tile kt + 1 exists
    if (GEN && cross && ts == nt && more) setup(nxt);
    const unsigned koff = (unsigned)(cross ? ts - nt : ts) * 128u;
    int allow_x = 16, allow_y = 16;
    if (GEN) {
      const bool ex1 = kt + 1 < nt || more, ex2 = kt + 2 < nt || more;
      allow_x = (ex1 ? 16 : 0) + (kt <= 1 ? est : 0);
      allow_y = (ex1 ? 8 : 0) + (ex2 ? 8 : 0) + (kt == 0 ? est : 0);
    }
    // ---- phase 0: A0 x B0; read B1(kt); stage A0(kt + 2)
    P1_LGKM0();
    if (GEN) p1_wait(allow_x); else wait_vmcnt<16>();
    P1_BARRIER();
    P1_PHASE(0, 0, a0, b0, b1, fb, CB, 2, true, 0, CB, koff, stage);
    // ---- phase 1: A0 x B1; read A1(kt); stage B0(kt + 2)
    P1_LGKM0();
    P1_PHASE(0, 1, a0, b1, a1, fa, CB, 3, true, 1, CB, koff, stage);
    // ---- phase 2: A1 x B1; read A0(kt + 1); stage B1(kt + 2)
    P1_LGKM0();
    if (GEN) p1_wait(allow_y); else wait_vmcnt<16>();
    P1_BARRIER();
    P1_PHASE(1, 1, a1, b1, a0, fa, NB, 0, rd_next, 2, CB, koff, stage);
    // ---- phase 3: A1 x B0; read B0(kt + 1) into the registers of B1; stage A1(kt + 2)
    P1_LGKM0();
    P1_PHASE(1, 0, a1, b0, b1, fb, NB, 1, rd_next, 3, CB, koff, stage);
    if (GEN && kt == 1) est = 0;
  };

  // ---- first item: the whole ring (k-tiles 0 and 1), then the fragments phase 0 starts with
  setup(item);
  m0 = m0x; n0 = n0x;
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(0, q, 0, 0u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(1, q, 0, 0u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(2, q, 0, 0u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(3, q, 0, 0u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(0, q, 1, 128u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(1, q, 1, 128u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(2, q, 1, 128u);
#pragma unroll
  for (int q = 0; q < 4; ++q) P1_DMA(3, q, 1, 128u);
  wait_vmcnt<24>();
  P1_BARRIER();
#define P1_RD8(DST, FBASE, BUF, UNIT) do { P1_RD(DST, FBASE, BUF, UNIT, 0, true); P1_RD(DST, FBASE, BUF, UNIT, 1, true); P1_RD(DST, FBASE, BUF, UNIT, 2, true); P1_RD(DST, FBASE, BUF, UNIT, 3, true); \
                                           P1_RD(DST, FBASE, BUF, UNIT, 4, true); P1_RD(DST, FBASE, BUF, UNIT, 5, true); P1_RD(DST, FBASE, BUF, UNIT, 6, true); P1_RD(DST, FBASE, BUF, UNIT, 7, true); } while (0)
  P1_RD8(a0, fa, 0, 0);
  P1_RD8(bx, fb, 0, 1);

  for (;;) {
    // accumulators start at the bias: register e of block (mb, nb) of B half bh is column 64 bh + 16 nb + 4 g + e of every row
    {
      typedef __attribute__((ext_vector_type(16))) float f32x16c;
      typedef const __attribute__((address_space(4))) f32x16c* cvec_p;
      const int g = lane >> 4;
#pragma unroll
      for (int bh = 0; bh < 2; ++bh)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          f32x16c bvv;
          if (p.bias) bvv = *(cvec_p)(uintptr_t)(p.bias + n0 + wc * 128 + bh * 64 + 16 * nb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float b = 0.f;
            if (p.bias) {
              const float lo = (g & 1) ? bvv[4 + e] : bvv[e], hi = (g & 1) ? bvv[12 + e] : bvv[8 + e];
              b = (g & 2) ? hi : lo;
            }
#pragma unroll
            for (int ah = 0; ah < 2; ++ah)
#pragma unroll
              for (int mb = 0; mb < 4; ++mb) acc[bh][ah][mb][nb][e] = b;
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // v_accvgpr_write -> MFMA SrcC (no hardware interlock; the MFMAs are inline asm)
    __builtin_amdgcn_sched_barrier(0);
    nxt = item + item_step;
    more = nxt < item_end;
    {
      using F = std::false_type; using T = std::true_type;
      using Z = std::integral_constant<int, 0>; using O = std::integral_constant<int, 1>;
      int kt = 0;
      if (nt > 2) { body(Z{}, T{}, 0); body(O{}, T{}, 1); kt = 2; }
      for (; kt + 2 < nt; kt += 2) { body(Z{}, F{}, kt); body(O{}, F{}, kt + 1); }
      body(Z{}, T{}, kt);
      body(O{}, T{}, kt + 1);
    }
    // ---- epilogue: the two 128 x 64 halves of the wave's tile through the 16 x 16 x 32 kernel's epilogue (gemm_pp16.h); the next
    //      item's first units are in flight or landed, its first fragments are in a0 / bx
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA results -> the epilogue's accumulator reads
    __builtin_amdgcn_sched_barrier(0);
    const int em0 = m0 + wr * 128, en0 = n0 + wc * 128;
    item = nxt;
    if (more) { m0 = m0x; n0 = n0x; }
    if (!(p.atomic & 0x800)) {                         // (diagnostic 0x800: main loop only)
#pragma unroll
      for (int bh = 0; bh < 2; ++bh) {
        PpOut out;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
        pp16_epi_compute_q<EPI>(p, acc[bh], out, em0, en0 + 64 * bh, ln, tpa);
        asm volatile("" : "+v"(ln));
        pp_epi_store<EPI, 2>(p, out, em0, en0 + 64 * bh, ln);
      }
      est = NSTORE;
    } else {
      est = 0;
    }
    if (!more) break;
  }
#undef P1_DMA
#undef P1_MF
#undef P1_GROUP
#undef P1_PAIR
#undef P1_QUAD
#undef P1_RD
#undef P1_RD8
#undef P1_PHASE
#undef P1_LGKM0
#undef P1_BARRIER
}

static int p1_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI>
static void p1_launch0(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 131072 + 4 * 4096;            // the ring + one 4 KiB transposition buffer per wave
  auto kern = gemm_bf16_p1_kernel<EPI>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = (p.N / 256) * (p.M / 256);
  int cus = p1_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, p);
}

// what variant 13 takes (the caller has checked whole 256 x 256 tiles, the 32-bit staging offsets and the 16-byte alignments)
bool uc2_gemm_p1_supported(const GemmArgs& p, int trans_a, int trans_b) {
  if (trans_a || trans_b || p.c_f32 || p.split_k != 1 || p.queue) return false;
  if ((p.K % 128) != 0) return false;
  return p.epi == EPI_NONE || p.epi == EPI_ADD || (p.epi == EPI_DGELU && p.aux_deriv);
}

void uc2_gemm_p1_launch(const GemmArgs& p, hipStream_t st) {
  if (p.epi == EPI_ADD) p1_launch0<EPI_ADD>(p, st);
  else if (p.epi == EPI_DGELU) p1_launch0<EPI_MUL>(p, st);
  else p1_launch0<EPI_NONE>(p, st);
}
