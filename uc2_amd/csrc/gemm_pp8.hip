#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// The ping-pong kernel of gemm_pp16.hip on OCP fp8 (e4m3) operands: v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
// (BASELINE.json configs[4]: "fp8 MFMA GEMMs").  Both operands k-contiguous e4m3 bytes, handed over as "bf16" matrices of half the
// width (same bytes: a k-tile of 64 "bf16" = 128 e4m3 values = the same 128-byte LDS rows, units, swizzles, LDS-DMA pieces and
// counted vmcnt as the bf16 kernel).  A phase is 8 MFMAs of 32 cycles on the same 64 x 32 quadrant where the bf16 kernel runs 16 of
// 16: the same phase time for twice the contraction depth, i.e. half the main loop per flop; the epilogue is the bf16 kernel's plus
// the de-scaling multiply (alpha = 1 / (scale_a scale_b), a power of two: the accumulators start at bias / alpha, exactly).
// bf16 output, whole 256 x 256 tiles, an even number of k-tiles; epilogues none / GELU + gelu' / x gelu' (+ column sums) / + residual.
// ------------------------------------------------------------------------------------------------------
#include "gemm_pp8.h"

#ifndef UC2_PP8_DIAG
#define UC2_PP8_DIAG 0            /* 1: build the main-loop diagnostics 0x100 / 0x200 (tools/bench_pp.py) */
#endif

#ifndef PP16_AUX_TOUCH
#define PP16_AUX_TOUCH 1     // touch the lines of the epilogue's aux_in tile two k-tiles ahead (residual-add / gelu'-multiply / dGELU kinds)
#endif
template <bool TA, bool TB, bool TACC, int EPI, int HI, bool QOUT = false>
__global__ __launch_bounds__(512, 2) void gemm_fp8_pp8_kernel(GemmArgs p) {
  static_assert(HI == 2 && TACC && !TA && !TB, "256-row tiles, transposed accumulators, k-contiguous e4m3 operands");
  const float sab = p.alpha_dev ? p.alpha_dev[0] * p.alpha_dev2[0] : 1.0f / p.alpha;       // scale_a scale_b (wave-uniform)
  const float alpha = 1.0f / sab;
  // QOUT: an e4m3 copy of the main output for the GEMM that consumes it (delayed scaling: half the scale of the previous use's maximum)
  const float qs = QOUT ? fp8_delayed_scale(p.q_amax_prev) : 0.f;
  float qmax = 0.f;
  if (QOUT && blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { p.q_amax_clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *p.q_scale_out = qs; }
  constexpr int NQ = QOUT ? 8 : 0;                         // stores of the e4m3 stream per wave and item
  constexpr int RW = 64 + 32 * HI, RT = 2 * RW;           // rows per wave row / per tile (128 / 256, or 96 / 192)
  constexpr int GA1 = HI == 2 ? 2 : 1;                      // LDS-DMA instructions per wave for unit A1
  constexpr int GKT = 6 + GA1;                             // ... per k-tile
  constexpr int NST = (2 + HI) * 4;                        // bf16 stores per wave in the direct epilogue
  constexpr bool GRP = EPI == EPI_GROUP;                  // grouped launch: operands, shapes and the partial-tile destination are per item
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;

  // ---- work items: (tile, k-split), tile-major inside a split.  Workgroups land on XCD blockIdx % 8; each XCD
  //      owns a contiguous range of items (neighbouring tiles share an A row panel in that XCD's L2) and its
  //      workgroups stride through it together.
  const int nbx = p.N / 256, ntile = nbx * (p.M / RT);
  const int nitems = GRP ? p.grp_items : ntile * p.split_k;
  const int ktiles = p.K / 64, per = ((ktiles + p.split_k - 1) / p.split_k + 1) & ~1;     // k-tiles per split: even
  int item, item_end, item_step;
  {
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nx = min(G, 8);                                    // XCDs in use
    const int q = nitems / nx, r = nitems % nx;
    const int beg = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    item_end = beg + q + (xcd < r ? 1 : 0);
    item_step = (G - xcd + 7) >> 3;                              // workgroups on this XCD
    item = beg + slot;
  }
  // Item queue (GemmArgs::queue, optional): the first two items of a workgroup are the static ones (beg + slot and one
  // stride further); from the third on the index comes from a per-XCD counter.  A workgroup that is placed late -- another
  // kernel (an overlapped all-reduce) holds its CU -- then delays two items instead of its whole share.  The fetch for
  // the item after next is issued by one lane at the start of an item (an ordinary vector atomic: counted vmcnt waits
  // only get stricter by one operation while it is in flight) and consumed in that item's epilogue, where everything
  // older has landed anyway; wave 0 publishes it through the first word of its transposition buffer (free between two
  // epilogues) and every wave picks it up behind the next item barrier.  The last workgroup to leave zeroes the queue.
  int* const queue = p.queue;
  const int qslot = blockIdx.x & 7;
  const int dyn0 = item - (int)(blockIdx.x >> 3) + 2 * item_step;
  auto queue_leave = [&]() __attribute__((always_inline)) {
    if (queue && threadIdx.x == 0) {
      const int old = atomicAdd(queue + 8, 1);
      if (old == (int)gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicExch(queue + i, 0);
      }
    }
  };
  if (item >= item_end) { queue_leave(); return; }
  int nxt = item + item_step;                          // the item after the current one
  int pend = 0;                                        // (wave 0, lane 0) ticket of the item after that
  // De-phasing: with equal tiles every CU reaches its epilogue at the same moment and the 256 store bursts (plus the
  // next tiles' first fetches) queue on HBM while the matrix pipes idle.  Four phase groups (by slot within the XCD)
  // start p.skew * ~8k cycles apart, so at most a quarter of the chip stores at a time.
  if (p.skew > 0) {
    const int g = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < g * p.skew; ++i) __builtin_amdgcn_s_sleep(127);
  }

  unsigned stepa = (TA ? 64u * (unsigned)p.lda : 64u) * 2u;            // bytes per k-tile (grouped launch: set per item)
  unsigned stepb = (TB ? 64u * (unsigned)p.ldb : 64u) * 2u;
  float* pdst = nullptr; float* pdstx = nullptr;                       // grouped launch: partial tile of the item computed / staged
  int pld = 0, pldx = 0;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void_p)smem);
  unsigned fa[2][4], fb[2][2];                                   // fragment bases [k-tile buffer][block or k-step]
  fa[0][0] = lds0 + pp8_frag_off(wr * 64, lane);                 // units A0 / A1: 64 rows per wave row, four 16-row blocks
  fb[0][0] = lds0 + pp8_frag_off(wc * 32, lane);                 // units B0 / B1: 32 columns per wave column, two blocks
  // k-strided image: one base per block (the k-step is an immediate offset); k-contiguous image: one base per k-step (the block
  // is an immediate offset) -- fa[b] resp. fa[ks]
  // -- fa[buf][b] resp. fa[buf][ks]; the unit is an immediate offset too, so a fragment read needs no address arithmetic
#pragma unroll
  for (int i = 1; i < 4; ++i) fa[0][i] = (i == 1 ? (fa[0][0] ^ 16u) : fa[0][0]);      // [1]: the second 16-byte chunk of the lane's 32 bytes
  fb[0][1] = fb[0][0] ^ 16u;
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[1][i] = fa[0][i] + 65536u;
  fb[1][0] = fb[0][0] + 65536u; fb[1][1] = fb[0][1] + 65536u;

  int m0, n0, nt, zsplit;       // the item being computed
  int m0x, n0x, ntx, zx;        // the item being staged: the same one, until the tail of its main loop starts fetching the next
  unsigned src[4][2];           // staging sources as 32-bit byte offsets from A / B (SGPR base + VGPR offset addressing:
                                // half the registers of 64-bit pointers); unit type J (0 A0, 1 B0, 2 B1, 3 A1), wave-instructions w, w + 8
  auto setup = [&](int it) __attribute__((always_inline)) {
    if (GRP) {
      // which problem: items are numbered problem by problem (item0 ascending); everything below is wave-uniform
      int g = 0;
#pragma unroll
      for (int i = 1; i < UC2_GEMM_MAX_GROUP; ++i) if (i < p.ngroup && it >= p.grp[i].item0) g = i;
      const GemmProb& q = p.grp[g];
      const int lit = it - q.item0, ntl = q.nbx * q.mt;
      const int z = lit / ntl, tile = lit - z * ntl;
      zx = z;
      {
        const int cg = q.col_group, per_group = q.mt * cg;
        const int gg = tile / per_group, r = tile - gg * per_group;
        const int cw = min(cg, q.nbx - gg * cg);
        const int tm = r / cw, tc = r - tm * cw;
        m0x = tm * RT; n0x = (gg * cg + tc) * 256;
      }
      const int tbeg = z * q.per;
      ntx = min(q.ktiles, tbeg + q.per) - tbeg;
      const int kbeg = tbeg * 64;
      A = reinterpret_cast<const bf16*>(q.A); B = reinterpret_cast<const bf16*>(q.B);
      stepa = 128u * (unsigned)q.lda; stepb = 128u * (unsigned)q.ldb;
      pdstx = q.partial + (size_t)z * q.M * q.N; pldx = q.N;
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        src[0][qq] = (unsigned)((const char*)pp16_src<TA, 0, HI>(A, q.lda, q.M, m0x, kbeg, w + 8 * qq, ln) - (const char*)A);
        src[1][qq] = (unsigned)((const char*)pp16_src<TB, 1, HI>(B, q.ldb, q.N, n0x, kbeg, w + 8 * qq, ln) - (const char*)B);
        src[2][qq] = (unsigned)((const char*)pp16_src<TB, 2, HI>(B, q.ldb, q.N, n0x, kbeg, w + 8 * qq, ln) - (const char*)B);
        src[3][qq] = (unsigned)((const char*)pp16_src<TA, 3, HI>(A, q.lda, q.M, m0x, kbeg, w + 8 * qq, ln) - (const char*)A);
      }
      return;
    }
    const int z = it / ntile, tile = it - z * ntile;
    zx = z;
    {
      // Tile order inside a k-split: column tiles in groups of `cg` (p.diag-selectable; default chosen on the host so that
      // cg <= 6), row panels inside a group, the group's columns fastest.  The 32 workgroups of an XCD work on 32
      // consecutive items: with all 12 column tiles of an N = 3072 GEMM in one row they covered 2.7 row panels x 12
      // weight tiles = 5.8 MB of operands, more than the XCD's 4 MiB L2 -- rocprofv3 FETCH_SIZE showed the 4.7 MB
      // weight re-fetched from beyond L2 for every row panel (2.4 GB per launch against 1.4 GB algorithmic).
      const int cg = p.col_group, mt = p.M / RT;
      const int per_group = mt * cg;
      const int g = tile / per_group, r = tile - g * per_group;
      const int cw = min(cg, nbx - g * cg);                  // (the last group may be narrower)
      const int tm = (g * cg + cw <= nbx && cw == cg) ? r / cg : r / cw;
      const int tc = r - tm * ((cw == cg) ? cg : cw);
      m0x = tm * RT; n0x = (g * cg + tc) * 256;
    }
    const int tbeg = z * per;
    ntx = min(ktiles, tbeg + per) - tbeg;
    const int kbeg = tbeg * 64;
    int ln = lane;                                     // opaque copy: keeps the per-lane address arithmetic from being
    asm volatile("" : "+v"(ln));                       // hoisted out of the item loop (it would live, and spill, across the main loop)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      src[0][q] = (unsigned)((const char*)pp16_src<TA, 0, HI>(A, p.lda, p.M, m0x, kbeg, w + 8 * q, ln) - (const char*)A);
      src[1][q] = (unsigned)((const char*)pp16_src<TB, 1, HI>(B, p.ldb, p.N, n0x, kbeg, w + 8 * q, ln) - (const char*)B);
      src[2][q] = (unsigned)((const char*)pp16_src<TB, 2, HI>(B, p.ldb, p.N, n0x, kbeg, w + 8 * q, ln) - (const char*)B);
      src[3][q] = (unsigned)((const char*)pp16_src<TA, 3, HI>(A, p.lda, p.M, m0x, kbeg, w + 8 * q, ln) - (const char*)A);
    }
  };

  // issue unit type J of the next k-tile that type has not fetched yet into buffer BUF (0/1)
#define PP_ISSUE(J, BUF)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < ((J) == 3 ? GA1 : 2); ++q) {                                         \
      __builtin_amdgcn_global_load_lds((glb_void_p)((const char*)(((J) == 0 || (J) == 3) ? (const void*)A : (const void*)B) + src[J][q]),                                                 \
                                       (lds_void_p)(smem + (BUF) * 65536 + (J) * PP_UNIT + (w + 8 * q) * 1024), 16, 0, 0); \
      src[J][q] += ((J) == 0 || (J) == 3) ? stepa : stepb;   /* bytes */                                                     \
    }                                                                                                          \
  } while (0)
  // first six units of an item in stream order (B0, A0, B1, A1 of k-tile 0, B0 and A0 of k-tile 1); the host guarantees nt >= 2
#define PP_PROLOGUE() do { PP_ISSUE(1, 0); PP_ISSUE(0, 0); PP_ISSUE(2, 0); PP_ISSUE(3, 0); PP_ISSUE(1, 1); PP_ISSUE(0, 1); } while (0)

  f32x4 acc[2][4][4];                                  // [A half][16-row block][16-column block: 2 * (B half) + nbl]
  bf16x8 a[4][2], bx[4], by[4];                        // a[block][k-step]; B sets [2 * nbl + ks], they swap roles (B0 / B1) every k-tile

  // inline asm with the accumulator tied in place: as a builtin hipcc gave the MFMA a destination other than its C operand,
  // rotated accumulator blocks through other registers and spilled 170 of them (8-register operand tuples leave it less room than the
  // bf16 kernel's).  What the compiler can no longer see -- vector writes before and reads after an MFMA -- gets s_nops (init_acc, epilogue).
  int usc = 0x7F7F7F7F;                                // unit block scales (E8M0 127) for both operands
  asm volatile("" : "+v"(usc));
#define PP_MFMA(H, JB, BREG)                                                                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                           \
      _Pragma("unroll") for (int nbl = 0; nbl < 2; ++nbl)                                                      \
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"              \
                     : "+v"(acc[H][mb][2 * (JB) + nbl])                                                        \
                     : "v"(pp8_cat(BREG[2 * nbl], BREG[2 * nbl + 1])), "v"(pp8_cat(a[mb][0], a[mb][1])), "v"(usc)); \
  } while (0)
#define PP16_FA(BUF, B_, KS_) (TA ? fa[BUF][B_] : fa[BUF][KS_])
#define PP16_FB(BUF, B_, KS_) (TB ? fb[BUF][B_] : fb[BUF][KS_])
#define PP_READ_A(BUF, UNIT)                                                                                   \
  do {                                                                                                         \
    pp16_read<TA, 0, 0, (UNIT) * PP_UNIT>(a[0][0], PP16_FA(BUF, 0, 0)); pp16_read<TA, 0, 1, (UNIT) * PP_UNIT>(a[0][1], PP16_FA(BUF, 0, 1)); \
    pp16_read<TA, 1, 0, (UNIT) * PP_UNIT>(a[1][0], PP16_FA(BUF, 1, 0)); pp16_read<TA, 1, 1, (UNIT) * PP_UNIT>(a[1][1], PP16_FA(BUF, 1, 1)); \
    pp16_read<TA, 2, 0, (UNIT) * PP_UNIT>(a[2][0], PP16_FA(BUF, 2, 0)); pp16_read<TA, 2, 1, (UNIT) * PP_UNIT>(a[2][1], PP16_FA(BUF, 2, 1)); \
    pp16_read<TA, 3, 0, (UNIT) * PP_UNIT>(a[3][0], PP16_FA(BUF, 3, 0)); pp16_read<TA, 3, 1, (UNIT) * PP_UNIT>(a[3][1], PP16_FA(BUF, 3, 1)); \
  } while (0)
#define PP_READ_B(BREG, BUF, UNIT)                                                                             \
  do {                                                                                                         \
    pp16_read<TB, 0, 0, (UNIT) * PP_UNIT>(BREG[0], PP16_FB(BUF, 0, 0)); pp16_read<TB, 0, 1, (UNIT) * PP_UNIT>(BREG[1], PP16_FB(BUF, 0, 1)); \
    pp16_read<TB, 1, 0, (UNIT) * PP_UNIT>(BREG[2], PP16_FB(BUF, 1, 0)); pp16_read<TB, 1, 1, (UNIT) * PP_UNIT>(BREG[3], PP16_FB(BUF, 1, 1)); \
  } while (0)
  // end of an L section: retire the units the next phase reads, publish, then wait for this phase's own reads
  // ALLOW = units (the oldest of the window f+3 .. f+6) that may stay in flight; P = phase: with HI = 1 the A1 unit
  // (one DMA instead of two) is the (4-P)&3-th of them
#define PP_SYNC_L(ALLOW, P)                                                                                    \
  do {                                                                                                         \
    const int al_ = (ALLOW) > 4 ? 4 : ((ALLOW) < 0 ? 0 : (ALLOW));                                             \
    pp_wait_small(2 * al_ - ((GA1 == 1 && al_ > ((4 - (P)) & 3)) ? 1 : 0));                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
  } while (0)
#define PP_SYNC_C()                                                                                            \
  do {                                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  // main-loop diagnostics (with 0x800): 0x100 = no LDS-DMA issue (stale LDS contents), 0x200 = no fragment reads after the first k-tile
  const bool dg_nodma = UC2_PP8_DIAG && (p.atomic & 0x100) != 0, dg_nord = UC2_PP8_DIAG && (p.atomic & 0x200) != 0;     // (make EXTRA=-DUC2_PP8_DIAG=1)
  bool more = false;                                   // another item follows the current one
  constexpr bool AUXK = TACC && (EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_DGELU || EPI == EPI_DROPADD);     // epilogues that read an aux_in tile
  unsigned aux_t0 = 0, aux_t1 = 0;                     // (PP16_AUX_TOUCH) destinations of the line-touching loads
  const bool aux_touch = AUXK && (size_t)p.M * (size_t)p.N <= ((size_t)32 << 20);
  auto body = [&](auto tail_c, auto swap_c, int kt) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool SW = decltype(swap_c)::value;       // k-tile parity: B0 lives in by, B1 in bx
    bf16x8 (&b0)[4] = SW ? by : bx;
    bf16x8 (&b1)[4] = SW ? bx : by;
    const int nunits = 4 * nt;
    constexpr int CB = SW ? 1 : 0;                     // this k-tile's buffer (the k-tile parity is the template argument)
    const int nb = (kt & 1) ^ 1;                       // buffer of k-tile kt+1 (kt+2 shares this tile's)
    const int f0 = 4 * kt;                             // first phase; phase f issues unit f+6, may leave min(4, nunits-3-f) units in flight
    // Fragment reads are spread 8 / 4 / 8 / 4 over the phases: B0 of k-tile kt+1 is read in phase 3 of k-tile kt, into
    // the registers of B1 (dead after phase 2; the current B0 is still needed by this phase's MFMAs) -- the two B
    // register sets swap roles every k-tile.  The unit order of the stream is therefore B0, A0, B1, A1: every phase
    // consumes exactly the unit the previous phase's wait retired.
    // ---- phase 0
    if (!dg_nord || kt == 0) PP_READ_A(CB, 0);
    if (PP16_AUX_TOUCH && AUXK && TAIL && !SW && aux_touch) {
      // Two k-tiles before the epilogue needs them: one 4-byte load per 128-byte line of this wave's aux_in tile (128 rows x 64
      // columns: lane l touches rows l and 64 + l), so that the epilogue's own loads -- 16 per lane, needed at once, with the
      // matrix pipe idle -- find their lines in L2 instead of paying the HBM latency (~5 k cycles per tile, round 2's stamps).
      // The two destination registers are pinned until the epilogue has consumed its aux values (loads return in order).
      // Only while the aux tensor is small (aux_touch: <= 64 MB, the reference's micro-batches: residual-add input gradients
      // -3..4 % at 9 984 tokens): at 196 608 tokens and more the touched lines push the operand panels out of L2 (same box:
      // gelu'-multiply 885 -> 948 us, residual add 684 -> 702 us).
      const char* ap = reinterpret_cast<const char*>(p.aux_in) + ((size_t)(m0 + wr * RW + lane) * p.ldaux + n0 + wc * 64) * 2;
      const char* ap2 = ap + (size_t)64 * p.ldaux * 2;
      asm volatile("global_load_dword %0, %1, off" : "=v"(aux_t0) : "v"(ap) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(aux_t1) : "v"(ap2) : "memory");
    }
    if ((!TAIL || f0 + 6 < nunits || more) && !dg_nodma) PP_ISSUE(2, nb);
    PP_SYNC_L((TAIL && !more) ? nunits - 3 - f0 : 4, 0);
    PP_MFMA(0, 0, b0);
    PP_SYNC_C();
    // ---- phase 1
    if (!dg_nord || kt == 0) PP_READ_B(b1, CB, 2);
    if ((!TAIL || f0 + 7 < nunits || more) && !dg_nodma) PP_ISSUE(3, nb);
    PP_SYNC_L((TAIL && !more) ? nunits - 4 - f0 : 4, 1);
    PP_MFMA(0, 1, b1);
    PP_SYNC_C();
    // ---- phase 2
    // The stream does not drain at the end of an item: the last six phases (from here on in the first of the two tail
    // k-tiles; every unit of the current item has been issued) fetch the NEXT item's first six units, in the order and
    // into the ring positions a prologue would use (nt is even).  The next item then starts with its operands in LDS
    // instead of issuing 96 KiB of LDS-DMA and waiting for it with the matrix pipe idle.
    if (TAIL && !SW && more) setup(nxt);
    if (!dg_nord || kt == 0) PP_READ_A(CB, 3);
    if ((!TAIL || f0 + 8 < nunits || more) && !dg_nodma) PP_ISSUE(1, nb ^ 1);
    PP_SYNC_L((TAIL && !more) ? nunits - 5 - f0 : 4, 2);
    PP_MFMA(1, 1, b1);
    PP_SYNC_C();
    // ---- phase 3
    if ((!TAIL || kt + 1 < nt) && (!dg_nord || kt == 0)) PP_READ_B(b1, CB ^ 1, 1);
    if ((!TAIL || f0 + 9 < nunits || more) && !dg_nodma) PP_ISSUE(0, nb ^ 1);
    PP_SYNC_L((TAIL && !more) ? nunits - 6 - f0 : 4, 3);
    PP_MFMA(1, 0, b0);
    if (TAIL && kt == nt - 1) {                        // wave row 1 has no partner barrier left after its last C section
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (wr == 0) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      PP_SYNC_C();
    }
  };

  // diagnostic time stamps (p.atomic & 0x10000, p.aux_out = uint32 [8 waves][16]): workgroup 0, around its third item
  const bool dbg = (p.atomic & 0x10000) && blockIdx.x == 0;
  unsigned ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int nitem_done = 0;
#define PP_STAMP(I) do { if (dbg && nitem_done == 2) { unsigned long long t64_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64_) :: "memory"); ts[I] = (unsigned)t64_; } } while (0)
  setup(item);
  m0 = m0x; n0 = n0x; nt = ntx; zsplit = zx;
  if (GRP) { pdst = pdstx; pld = pldx; }
  PP_PROLOGUE();
  int younger = GKT;                                   // VMEM operations issued after the first two units of the current item
  for (;;) {
    // Accumulators start at the bias (scalar loads: uniform address in constant space, lgkmcnt, no vector registers).
    // Register 8g+4cc+e of block j is column 32j+16g+8cc+4h+e.  128 VALU writes per wave: wave row 0 does them BEFORE the
    // item barrier (it finishes its epilogue about a thousand cycles ahead of wave row 1 and would only wait there), wave
    // row 1 after its second barrier, beside wave row 0's first C section.
    auto init_acc = [&]() __attribute__((always_inline)) {
      if (p.bias) {
        // register e of block (mb, nb) is column 16 nb + 4 g + e (g = lane >> 4) of every row: 16 scalar-loaded values per
        // block column, selected by g
        typedef __attribute__((ext_vector_type(16))) float f32x16c;
        typedef const __attribute__((address_space(4))) f32x16c* cvec_p;
        const int g = lane >> 4;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const f32x16c bvv = *(cvec_p)(uintptr_t)(p.bias + n0 + wc * 64 + 16 * nb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = (g & 1) ? bvv[4 + e] : bvv[e], hi = (g & 1) ? bvv[12 + e] : bvv[8 + e];
            const float b = ((g & 2) ? hi : lo) * sab;         // (acc + bias / alpha) alpha = acc alpha + bias
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int mb = 0; mb < 4; ++mb) acc[hh][mb][nb][e] = b;
          }
        }
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[hh][mb][nb][e] = 0.f;
      }
    };
    // units 0 and 1 of this item: this wave's part has landed, then publish
    auto item_barrier = [&]() __attribute__((always_inline)) {
      if (younger == GKT) { if (HI == 2) wait_vmcnt<8>(); else wait_vmcnt<7>(); }
      else if (younger == GKT + NST) { if (HI == 2) wait_vmcnt<24>(); else wait_vmcnt<19>(); }
      else if (younger == GKT + 2 * NST) { if (HI == 2) wait_vmcnt<40>(); else wait_vmcnt<31>(); }
      else if (QOUT && younger == GKT + NST + 8) wait_vmcnt<32>();
      else if (QOUT && younger == GKT + 2 * NST + 8) wait_vmcnt<48>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    };
    if (wr == 1) {                                     // wave row 1 runs one barrier interval behind
      item_barrier();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    init_acc();
    asm volatile("s_nop 7" ::: "memory");              // vector writes -> MFMA C operand (inline-asm MFMAs: no compiler-inserted wait states)
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) item_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(0);
    if (dbg && nitem_done == 3) { unsigned long long t64_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64_) :: "memory"); ts[6] = (unsigned)t64_; }
    if (queue) {
      if (nitem_done > 0) nxt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem + 131072));
      if (w == 0 && lane == 0) pend = __hip_atomic_fetch_add(queue + qslot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      nxt = item + item_step;
    }
    more = nxt < item_end;
    if (!(p.atomic & 0x4000)) {                        // (diagnostic 0x4000: epilogue only)
      using F = std::false_type; using T = std::true_type;
      PP_READ_B(bx, 0, 1);                             // B0 of k-tile 0 (later k-tiles get theirs one phase ahead)
      int kt = 0;
      for (; kt + 2 < nt; kt += 2) { body(F{}, F{}, kt); body(F{}, T{}, kt + 1); }     // nt is even (host-checked)
      body(T{}, F{}, kt);
      body(T{}, T{}, kt + 1);
    } else if (more) {
      setup(nxt);
      PP_PROLOGUE();
    }
    // The next item's first six units are in flight or landed (issued by the tail above); the epilogue below touches
    // only the transposition buffers behind the ring.
    PP_STAMP(1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results -> the epilogue's vector reads
    __builtin_amdgcn_sched_barrier(0);
    const int em0 = m0 + wr * RW, en0 = n0 + wc * 64, ez = zsplit;
    float* const edst = pdst; const int eld = pld;
    item = nxt;
    if (more) { m0 = m0x; n0 = n0x; nt = ntx; zsplit = zx; if (GRP) { pdst = pdstx; pld = pldx; } }
    // (queue) the ticket fetched at the start of this item -> index of the item after the next one; the use makes hipcc wait
    // for it here, after the epilogue arithmetic and before the stores; wave 0 writes it once its transposition buffer is free
    int ticket_item = 0;
    auto ticket_ready = [&]() __attribute__((always_inline)) {
      if (queue && w == 0 && lane == 0) { ticket_item = dyn0 + pend; asm volatile("" : "+v"(ticket_item)); }
    };
    auto ticket_publish = [&]() __attribute__((always_inline)) {
      if (queue && w == 0 && lane == 0) {
        *reinterpret_cast<volatile int*>(smem + 131072) = ticket_item;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    };
    const bool store = !(p.atomic & 0x800);            // (diagnostic 0x800: main loop only)
    if (!store) {
      ticket_ready(); ticket_publish();
      younger = GKT;
    } else if (TACC && (EPI == EPI_ACC || GRP || p.partial)) { // split-K item of a two-stage reduction (fp32 partial, plain stores),
      {                                                 // or C += tile for an unsplit fp32 weight gradient (EPI_ACC)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
        ticket_ready();
        if (EPI == EPI_ACC) pp16_partial_store<true>(reinterpret_cast<float*>(p.C), p.ldc, acc, em0, en0, ln, tpa);
        else if (GRP) pp16_partial_store<false>(edst, eld, acc, em0, en0, ln, tpa);
        else pp16_partial_store<false>(p.partial + (size_t)ez * p.M * p.N, p.N, acc, em0, en0, ln, tpa);
      }
      ticket_publish();
      younger = GKT + 2 * NST;
    } else if (TACC) {
      PpOut out;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      {
        const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
        pp8_epi_compute_q<EPI, QOUT>(p, acc, out, em0, en0, ln, tpa, alpha, qs, qmax);
      }
      if (PP16_AUX_TOUCH && AUXK) asm volatile("" :: "v"(aux_t0), "v"(aux_t1));      // (the touch loads are older than the aux loads just consumed)
      // pin the finished outputs here: hipcc must not sink the aux-dependent arithmetic into the store sequence below
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (!(hh == 1 && i >= HI)) {
              asm volatile("" : "+v"(out.o[hh][i][it]));
              if (!QOUT && (EPI == EPI_GELU || EPI == EPI_GELU_D)) asm volatile("" : "+v"(out.pre[hh][i][it]));
            }
      PP_STAMP(2);
      ticket_ready(); ticket_publish();
      PP_STAMP(3);
      asm volatile("" : "+v"(ln));
      if (QOUT) pp_epi_store<EPI_NONE, HI>(p, out, em0, en0, ln);       // (the second stream and the e4m3 stream left block by block)
      else pp_epi_store<EPI, HI>(p, out, em0, en0, ln);
      PP_STAMP(4);
      younger = GKT + ((EPI == EPI_GELU || EPI == EPI_GELU_D) ? 2 * NST : NST) + NQ;     // stores issued after the DMA
    }
    ++nitem_done;
    if (!more) break;
  }
  queue_leave();
  if (QOUT) {                                          // one atomic per WORKGROUP and launch (2048 per-wave atomics on one address cost ~25 us)
    const float m = wave_max(qmax);
    float* red = reinterpret_cast<float*>(smem + 131072);          // (the transposition buffers are free now)
    __syncthreads();
    if (lane == 0) red[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = red[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) mm = fmaxf(mm, red[i]);
      amax_cell_raise(p.q_amax_next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), mm);
    }
  }
  if (dbg) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
      unsigned* o = reinterpret_cast<unsigned*>(p.aux_out) + w * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
#undef PP_STAMP
#undef PP_ISSUE
#undef PP_PROLOGUE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B
#undef PP_SYNC_L
#undef PP_SYNC_C
}

static int pp8_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI, bool QOUT = false>
static void pp8_launch0(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 131072 + 8 * 4096;
  auto kern = gemm_fp8_pp8_kernel<false, false, true, EPI, 2, QOUT>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = (p.N / 256) * (p.M / 256);
  int cus = pp8_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, p);
}

// p is the "bf16 view" of the e4m3 problem (K, lda, ldb halved; gemm_fast.hip uc2_gemm_fp8_launch): whole 256 x 256 tiles, an even
// number (>= 2) of 128-byte k-tiles, 32-bit staging offsets, 16-byte aligned bf16 output / aux tensors
bool uc2_gemm_pp8_supported(const GemmArgs& p) {
  const int ktiles = p.K / 64;
  if ((p.M % 256) || (p.N & 255) || (p.K % 64) || (ktiles & 1) || ktiles < 2) return false;
  if (2ull * p.M * p.lda >= (1ull << 32) || 2ull * p.N * p.ldb >= (1ull << 32)) return false;
  if (p.c_f32 || p.accumulate || (p.ldc & 7) || (p.ldaux & 7) || ((uintptr_t)p.C & 15) || ((uintptr_t)p.aux_in & 15) ||
      ((uintptr_t)p.aux_out & 15) || ((uintptr_t)p.bias & 15)) return false;
  if (p.epi == EPI_GELU) return p.aux_deriv && p.aux_out != nullptr;
  if (p.epi == EPI_DGELU) return p.aux_deriv != 0;
  return p.epi == EPI_NONE || p.epi == EPI_ADD || p.epi == EPI_DROPADD;
}

void uc2_gemm_pp8_launch(const GemmArgs& p0, hipStream_t st) {
  GemmArgs p = p0;
  p.split_k = 1; p.partial = nullptr; p.queue = nullptr; p.skew = 0;
  {
    const int nbx = p.N / 256;
    int cg = nbx;
    if (nbx > 6) { cg = 1; for (int d = 6; d >= 2; --d) if (nbx % d == 0) { cg = d; break; } if (cg == 1) cg = 6; }
    p.col_group = cg;
  }
  if (p.q_out) {                                       // (uc2_gemm_fp8_q has checked: GELU + gelu' or x gelu' kinds only)
    if (p.epi == EPI_GELU) pp8_launch0<EPI_GELU_D, true>(p, st);
    else pp8_launch0<EPI_MUL, true>(p, st);
    return;
  }
  if (p.epi == EPI_GELU) pp8_launch0<EPI_GELU_D>(p, st);
  else if (p.epi == EPI_DGELU) pp8_launch0<EPI_MUL>(p, st);
  else if (p.epi == EPI_ADD) pp8_launch0<EPI_ADD>(p, st);
  else if (p.epi == EPI_DROPADD) pp8_launch0<EPI_DROPADD>(p, st);
  else pp8_launch0<EPI_NONE>(p, st);
}
