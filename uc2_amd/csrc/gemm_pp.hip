#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// Ping-pong variant: tile 256x256x64, 8 waves = 2 (M) x 4 (N), 128x64 of output per wave (128 accumulator
// registers), one workgroup per CU, 128 KiB of LDS = 2 k-tile buffers x 4 staging units of 16 KiB.
//
// A k-tile is consumed in four phases, one 64x32 quadrant of the wave's output each (8 MFMAs); the fragment
// reads are spread 8 / 4 / 8 / 4 (ds_read_b128 equivalents) over them:
//     phase 0: read A0 (rows 0-63 of the wave's 128)                         -> A0 x B0   (B0 was read one phase ago)
//     phase 1: read B1 (columns 32-63 of the wave's 64)                      -> A0 x B1
//     phase 2: read A1 (overwrites A0)                                       -> A1 x B1
//     phase 3: read B0 of the NEXT k-tile into the registers of B1 (dead)    -> A1 x B0
// (the two B register sets swap roles every k-tile; the main loop is unrolled by two, k-tile counts are even).
// Every phase is  [L: fragment reads + 2 LDS-DMA issues + counted vmcnt] barrier [C: 8 MFMAs] barrier.  The two
// wave rows run one barrier apart (wave row 1 takes one extra barrier at the start), so in every barrier
// interval one wave of each SIMD is in its C section while the other is in its L section: the matrix pipe
// always has a wave feeding it and the other wave's LDS reads / DMA issues cost it nothing.
//
// Staging: the k-tile is cut into four units of 128 rows x 64 k, streamed in the order the phases consume them:
// B0 (columns 0-31 of all four wave columns), A0 (rows 0-63 of both wave rows), B1, A1 -- one unit per phase.
// Phase f issues unit f+6 (each wave two 1-KiB LDS-DMA instructions), so a unit is issued 5-6 phases before
// its first read and at least two barrier intervals after the last read of the unit it overwrites;
// `s_waitcnt vmcnt(8)` at the end of every L section retires exactly the unit the next phase reads (all but
// the 4 youngest units), and the barrier that follows publishes it to the other waves.
//
// HI = 1 variant: tiles of 192 rows (96 per wave row: A0 = 64 rows, A1 = 32).  The N = 768 GEMMs of the encoder have
// 576 tiles of 256 rows on 256 CUs (2.25 rounds, the last a quarter full); with 192 rows they are 768 = three full
// rounds.  Unit A1 is then 8 KiB (one LDS-DMA per wave, 7 per k-tile instead of 8) and phases 2, 3 run 4 MFMAs.
// HI = 0 variant (round 6): tiles of 128 rows (64 per wave row: A0 only, no unit A1, 6 LDS-DMA per k-tile; phases 2 and 3 keep
// their barriers and issue slots but run no MFMA).  At the reference's 104-pair micro-batch (9 984 tokens) the N = 768 GEMMs
// are 117 tiles of 256 rows = 46 % of the 256 CUs, 156 of 192 rows, 234 of 128 rows = 91 %: half the matrix work per k-tile
// for the same B traffic, on twice the CUs (priced in profiles/r05_experiments.md section 3a, measured in r06_experiments.md).
//
// The kernel is persistent: one workgroup per CU walks over (output tile, k-split) work items, and the LDS-DMA stream
// does not drain between them: the last six phases of an item's main loop (which have no unit of their own item left to
// issue) fetch the NEXT item's first six units, in the order and ring positions a prologue would use (k-tile counts are
// even), so the next main loop starts with its operands in LDS.  The epilogue never touches the ring (it transposes
// through 4 KiB of wave-private LDS beside it, see tp_* below); its stores are issued after those units
// (`vmcnt(8 + stores)` at the item barrier then retires exactly the two units the first phase reads) and are
// non-temporal: outputs stream past the L2 instead of evicting the weight tiles every row panel re-reads.
// At K = 768 the per-tile launch + first-fetch latency and the store tail were 40 % of a non-persistent tile.
// ------------------------------------------------------------------------------------------------------
#include "gemm_pp.h"

#ifndef UC2_PP_DIAG
#define UC2_PP_DIAG 0            /* 1: build the main-loop diagnostics 0x100 / 0x200 (tools/bench_pp.py) */
#endif

template <bool TA, bool TB, bool TACC, int EPI, int HI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(GemmArgs p) {
  constexpr int RW = 64 + 32 * HI, RT = 2 * RW;           // rows per wave row / per tile (128 / 256, or 96 / 192)
  constexpr int GA1 = HI;                                   // LDS-DMA instructions per wave for unit A1 (2 / 1 / none)
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
  unsigned fa[2], fa1[2], fb;                                    // fragment addresses (buffer 0; the other is + 65536)
  fa[0] = lds0 + pp_frag_off<TA>(wr * 64, lane);                 // unit A0: 64 rows per wave row
  fa[1] = lds0 + pp_frag_off<TA>(wr * 64 + 32, lane);
  fa1[0] = lds0 + pp_frag_off<TA>(wr * 32 * HI, lane);           // unit A1: 32*HI rows per wave row
  fa1[1] = lds0 + pp_frag_off<TA>(wr * 32 * HI + 32, lane);
  fb = lds0 + pp_frag_off<TB>(wc * 32, lane);

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
        src[0][qq] = (unsigned)((const char*)pp_src<TA, 0, HI>(A, q.lda, q.M, m0x, kbeg, w + 8 * qq, ln) - (const char*)A);
        src[1][qq] = (unsigned)((const char*)pp_src<TB, 1, HI>(B, q.ldb, q.N, n0x, kbeg, w + 8 * qq, ln) - (const char*)B);
        src[2][qq] = (unsigned)((const char*)pp_src<TB, 2, HI>(B, q.ldb, q.N, n0x, kbeg, w + 8 * qq, ln) - (const char*)B);
        src[3][qq] = (unsigned)((const char*)pp_src<TA, 3, HI>(A, q.lda, q.M, m0x, kbeg, w + 8 * qq, ln) - (const char*)A);
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
      src[0][q] = (unsigned)((const char*)pp_src<TA, 0, HI>(A, p.lda, p.M, m0x, kbeg, w + 8 * q, ln) - (const char*)A);
      src[1][q] = (unsigned)((const char*)pp_src<TB, 1, HI>(B, p.ldb, p.N, n0x, kbeg, w + 8 * q, ln) - (const char*)B);
      src[2][q] = (unsigned)((const char*)pp_src<TB, 2, HI>(B, p.ldb, p.N, n0x, kbeg, w + 8 * q, ln) - (const char*)B);
      src[3][q] = (unsigned)((const char*)pp_src<TA, 3, HI>(A, p.lda, p.M, m0x, kbeg, w + 8 * q, ln) - (const char*)A);
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

  f32x16 acc[2][2][2];                                 // [A half][i][j]
  bf16x8 a[2][4], bx[4], by[4];                        // the two B register sets swap roles (B0 / B1) every k-tile

#define PP_MFMA(H, JB, BREG)                                                                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
      _Pragma("unroll") for (int i = 0; i < ((H) == 1 ? HI : 2); ++i) {                                        \
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
#define PP_READ_A1(BASEOFF)                                                                                    \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < HI; ++i) {                                                           \
      pp_read<TA, 0>(a[i][0], fa1[i] + (BASEOFF)); pp_read<TA, 1>(a[i][1], fa1[i] + (BASEOFF));                \
      pp_read<TA, 2>(a[i][2], fa1[i] + (BASEOFF)); pp_read<TA, 3>(a[i][3], fa1[i] + (BASEOFF));                \
    }                                                                                                          \
  } while (0)
#define PP_READ_B(BREG, BASEOFF)                                                                               \
  do {                                                                                                         \
    pp_read<TB, 0>(BREG[0], fb + (BASEOFF)); pp_read<TB, 1>(BREG[1], fb + (BASEOFF));                          \
    pp_read<TB, 2>(BREG[2], fb + (BASEOFF)); pp_read<TB, 3>(BREG[3], fb + (BASEOFF));                          \
  } while (0)
  // end of an L section: retire the units the next phase reads, publish, then wait for this phase's own reads
  // ALLOW = units (the oldest of the window f+3 .. f+6) that may stay in flight; P = phase: with HI = 1 the A1 unit
  // (one DMA instead of two) is the (4-P)&3-th of them
#define PP_SYNC_L(ALLOW, P)                                                                                    \
  do {                                                                                                         \
    const int al_ = (ALLOW) > 4 ? 4 : ((ALLOW) < 0 ? 0 : (ALLOW));                                             \
    pp_wait_small(2 * al_ - ((al_ > ((4 - (P)) & 3)) ? 2 - GA1 : 0));                                          \
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
  const bool dg_nodma = UC2_PP_DIAG && (p.atomic & 0x100) != 0, dg_nord = UC2_PP_DIAG && (p.atomic & 0x200) != 0;     // (make EXTRA=-DUC2_PP_DIAG=1)
  bool more = false;                                   // another item follows the current one
  auto body = [&](auto tail_c, auto swap_c, int kt) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool SW = decltype(swap_c)::value;       // k-tile parity: B0 lives in by, B1 in bx
    bf16x8 (&b0)[4] = SW ? by : bx;
    bf16x8 (&b1)[4] = SW ? bx : by;
    const int nunits = 4 * nt;
    unsigned cb = (kt & 1) * 65536u;                   // this k-tile's buffer
    asm volatile("" : "+s"(cb));                       // opaque: the fragment addresses (base ^ step, + buffer, + unit) are recomputed per
                                                       // read (1-2 VALU) instead of living in ~40 loop-invariant registers
    const int nb = (kt & 1) ^ 1;                       // buffer of k-tile kt+1 (kt+2 shares this tile's)
    const int f0 = 4 * kt;                             // first phase; phase f issues unit f+6, may leave min(4, nunits-3-f) units in flight
    // Fragment reads are spread 8 / 4 / 8 / 4 over the phases: B0 of k-tile kt+1 is read in phase 3 of k-tile kt, into
    // the registers of B1 (dead after phase 2; the current B0 is still needed by this phase's MFMAs) -- the two B
    // register sets swap roles every k-tile.  The unit order of the stream is therefore B0, A0, B1, A1: every phase
    // consumes exactly the unit the previous phase's wait retired.
    // ---- phase 0
    if (!dg_nord || kt == 0) PP_READ_A(cb + 0 * PP_UNIT);
    if ((!TAIL || f0 + 6 < nunits || more) && !dg_nodma) PP_ISSUE(2, nb);
    PP_SYNC_L((TAIL && !more) ? nunits - 3 - f0 : 4, 0);
    PP_MFMA(0, 0, b0);
    PP_SYNC_C();
    // ---- phase 1
    if (!dg_nord || kt == 0) PP_READ_B(b1, cb + 2 * PP_UNIT);
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
    if (!dg_nord || kt == 0) PP_READ_A1(cb + 3 * PP_UNIT);
    if ((!TAIL || f0 + 8 < nunits || more) && !dg_nodma) PP_ISSUE(1, nb ^ 1);
    PP_SYNC_L((TAIL && !more) ? nunits - 5 - f0 : 4, 2);
    PP_MFMA(1, 1, b1);
    PP_SYNC_C();
    // ---- phase 3
    if ((!TAIL || kt + 1 < nt) && (!dg_nord || kt == 0)) PP_READ_B(b1, (cb ^ 65536u) + 1 * PP_UNIT);
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
      if (TACC && p.bias) {
        typedef __attribute__((ext_vector_type(16))) float f32x16c;
        typedef const __attribute__((address_space(4))) f32x16c* cvec_p;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const f32x16c bvv = *(cvec_p)(uintptr_t)(p.bias + n0 + wc * 64 + 32 * j + 16 * g);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float lo = bvv[8 * cc + e], hi = bvv[8 * cc + 4 + e];
                const float b = (lane >> 5) ? hi : lo;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                  for (int i = 0; i < 2; ++i) acc[hh][i][j][8 * g + 4 * cc + e] = b;
              }
          }
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[hh][i][j][r] = 0.f;
      }
    };
    // units 0 and 1 of this item: this wave's part has landed, then publish
    auto item_barrier = [&]() __attribute__((always_inline)) {
      // (the six units of a prologue are 10 + GA1 operations, the first two units 4: GKT = 6 + GA1 may stay in flight -- 8 / 7 / 6)
      if (younger == GKT) wait_vmcnt<GKT>();
      else if (younger == GKT + NST) wait_vmcnt<GKT + NST>();
      else if (younger == GKT + 2 * NST) wait_vmcnt<GKT + 2 * NST>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    };
    if (wr == 1) {                                     // wave row 1 runs one barrier interval behind
      item_barrier();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    init_acc();
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
      PP_READ_B(bx, 1 * PP_UNIT);                      // B0 of k-tile 0 (later k-tiles get theirs one phase ahead)
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
        if (EPI == EPI_ACC) pp_partial_store<HI, true>(reinterpret_cast<float*>(p.C), p.ldc, acc, em0, en0, ln, tpa);
        else if (GRP) pp_partial_store<HI>(edst, eld, acc, em0, en0, ln, tpa);
        else pp_partial_store<HI>(p.partial + (size_t)ez * p.M * p.N, p.N, acc, em0, en0, ln, tpa);
      }
      ticket_publish();
      younger = GKT + 2 * NST;
    } else if (TACC) {
      PpOut out;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      {
        const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
        pp_epi_compute_q<EPI, HI>(p, acc, out, em0, en0, ln, tpa);
      }
      // pin the finished outputs here: hipcc must not sink the aux-dependent arithmetic into the store sequence below
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            if (!(hh == 1 && i >= HI)) {
              asm volatile("" : "+v"(out.o[hh][i][it]));
              if (EPI == EPI_GELU || EPI == EPI_GELU_D) asm volatile("" : "+v"(out.pre[hh][i][it]));
            }
      PP_STAMP(2);
      ticket_ready(); ticket_publish();
      PP_STAMP(3);
      asm volatile("" : "+v"(ln));
      pp_epi_store<EPI, HI>(p, out, em0, en0, ln);
      PP_STAMP(4);
      younger = GKT + ((EPI == EPI_GELU || EPI == EPI_GELU_D) ? 2 * NST : NST);     // stores issued after the DMA
    } else {                                           // fp32 output (accumulate / split-K atomics): row segments per register
      bf16_tile_epilogue<false>(p, acc[0], em0, en0, 0, 0, lane, smem);
      bf16_tile_epilogue<false>(p, acc[1], em0 + 64, en0, 0, 0, lane, smem);
      ticket_ready(); ticket_publish();
      younger = -1;                                     // (the atomics are younger than the staged units: uncounted -> wait for all)
    }
    ++nitem_done;
    if (!more) break;
  }
  queue_leave();
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
#undef PP_READ_A1
#undef PP_READ_B
#undef PP_SYNC_L
#undef PP_SYNC_C
}

static int pp_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <bool TA, bool TB, bool TACC, int EPI, int HI = 2>
static void pp_launch0(const GemmArgs& p, hipStream_t st) {
  static_assert(HI == 2 || TACC, "the 192- and 128-row variants exist for the bf16-output epilogue only");
  constexpr int smem = 131072 + 8 * 4096;            // the ring + one 4 KiB transposition buffer per wave = all 160 KiB
  auto kern = gemm_bf16_pp_kernel<TA, TB, TACC, EPI, HI>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = EPI == EPI_GROUP ? p.grp_items : (p.N / 256) * (p.M / (128 + 64 * HI)) * p.split_k;
  int cus = pp_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, p);
}

// The epilogue kind is a template parameter; only the combinations the encoder uses are instantiated:
//   X*W^T (forward): none, GELU, +residual, tanh;  dY*W (input gradient): none, dGELU, +residual;  everything else: none.
// tile_rows = 192 / 128 (the N = 768 shapes, see the header comment): forward none, input gradient none / +residual.
bool uc2_gemm_pp_supported(int trans_a, int trans_b, int c_f32, int epi, int tile_rows) {
  if (tile_rows == 192 || tile_rows == 128) {
    if (c_f32 || trans_a) return false;
    return trans_b ? (epi == EPI_NONE || epi == EPI_ADD) : (epi == EPI_NONE);
  }
  if (c_f32) return epi == EPI_NONE;
  if (!trans_a && !trans_b) return epi == EPI_NONE || epi == EPI_GELU || epi == EPI_ADD || epi == EPI_TANH;
  if (!trans_a && trans_b) return epi == EPI_NONE || epi == EPI_DGELU || epi == EPI_ADD;
  return epi == EPI_NONE;
}

// one launch over the split-K items of p.ngroup weight gradients (p.grp, uc2_gemm_wgrad_group in gemm.hip)
void uc2_gemm_pp_group_launch(const GemmArgs& p, hipStream_t st) { pp_launch0<true, true, true, EPI_GROUP>(p, st); }

void uc2_gemm_pp_launch(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st, int tile_rows) {
  if (tile_rows == 128) {
    if (!trans_b) pp_launch0<false, false, true, EPI_NONE, 0>(p, st);
    else if (p.epi == EPI_ADD) pp_launch0<false, true, true, EPI_ADD, 0>(p, st);
    else pp_launch0<false, true, true, EPI_NONE, 0>(p, st);
    return;
  }
  if (tile_rows == 192) {
    if (!trans_b) pp_launch0<false, false, true, EPI_NONE, 1>(p, st);
    else if (p.epi == EPI_ADD) pp_launch0<false, true, true, EPI_ADD, 1>(p, st);
    else pp_launch0<false, true, true, EPI_NONE, 1>(p, st);
    return;
  }
  if (p.c_f32 && p.partial) {                          // the bf16-output kernels double as partial-tile producers
    if (!trans_a && !trans_b) pp_launch0<false, false, true, EPI_NONE>(p, st);
    else if (!trans_a && trans_b) pp_launch0<false, true, true, EPI_NONE>(p, st);
    else if (trans_a && !trans_b) pp_launch0<true, false, true, EPI_NONE>(p, st);
    else pp_launch0<true, true, true, EPI_NONE>(p, st);
  } else if (p.c_f32 && p.accumulate && !(p.atomic & 1) && p.split_k == 1 && !p.bias && p.alpha == 1.0f && !p.alpha_dev && trans_a && trans_b) {
    pp_launch0<true, true, true, EPI_ACC>(p, st);      // dE += dlogits^T z: the accumulators leave as whole lines, C read ahead
  } else if (p.c_f32) {
    if (!trans_a && !trans_b) pp_launch0<false, false, false, EPI_NONE>(p, st);
    else if (!trans_a && trans_b) pp_launch0<false, true, false, EPI_NONE>(p, st);
    else if (trans_a && !trans_b) pp_launch0<true, false, false, EPI_NONE>(p, st);
    else pp_launch0<true, true, false, EPI_NONE>(p, st);
  } else if (!trans_a && !trans_b) {
    if (p.epi == EPI_GELU && !p.aux_out) pp_launch0<false, false, true, EPI_GELU_NOAUX>(p, st);
    else if (p.epi == EPI_GELU && p.aux_deriv) pp_launch0<false, false, true, EPI_GELU_D>(p, st);
    else if (p.epi == EPI_GELU) pp_launch0<false, false, true, EPI_GELU>(p, st);
    else if (p.epi == EPI_ADD) pp_launch0<false, false, true, EPI_ADD>(p, st);
    else if (p.epi == EPI_TANH) pp_launch0<false, false, true, EPI_TANH>(p, st);
    else pp_launch0<false, false, true, EPI_NONE>(p, st);
  } else if (!trans_a && trans_b) {
    if (p.epi == EPI_DGELU && p.aux_deriv) pp_launch0<false, true, true, EPI_MUL>(p, st);
    else if (p.epi == EPI_DGELU) pp_launch0<false, true, true, EPI_DGELU>(p, st);
    else if (p.epi == EPI_ADD) pp_launch0<false, true, true, EPI_ADD>(p, st);
    else pp_launch0<false, true, true, EPI_NONE>(p, st);
  } else if (trans_a && !trans_b) {
    pp_launch0<true, false, true, EPI_NONE>(p, st);
  } else {
    pp_launch0<true, true, true, EPI_NONE>(p, st);
  }
}
