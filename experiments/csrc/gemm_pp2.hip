#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------------
// Ping-pong GEMM with TWO phases per k-tile (variant 11).  Same tile (256 x 256 x 64, 8 waves = 2 x 4, 128 x 64 per wave), same
// ring of eight 16-KiB units (B0, A0, B1, A1 of two k-tiles), same persistent item walk, epilogues and wave-row offset as
// gemm_pp.hip -- read its header first.  What changes is the granularity of the hand-over between the two wave rows:
//
//     phase E:  read B0, B1, A0 (16 fragment reads)   issue units B1, A1 of k-tile kt+1   -> 16 MFMAs  (A0 x B0, A0 x B1)
//     phase O:  read A1 (8 fragment reads)            issue units B0, A0 of k-tile kt+2   -> 16 MFMAs  (A1 x B1, A1 x B0)
//
// In-kernel stamps put a K = 768 main loop of gemm_pp.hip at 323 cycles per barrier interval for 256 cycles of MFMA: every
// hand-over (barrier release -> first MFMA of the other wave row) leaves the matrix pipe idle for ~67 cycles, once per 8 MFMAs.
// With 16 MFMAs per C section the same hand-over is paid once per 512 cycles.  The fragment and LDS-DMA work per MFMA is unchanged
// (12 reads and 4 DMA issues per 16 MFMAs), only bunched 16 + 4 / 8 + 4 instead of 8 + 2 / 4 + 2.
// Ring discipline: a unit is overwritten (its DMA issued) in the L section after the phase that read it -- ONE barrier interval
// after the delayed wave row's reads, not two as in gemm_pp.hip -- so every L section waits for its own fragment reads
// (lgkmcnt(0)) BEFORE its barrier: behind that barrier nobody still reads the slots the next L section refills.
// Counted waits: the end of phase E retires unit A1 of this k-tile (4 younger units may fly), the end of phase O unit B1 of the
// next (3 younger units).  256-row tiles only.
// ------------------------------------------------------------------------------------------------------
#include "gemm_pp.h"

#ifndef UC2_PP_DIAG
#define UC2_PP_DIAG 0            /* 1: build the main-loop diagnostics 0x100 / 0x200 (tests/bench_pp.py) */
#endif

template <bool TA, bool TB, bool TACC, int EPI, int HI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp2_kernel(GemmArgs p) {
  constexpr int RW = 64 + 32 * HI, RT = 2 * RW;           // rows per wave row / per tile (128 / 256, or 96 / 192)
  constexpr int GA1 = HI == 2 ? 2 : 1;                      // LDS-DMA instructions per wave for unit A1
  constexpr int GKT = 6 + GA1;                             // ... per k-tile
  constexpr int NST = (2 + HI) * 4;                        // bf16 stores per wave in the direct epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;

  // ---- work items: (tile, k-split), tile-major inside a split.  Workgroups land on XCD blockIdx % 8; each XCD
  //      owns a contiguous range of items (neighbouring tiles share an A row panel in that XCD's L2) and its
  //      workgroups stride through it together.
  const int nbx = p.N / 256, ntile = nbx * (p.M / RT);
  const int nitems = ntile * p.split_k;
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

  const unsigned stepa = (TA ? 64u * (unsigned)p.lda : 64u) * 2u;      // bytes per k-tile
  const unsigned stepb = (TB ? 64u * (unsigned)p.ldb : 64u) * 2u;
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

  // main-loop diagnostics (with 0x800): 0x100 = no LDS-DMA issue
  const bool dg_nodma = UC2_PP_DIAG && (p.atomic & 0x100) != 0;
  bool more = false;                                   // another item follows the current one
  bf16x8 (&b0)[4] = bx;
  bf16x8 (&b1)[4] = by;
  // end of an L section: counted wait, this phase's fragment reads, publish
#define PP2_SYNC_L(UNITS)                                                                                      \
  do {                                                                                                         \
    pp_wait_small(2 * (UNITS));                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
  } while (0)
  auto body = [&](auto tail_c, int kt) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_c)::value;     // one of the last two k-tiles of the item
    unsigned cb = (kt & 1) * 65536u;                   // this k-tile's buffer
    asm volatile("" : "+s"(cb));
    const int nb = (kt & 1) ^ 1;                       // buffer of k-tile kt+1 (kt+2 shares this tile's)
    const bool c1 = !TAIL || kt + 1 < nt || more;      // k-tile kt+1 exists (in this item or the next)
    const bool c2 = !TAIL || kt + 2 < nt || more;
    // ---- phase E
    PP_READ_B(b0, cb + 1 * PP_UNIT);
    PP_READ_A(cb + 0 * PP_UNIT);
    PP_READ_B(b1, cb + 2 * PP_UNIT);
    if (c1 && !dg_nodma) { PP_ISSUE(2, nb); PP_ISSUE(3, nb); }
    PP2_SYNC_L(c1 ? 4 : 0);                            // A1 of this k-tile has landed; B0, A0 (kt+1) and the two units just issued may fly
    PP_MFMA(0, 0, b0);
    PP_MFMA(0, 1, b1);
    PP_SYNC_C();
    // ---- phase O
    if (TAIL && kt == nt - 2 && more) setup(nxt);      // from here on the stream fetches the next item (its first six units, prologue order)
    PP_READ_A1(cb + 3 * PP_UNIT);
    if (c2 && !dg_nodma) { PP_ISSUE(1, nb ^ 1); PP_ISSUE(0, nb ^ 1); }
    PP2_SYNC_L((c1 ? 1 : 0) + (c2 ? 2 : 0));           // B0, A0, B1 of k-tile kt+1 have landed; A1 (kt+1) and the two units just issued may fly
    PP_MFMA(1, 1, b1);
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
      // units B0, A0, B1 of k-tile 0: this wave's part has landed (A1 of k-tile 0, B0 and A0 of k-tile 1 and the stores are younger)
      if (younger == GKT) wait_vmcnt<6>();
      else if (younger == GKT + NST) wait_vmcnt<6 + NST>();
      else if (younger == GKT + 2 * NST) wait_vmcnt<6 + 2 * NST>();
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
      int kt = 0;
      for (; kt + 2 < nt; kt += 2) { body(F{}, kt); body(F{}, kt + 1); }     // nt is even (host-checked)
      body(T{}, kt);
      body(T{}, kt + 1);
    } else if (more) {
      setup(nxt);
      PP_PROLOGUE();
    }
    // The next item's first six units are in flight or landed (issued by the tail above); the epilogue below touches
    // only the transposition buffers behind the ring.
    PP_STAMP(1);
    const int em0 = m0 + wr * RW, en0 = n0 + wc * 64, ez = zsplit;
    item = nxt;
    if (more) { m0 = m0x; n0 = n0x; nt = ntx; zsplit = zx; }
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
    } else if (TACC && p.partial) {                     // split-K item of a two-stage reduction (fp32 partial, plain stores)
      {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
        ticket_ready();
        pp_partial_store<HI>(p.partial + (size_t)ez * p.M * p.N, p.N, acc, em0, en0, ln, tpa);
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
#undef PP2_SYNC_L
}

static int pp2_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <bool TA, bool TB, bool TACC, int EPI, int HI = 2>
static void pp2_launch0(const GemmArgs& p, hipStream_t st) {
  static_assert(HI == 2 || TACC, "the 192-row variant exists for the bf16-output epilogue only");
  constexpr int smem = 131072 + 8 * 4096;            // the ring + one 4 KiB transposition buffer per wave = all 160 KiB
  auto kern = gemm_bf16_pp2_kernel<TA, TB, TACC, EPI, HI>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = (p.N / 256) * (p.M / (128 + 64 * HI)) * p.split_k;
  int cus = pp2_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, p);
}

// The epilogue kind is a template parameter; only the combinations the encoder uses are instantiated:
//   X*W^T (forward): none, GELU, +residual, tanh;  dY*W (input gradient): none, dGELU, +residual;  everything else: none.
// tile_rows = 192 (the N = 768 shapes, see the header comment): forward none, input gradient none / +residual.
bool uc2_gemm_pp2_supported(int trans_a, int trans_b, int c_f32, int epi, int tile_rows) {
  if (tile_rows != 256) return false;
  if (c_f32) return epi == EPI_NONE;
  if (!trans_a && !trans_b) return epi == EPI_NONE || epi == EPI_GELU || epi == EPI_ADD || epi == EPI_TANH;
  if (!trans_a && trans_b) return epi == EPI_NONE || epi == EPI_DGELU || epi == EPI_ADD;
  return epi == EPI_NONE;
}

void uc2_gemm_pp2_launch(const GemmArgs& p, int trans_a, int trans_b, hipStream_t st, int tile_rows) {
  if (p.c_f32 && p.partial) {                          // the bf16-output kernels double as partial-tile producers
    if (!trans_a && !trans_b) pp2_launch0<false, false, true, EPI_NONE>(p, st);
    else if (!trans_a && trans_b) pp2_launch0<false, true, true, EPI_NONE>(p, st);
    else if (trans_a && !trans_b) pp2_launch0<true, false, true, EPI_NONE>(p, st);
    else pp2_launch0<true, true, true, EPI_NONE>(p, st);
  } else if (p.c_f32) {
    if (!trans_a && !trans_b) pp2_launch0<false, false, false, EPI_NONE>(p, st);
    else if (!trans_a && trans_b) pp2_launch0<false, true, false, EPI_NONE>(p, st);
    else if (trans_a && !trans_b) pp2_launch0<true, false, false, EPI_NONE>(p, st);
    else pp2_launch0<true, true, false, EPI_NONE>(p, st);
  } else if (!trans_a && !trans_b) {
    if (p.epi == EPI_GELU && !p.aux_out) pp2_launch0<false, false, true, EPI_GELU_NOAUX>(p, st);
    else if (p.epi == EPI_GELU && p.aux_deriv) pp2_launch0<false, false, true, EPI_GELU_D>(p, st);
    else if (p.epi == EPI_GELU) pp2_launch0<false, false, true, EPI_GELU>(p, st);
    else if (p.epi == EPI_ADD) pp2_launch0<false, false, true, EPI_ADD>(p, st);
    else if (p.epi == EPI_TANH) pp2_launch0<false, false, true, EPI_TANH>(p, st);
    else pp2_launch0<false, false, true, EPI_NONE>(p, st);
  } else if (!trans_a && trans_b) {
    if (p.epi == EPI_DGELU && p.aux_deriv) pp2_launch0<false, true, true, EPI_MUL>(p, st);
    else if (p.epi == EPI_DGELU) pp2_launch0<false, true, true, EPI_DGELU>(p, st);
    else if (p.epi == EPI_ADD) pp2_launch0<false, true, true, EPI_ADD>(p, st);
    else pp2_launch0<false, true, true, EPI_NONE>(p, st);
  } else if (trans_a && !trans_b) {
    pp2_launch0<true, false, true, EPI_NONE>(p, st);
  } else {
    pp2_launch0<true, true, true, EPI_NONE>(p, st);
  }
}
