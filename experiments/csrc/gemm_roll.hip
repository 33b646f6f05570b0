#include "gemm_pp.h"

// ------------------------------------------------------------------------------------------------------
// Rolling-epilogue variant of the persistent ping-pong GEMM (variant 10; same tile, ring, phases and wave-row
// offset as gemm_pp.hip -- read its header first).  bf16 output, A k-contiguous, 256-row tiles, >= 4 k-tiles.
//
// In gemm_pp.hip both wave rows of a CU finish a tile together and the matrix pipe idles while they convert,
// transpose (LDS) and store 128 KiB of output: ~6 k cycles per item against a K = 768 main loop of 31 k.  Here the
// k-tile stream never stops at an item boundary (the ring already carried the next item's operands), and the
// epilogue of item n is cut into four blocks of 32 rows x 64 columns per wave that ride in the L sections (and the
// stores in the C sections) of the phases around the boundary -- beside the partner wave's MFMAs:
//
//     last k-tile of item n      phase 2   L: S1(X0)                                  (rows 0-63 of the wave are final after phase 1)
//                                phase 3   L: R(X0), ST(X0), S1(X1) [, bias -> acc rows 0-63]
//     first k-tile of item n+1   phase 0   L: R(X1), ST(X1), S1(X2)                   (rows 64-127 final after phase 3)
//                                phase 1   L: R(X2), ST(X2), S1(X3) [, bias -> acc rows 64-127]
//                                phase 2   L: R(X3), ST(X3)
//
//     S1(X) : accumulators of block X -> lane-half exchange -> epilogue arithmetic -> bf16 -> ds_write (row, half) layout
//     R(X)  : ds_read the block back in the LINE layout (whole 128-byte lines per 8 lanes), first thing in the L section
//     ST(X) : 4 non-temporal 16-byte stores per lane, after the fragment reads and the LDS-DMA of the L section have been
//             issued (a counted lgkmcnt: only R's four reads must have returned); the C sections stay pure MFMA and the
//             block is never live beside the next block's S1 temporaries (the kernel has no registers to spare)
//
// A block's accumulators are free one L section after the phase that finished them, i.e. before that phase comes round
// again in the next item; the first MFMA of every accumulator in the next item takes C = 0 (no bias) or finds the bias
// written by the L sections marked above.  The transposition buffer (4 KiB per wave) holds one block at a time; LDS
// executes a wave's operations in order, so R(X) followed by the writes of S1(X+1) needs no wait.
// Stores count in vmcnt like the LDS-DMA: the counted wait of an L section allows, besides the 8 youngest DMAs, the
// stores issued in this and the four L sections before it (E below): they all follow the DMA of the unit the wait retires.
// The workgroup's last item has nothing to hide behind: its epilogue is the exposed one of gemm_pp.hip.
// ------------------------------------------------------------------------------------------------------

#ifndef UC2_ROLL_DIAG
#define UC2_ROLL_DIAG 0                 /* 1: build the diagnostic launch modes (UC2_GEMM_DIAG bits, see the kernel) */
#endif

template <int N> __device__ __forceinline__ void roll_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <bool TB, int EPI, bool HASB>
__global__ __launch_bounds__(512, 2) void gemm_bf16_roll_kernel(GemmArgs p) {
  constexpr bool TA = false;
  constexpr int HI = 2, RW = 128, RT = 256;
  constexpr int NS = 4;                                    // stores per block and wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(p.B);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;

  // ---- work items (whole tiles; no split-K here): same XCD-contiguous static partition as gemm_pp.hip
  const int nbx = p.N / 256, ntile = nbx * (p.M / RT);
  const int nt = p.K / 64;                                 // k-tiles per item: even, >= 4 (host-checked)
  int item, item_end, item_step;
  {
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nx = min(G, 8);
    const int q = ntile / nx, r = ntile % nx;
    const int beg = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    item_end = beg + q + (xcd < r ? 1 : 0);
    item_step = (G - xcd + 7) >> 3;
    item = beg + slot;
  }
  if (item >= item_end) return;
  // De-phasing (UC2_GEMM_SKEW(n)): with equal items every CU reaches its item boundaries at the same moment and the chip writes
  // 32 MiB within a few phases; HBM takes microseconds to drain that, store acknowledgements come late, and the in-order vmcnt
  // of the LDS-DMA waits sits behind them.  Eight groups (by slot within the XCD) start n * ~1k cycles apart.
  if (UC2_ROLL_DIAG && p.skew > 0) {
    const int g = (blockIdx.x >> 3) & 7;
    for (int i = 0; i < g * p.skew; ++i) __builtin_amdgcn_s_sleep(16);
  }

  const unsigned stepa = 64u * 2u;                                      // bytes per k-tile
  const unsigned stepb = (TB ? 64u * (unsigned)p.ldb : 64u) * 2u;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void_p)smem);
  unsigned fa[2], fb;                                  // (units A0 and A1 have the same 64 rows per wave row here)
  fa[0] = lds0 + pp_frag_off<TA>(wr * 64, lane);
  fa[1] = lds0 + pp_frag_off<TA>(wr * 64 + 32, lane);
  fb = lds0 + pp_frag_off<TB>(wc * 32, lane);
  // transposition buffer of this wave: LINE layout address (+ it * 1024) and the (row, half) address of piece 0
  // (piece k = 2j + g is at rh0 ^ (k << 5): the buffer is 4-KiB aligned and the swizzle only touches bits 4-6)
  unsigned tp_line, tp_rh0, tp_q0;
  {
    const unsigned tb = lds0 + 131072u + (unsigned)w * 4096u;
    const int lr = lane >> 3, lc = lane & 7, r = lane & 31, h = lane >> 5;
    tp_line = tb + lr * 128 + ((lc ^ lr) << 4);
    tp_rh0 = tb + r * 128 + ((h ^ (r & 7)) << 4);
    tp_q0 = tb + r * 128 + ((r & 7) << 4) + 8 * h;    // 8-byte half h of 16-byte chunk 0 of row r (chunk k: ^ (k << 4))
  }

  int m0, n0;                   // the item being computed
  int m0x, n0x;                 // the item being staged
  int em0 = 0, en0 = 0;         // first row / column of this wave's part of the item whose epilogue is rolling
  unsigned src[4];              // staging sources (wave-instruction w of each unit type) as byte offsets from A / B; wave-instruction
                                // w + 8 is a uniform distance further: 128 rows (k-contiguous image) or 32 k-rows (k-strided)
  const unsigned dqa = 128u * (unsigned)p.lda * 2u, dqb = (TB ? 32u : 128u) * (unsigned)p.ldb * 2u;
  auto setup = [&](int it) __attribute__((always_inline)) {
    {
      const int cg = p.col_group, mt = p.M / RT;
      const int per_group = mt * cg;
      const int g = it / per_group, r = it - g * per_group;
      const int cw = min(cg, nbx - g * cg);
      const int tm = (g * cg + cw <= nbx && cw == cg) ? r / cg : r / cw;
      const int tc = r - tm * ((cw == cg) ? cg : cw);
      m0x = tm * RT; n0x = (g * cg + tc) * 256;
    }
    int ln = lane;
    asm volatile("" : "+v"(ln));
    src[0] = (unsigned)((const char*)pp_src<TA, 0, HI>(A, p.lda, p.M, m0x, 0, w, ln) - (const char*)A);
    src[1] = (unsigned)((const char*)pp_src<TB, 1, HI>(B, p.ldb, p.N, n0x, 0, w, ln) - (const char*)B);
    src[2] = (unsigned)((const char*)pp_src<TB, 2, HI>(B, p.ldb, p.N, n0x, 0, w, ln) - (const char*)B);
    src[3] = (unsigned)((const char*)pp_src<TA, 3, HI>(A, p.lda, p.M, m0x, 0, w, ln) - (const char*)A);
  };

#define PP_ISSUE(J, BUF)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                            \
      const char* base_ = ((J) == 0 || (J) == 3) ? (const char*)A + (q ? dqa : 0u) : (const char*)B + (q ? dqb : 0u); \
      __builtin_amdgcn_global_load_lds((glb_void_p)(base_ + src[J]),                                           \
                                       (lds_void_p)(smem + (BUF) * 65536 + (J) * PP_UNIT + (w + 8 * q) * 1024), 16, 0, 0); \
    }                                                                                                          \
    src[J] += ((J) == 0 || (J) == 3) ? stepa : stepb;                                                          \
  } while (0)
#define PP_PROLOGUE() do { PP_ISSUE(1, 0); PP_ISSUE(0, 0); PP_ISSUE(2, 0); PP_ISSUE(3, 0); PP_ISSUE(1, 1); PP_ISSUE(0, 1); } while (0)

  // diagnostic (UC2_GEMM_DIAG bit 16, aux_out = uint32 [8 waves][8]): cycles this wave spent in the counted vmcnt waits of
  // plain L sections [0], of L sections whose window holds rolled stores [1], and at the barrier that ends an L section [2]
  const bool dg_stamp = UC2_ROLL_DIAG && ((p.atomic >> 8) & 16);
  unsigned stamp_t0 = 0, stamp_acc[3] = {0, 0, 0};
#define STAMP_T0() do { if (dg_stamp) { unsigned long long t64_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64_) :: "memory"); stamp_t0 = (unsigned)t64_; } } while (0)
#define STAMP_ADD(I) do { if (dg_stamp) { unsigned long long t64_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t64_) :: "memory"); stamp_acc[I] += (unsigned)t64_ - stamp_t0; } } while (0)
  f32x16 acc[2][2][2];                                 // [A half][i][j]
  bf16x8 a[2][4], bx[4], by[4];

  // 8 MFMAs of one phase in two halves (k16 steps S0 .. S0+1); ZC: the accumulators start at C = 0 (first k16 step
  // of an item whose accumulators still held the previous item's results)
#define PP_MFMA2(H, JB, BREG, S0, ZC)                                                                           \
  do {                                                                                                         \
    _Pragma("unroll") for (int s = (S0); s < (S0) + 2; ++s)                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
        if ((ZC) && s == 0) { const f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
          acc[H][i][JB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BREG[s], a[i][s], z_, 0, 0, 0); }           \
        else acc[H][i][JB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BREG[s], a[i][s], acc[H][i][JB], 0, 0, 0); \
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
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
      pp_read<TA, 0>(a[i][0], fa[i] + (BASEOFF)); pp_read<TA, 1>(a[i][1], fa[i] + (BASEOFF));                \
      pp_read<TA, 2>(a[i][2], fa[i] + (BASEOFF)); pp_read<TA, 3>(a[i][3], fa[i] + (BASEOFF));                \
    }                                                                                                          \
  } while (0)
#define PP_READ_B(BREG, BASEOFF)                                                                               \
  do {                                                                                                         \
    pp_read<TB, 0>(BREG[0], fb + (BASEOFF)); pp_read<TB, 1>(BREG[1], fb + (BASEOFF));                          \
    pp_read<TB, 2>(BREG[2], fb + (BASEOFF)); pp_read<TB, 3>(BREG[3], fb + (BASEOFF));                          \
  } while (0)
  // end of an L section.  ALLOW = units that may stay in flight (4 in the stream; fewer while the last item drains);
  // E = stores issued in this L section and the four before it (compile-time)
#define PP_SYNC_L(ALLOW, E)                                                                                    \
  do {                                                                                                         \
    STAMP_T0();                                                                                                \
    if ((E) > 0 && prev && counted) roll_wait_vm<8 + (E)>();                                                   \
    else { const int al_ = (ALLOW) > 4 ? 4 : ((ALLOW) < 0 ? 0 : (ALLOW)); pp_wait_small(2 * al_); }            \
    STAMP_ADD(((E) > 0) ? 1 : 0);                                                                              \
    PP_SYNC_L_REST();                                                                                          \
  } while (0)
#define PP_SYNC_L_REST()                                                                                       \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    STAMP_T0();                                                                                                \
    __builtin_amdgcn_s_barrier();                                                                              \
    STAMP_ADD(2);                                                                                              \
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

  // diagnostics (UC2_GEMM_DIAG): 1 = no output stores of the rolled blocks, 2 = no rolled epilogue at all (main-loop stream only;
  // results are wrong), 4 = write-back instead of non-temporal stores
  const bool dg_nostore = UC2_ROLL_DIAG && ((p.atomic >> 8) & 1), dg_noepi = UC2_ROLL_DIAG && ((p.atomic >> 8) & 2), dg_wb = UC2_ROLL_DIAG && ((p.atomic >> 8) & 4);
  const bool dg_tiled = UC2_ROLL_DIAG && ((p.atomic >> 8) & 32);          // 32 = tile-major output (each 256 x 256 tile a dense 128 KiB block; layout experiment)
  const bool dg_near = UC2_ROLL_DIAG && ((p.atomic >> 8) & 8);            // 8 = every store of a workgroup goes to the same 8 rows (L2-resident, results wrong)
  const bool counted = !(dg_nostore || dg_noepi);      // the rolled stores are in the vmcnt queue
  // ---- the rolling epilogue stages --------------------------------------------------------------------------------
  // S1: block (hh, i) -> (row, half) pieces in the transposition buffer.  After the lane-half exchange lane (row m, half h)
  // holds columns 32 j + 16 g + 8 h .. + 7 (piece k = 2 j + g) -- see pp_epi_compute in gemm_pp.h.
  auto s1 = [&](auto hh_c, auto i_c) __attribute__((always_inline)) {
    constexpr int hh = decltype(hh_c)::value, i = decltype(i_c)::value;
    // Register group 4c .. 4c+3 of accumulator block j is four consecutive columns 32 j + 8 c + 4 h .. + 3 of row (lane & 31): two
    // packed conversions and one 8-byte LDS store put them where the LINE reads expect them (16-byte chunk 4 j + c of the row,
    // half h of the chunk) -- no lane-half exchange (16 v_permlane32_swap per block cost more than everything else in S1).
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[hh][i][j][4 * c + e];
        if (EPI == EPI_GELU_NOAUX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_bf(v[e]);
        } else if (EPI == EPI_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanh_bf(v[e]);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        const unsigned qa = tp_q0 ^ (unsigned)((4 * j + c) << 4);
        asm volatile("ds_write_b64 %0, %1" :: "v"(qa), "v"(o) : "memory");
      }
  };
  // R + ST: the block comes back in the LINE layout and leaves as whole 128-byte lines.  NREADS = LDS reads issued after
  // these four in the same L section (they may stay in flight while the stores go out: LDS returns in order).
  // The per-lane part of the store address (row lr of 8, 16-byte chunk lc) is the same for every block: one register.
  const unsigned st_voff = (unsigned)(((lane >> 3) * p.ldc + 8 * (lane & 7)) * 2);
  auto rd = [&](bf16x8 (&o)[4]) __attribute__((always_inline)) {
    tp_read_o<0>(o[0], tp_line); tp_read_o<1024>(o[1], tp_line); tp_read_o<2048>(o[2], tp_line); tp_read_o<3072>(o[3], tp_line);
  };
  auto st = [&](auto hh_c, auto i_c, auto nreads_c, bf16x8 (&o)[4]) __attribute__((always_inline)) {
    constexpr int hh = decltype(hh_c)::value, i = decltype(i_c)::value, NREADS = decltype(nreads_c)::value;
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]) : "n"(NREADS) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    char* cb_ = reinterpret_cast<char*>(p.C) + ((size_t)(em0 + hh * 64 + i * 32) * p.ldc + en0) * 2;      // wave-uniform
    if (dg_tiled) {
      const size_t tile_ = (size_t)(em0 / 256) * (p.N / 256) + (size_t)(en0 / 256);
      cb_ = reinterpret_cast<char*>(p.C) + tile_ * 131072 + (size_t)((em0 & 255) + hh * 64 + i * 32) * 512 + (size_t)(en0 & 255) * 2;
    }
    if (dg_near) cb_ = reinterpret_cast<char*>(p.C) + ((size_t)(blockIdx.x * 8) * p.ldc + wc * 64 + wr * 256) * 2;
    if (dg_nostore) return;
    if (dg_tiled) {
      const unsigned voff_ = (unsigned)(((lane >> 3) * 256 + 8 * (lane & 7)) * 2);
#pragma unroll
      for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(o[it], reinterpret_cast<bf16x8*>(cb_ + (size_t)(8 * it) * 512 + voff_));
    } else if (dg_wb || dg_near) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<bf16x8*>(cb_ + (dg_near ? (size_t)0 : (size_t)(8 * it) * p.ldc * 2) + st_voff) = o[it];
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        __builtin_nontemporal_store(o[it], reinterpret_cast<bf16x8*>(cb_ + (size_t)(8 * it) * p.ldc * 2 + st_voff));
    }
  };
  // accumulators of rows hh*64 .. +63 of the wave <- bias of the item that starts at column nc (register 8g+4cc+e of
  // block j is column 32j+16g+8cc+4h+e): scalar loads + one select per value
  auto init_half = [&](auto hh_c, int nc) __attribute__((always_inline)) {
    constexpr int hh = decltype(hh_c)::value;
    typedef __attribute__((ext_vector_type(16))) float f32x16c;
    typedef const __attribute__((address_space(4))) f32x16c* cvec_p;
    asm volatile("" : "+s"(nc));                       // opaque: the 32 select results must not be kept from one call to the next
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x16c bvv = *(cvec_p)(uintptr_t)(p.bias + nc + 32 * j + 16 * g);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bvv[8 * cc + e], hi = bvv[8 * cc + 4 + e];
            const float b = (lane >> 5) ? hi : lo;
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[hh][i][j][8 * g + 4 * cc + e] = b;
          }
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  bool more = false;                                   // another item follows the current one
  bool prev = false;                                   // a rolled item precedes the current one (false for the workgroup's first item)
  // ROLE 0: plain k-tile; 1: first k-tile of an item that follows a rolled one (carries its blocks X1..X3);
  //      2: the k-tile after that (only the store counts of the waits differ)
  auto body = [&](auto role_c, auto tail_c, auto swap_c, auto zc_c, int kt) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool SW = decltype(swap_c)::value;
    constexpr bool LAST = TAIL && SW;                  // k-tile nt-1 (nt is even)
    constexpr bool ZC = decltype(zc_c)::value;         // first k-tile of an item without a bias: the accumulators start at C = 0
    bf16x8 (&b0)[4] = SW ? by : bx;
    bf16x8 (&b1)[4] = SW ? bx : by;
    const int nunits = 4 * nt;
    unsigned cb = (kt & 1) * 65536u;
    asm volatile("" : "+s"(cb));
    const int nb = (kt & 1) ^ 1;
    const int f0 = 4 * kt;
    const bool drain = TAIL && !more;                  // the workgroup's last item: the stream ends with it
    const bool roll = LAST && more && !dg_noepi;
    bf16x8 o[4];
    using N12 = std::integral_constant<int, 12>;       // 4 B-fragment reads (8 ds_read_b64_tr_b16 when TB) or 8 A-fragment reads follow R
    using NA = std::integral_constant<int, 8>;
    using NB = std::integral_constant<int, TB ? 8 : 4>;
    // ---- phase 0
    if (ROLE == 1 && prev && !dg_noepi) rd(o);                      // R(X1)
    PP_READ_A(cb + 0 * PP_UNIT);
    if (!TAIL || f0 + 6 < nunits || more) PP_ISSUE(2, nb);
    if (ROLE == 1 && prev && !dg_noepi) { st(I0{}, I1{}, NA{}, o); s1(I1{}, I0{}); }                         // ST(X1), S1(X2)
    PP_SYNC_L(drain ? nunits - 3 - f0 : 4, ROLE == 1 ? 2 * NS : ROLE == 2 ? 3 * NS : 0);
    PP_MFMA2(0, 0, b0, 0, ZC);
    PP_MFMA2(0, 0, b0, 2, false);
    PP_SYNC_C();
    // ---- phase 1
    if (ROLE == 1 && prev && !dg_noepi) rd(o);                      // R(X2)
    PP_READ_B(b1, cb + 2 * PP_UNIT);
    if (!TAIL || f0 + 7 < nunits || more) PP_ISSUE(3, nb);
    if (ROLE == 1) {
      if (prev && !dg_noepi) {
        st(I1{}, I0{}, NB{}, o);                       // ST(X2)
        s1(I1{}, I1{});                                // S1(X3)
      }
      if (HASB) init_half(I1{}, n0 + wc * 64);
    }
    PP_SYNC_L(drain ? nunits - 4 - f0 : 4, ROLE == 1 ? 3 * NS : ROLE == 2 ? 2 * NS : 0);
    PP_MFMA2(0, 1, b1, 0, ZC);
    PP_MFMA2(0, 1, b1, 2, false);
    PP_SYNC_C();
    // ---- phase 2
    if (TAIL && !SW && more) setup(item + item_step);
    if (ROLE == 1 && prev && !dg_noepi) rd(o);                      // R(X3)
    PP_READ_A1(cb + 3 * PP_UNIT);
    if (!TAIL || f0 + 8 < nunits || more) PP_ISSUE(1, nb ^ 1);
    if (ROLE == 1 && prev && !dg_noepi) st(I1{}, I1{}, NA{}, o);    // ST(X3)
    if (LAST) { if (roll) { em0 = m0 + wr * RW; en0 = n0 + wc * 64; s1(I0{}, I0{}); } }       // S1(X0)
    PP_SYNC_L(drain ? nunits - 5 - f0 : 4, ROLE == 1 ? 4 * NS : ROLE == 2 ? NS : 0);
    PP_MFMA2(1, 1, b1, 0, ZC);
    PP_MFMA2(1, 1, b1, 2, false);
    PP_SYNC_C();
    // ---- phase 3
    if (LAST) { if (roll) rd(o); }                     // R(X0)
    if (!TAIL || kt + 1 < nt || more) PP_READ_B(b1, (cb ^ 65536u) + 1 * PP_UNIT);
    if (!TAIL || f0 + 9 < nunits || more) PP_ISSUE(0, nb ^ 1);
    if (LAST) {
      if (roll) {
        st(I0{}, I0{}, NB{}, o);                       // ST(X0)
        s1(I0{}, I1{});                                // S1(X1)
        if (HASB) init_half(I0{}, n0x + wc * 64);
      }
    }
    if (LAST) {                                        // (the wait of a rolling last phase allows its own four stores)
      if (roll && counted) { STAMP_T0(); roll_wait_vm<8 + NS>(); STAMP_ADD(1); PP_SYNC_L_REST(); } else { PP_SYNC_L(drain ? nunits - 6 - f0 : 4, 0); }
    } else {
      PP_SYNC_L(drain ? nunits - 6 - f0 : 4, ROLE == 1 ? 4 * NS : 0);
    }
    PP_MFMA2(1, 0, b0, 0, ZC);
    PP_MFMA2(1, 0, b0, 2, false);
    if (LAST) {
      if (more) {
        PP_SYNC_C();
      } else {                                         // wave row 1 has no partner barrier left after its last C section
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      PP_SYNC_C();
    }
  };

  setup(item);
  m0 = m0x; n0 = n0x;
  PP_PROLOGUE();
  // first item: units 0 and 1 have landed (this wave's part), publish; wave row 1 runs one barrier interval behind
  if (wr == 1) {
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);
  if (HASB) init_half(I0{}, n0 + wc * 64);             // (rows 64-127 get theirs in phase 1 of the first k-tile, like every item)
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 0) {
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);
  PP_READ_B(bx, 1 * PP_UNIT);                          // B0 of k-tile 0 (later k-tiles, and later items, get theirs one phase ahead)
  {
    using F = std::false_type; using T = std::true_type;
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    for (;;) {
      more = item + item_step < item_end;
      using Z = std::integral_constant<bool, !HASB>;
      body(R1{}, F{}, F{}, Z{}, 0);
      body(R2{}, F{}, T{}, F{}, 1);
      int kt = 2;
      for (; kt + 2 < nt; kt += 2) { body(R0{}, F{}, F{}, F{}, kt); body(R0{}, F{}, T{}, F{}, kt + 1); }
      body(R0{}, T{}, F{}, F{}, kt);
      body(R0{}, T{}, T{}, F{}, kt + 1);
      if (!more) break;
      item += item_step;
      m0 = m0x; n0 = n0x;
      prev = true;
    }
  }
  if (dg_stamp) {
    if (lane == 0 && blockIdx.x == 8) {
      unsigned* o_ = reinterpret_cast<unsigned*>(p.aux_out) + w * 8;
      o_[0] = stamp_acc[0]; o_[1] = stamp_acc[1]; o_[2] = stamp_acc[2];
    }
    return;
  }
  // ---- the last item of this workgroup: exposed epilogue (gemm_pp.h)
  {
    PpOut out;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int fm0 = m0 + wr * RW, fn0 = n0 + wc * 64;
    {
      const TpAddr tpa = tp_addr(lds0 + 131072u + (unsigned)w * 4096u, ln);
      // (the accumulators hold bias + product already: HASB items start at the bias)
      pp_epi_compute_q<EPI, HI>(p, acc, out, fm0, fn0, ln, tpa);
    }
    asm volatile("" : "+v"(ln));
    pp_epi_store<EPI, HI>(p, out, fm0, fn0, ln);
  }
#undef PP_ISSUE
#undef PP_PROLOGUE
#undef PP_MFMA2
#undef PP_READ_A
#undef PP_READ_A1
#undef PP_READ_B
#undef PP_SYNC_L
#undef PP_SYNC_L_REST
#undef STAMP_T0
#undef STAMP_ADD
#undef PP_SYNC_C
}

static int roll_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <bool TB, int EPI, bool HASB>
static void roll_launch0(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 131072 + 8 * 4096;
  auto kern = gemm_bf16_roll_kernel<TB, EPI, HASB>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  const int nitems = (p.N / 256) * (p.M / 256);
  int cus = roll_num_cus() - p.spare_cus;
  if (cus < 8) cus = 8;
  const int grid = nitems < cus ? nitems : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, p);
}

// Which calls the rolling kernel takes (everything else that asks for variant 10 runs on the ping-pong kernel, variant 8):
// bf16 output, A k-contiguous, whole 256 x 256 tiles, an even number >= 4 of k-tiles, no item queue, and
//   X W^T + b (forward): no epilogue, GELU without a second stream, tanh;   dY W (input gradient, no bias): no epilogue
bool uc2_gemm_roll_supported(const GemmArgs& p, int trans_a, int trans_b) {
  if (trans_a || p.c_f32 || p.queue || p.split_k != 1 || (p.diag && !UC2_ROLL_DIAG)) return false;
  const int kt = p.K / 64;
  if ((p.K % 64) || (kt & 1) || kt < 4 || (p.M % 256) || (p.N % 256)) return false;
  if (!trans_b) return p.bias != nullptr && (p.epi == EPI_NONE || (p.epi == EPI_GELU && !p.aux_out) || p.epi == EPI_TANH);
  return p.bias == nullptr && p.epi == EPI_NONE;
}

void uc2_gemm_roll_launch(const GemmArgs& p, int trans_b, hipStream_t st) {
  if (!trans_b) {
    if (p.epi == EPI_GELU) roll_launch0<false, EPI_GELU_NOAUX, true>(p, st);
    else if (p.epi == EPI_TANH) roll_launch0<false, EPI_TANH, true>(p, st);
    else roll_launch0<false, EPI_NONE, true>(p, st);
  } else {
    roll_launch0<true, EPI_NONE, false>(p, st);
  }
}
