// MFMA attention for bf16 (gfx950): one workgroup per (batch, head), one 64-lane wave per 32 rows of
// the (padded) sequence, v_mfma_f32_32x32x16_bf16 everywhere, fp32 softmax statistics.
//
// Layout trick (CDNA4): every product is arranged so that the softmax axis never has to cross lanes
// through LDS and no operand needs a transposed copy:
//   * scores are computed "swapped", S^T[key][query] = K Q^T, so a lane owns one query column and the
//     key axis runs over its 16 accumulator registers (+ the partner lane 32 away): max/sum are
//     register reductions plus ONE __shfl_xor(…, 32);
//   * a 32x32 fp32 accumulator tile, converted to bf16 in place, IS the B operand of the next MFMA that
//     contracts over its row index (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS);
//     the k order inside such a step is permuted (element j of lane half h = row 16s + 8(j>>2) + 4h +
//     (j&3)), and the matching A operand (V^T, K^T, dO^T, Q^T) is read with that same permutation
//     straight out of the row-major LDS tile by ds_read_b64_tr_b16 (hardware transpose);
//   * the backward computes S and dP in both orientations (7 small MFMA products instead of 5) so that
//     dQ, dK and dV are all "contract over the accumulator's row index" products: no LDS round trip of
//     P or dS, no atomics, each wave owns its 32 query rows (dQ) and its 32 key rows (dK, dV).
// K, V (and Q, dO in the backward) are staged once per workgroup into LDS as row-major tiles with
// 16 bytes of padding per row (conflict-free ds_read_b128 fragments).  Attention is ~2 % of the
// layer's flops at L = 96, so the goal here is to be near the HBM time of reading QKV once.
//
// Checked on the device against the fp32-math kernels of attention.hip (tests/test_gpu_kernels.py).
#include "common.h"

#define AM_MAXW 5       // up to 160 positions
#define AM_LOG2E 1.4426950408889634f
// QKV layouts (runtime, `ilv`): 0 = [B L][3][nh][D] -- the fused projection's natural column order q | k | v, head h at column h D;
// 1 = head-interleaved [B L][nh][3][D] -- one 384-byte segment per (token, head) at D = 64 instead of three 128-byte segments 1536
// bytes apart (ops.BertLayerFn runs the QKV GEMM on a row-permuted copy of the weights; round 4: forward 144 -> 128 us at 12 288
// heads).  AM_HS = offset of head h inside a row / h, AM_WS = offset of k (and half that of v) from q.
#define AM_HS(D, H) (ilv ? 3 * (D) : (D))
#define AM_WS(D, H) (ilv ? (D) : (H))
#ifndef AM_BWD_STRIDED
#define AM_BWD_STRIDED 0     // backward, static partition: 1 = a workgroup walks heads chunk + j * (number of workgroups) instead of a contiguous range (measured: 18.55-18.65 ms per step against 18.22-18.60, same box -- no gain)
#endif
#ifndef AM_LINE_STORES
#define AM_LINE_STORES 0     // backward: 1 = dQ / dK / dV leave as whole 128-byte lines through LDS (0: 16-byte pieces of 32 rows per
                             // instruction).  Measured (round 4, same box, bit-identical): 295.9 / 577.8 us with, 299.5 / 579.3 without at
                             // 12 288 / 24 576 heads -- the store granularity is not what bounds the kernel; off
#endif
#ifndef AM_LEAN
#define AM_LEAN 1            // L == 32 NW: the kernels' lean load / store addressing (tile_lane_offset); 0 = the general path for every L
#endif
#ifndef AM_BWD_EARLY
#define AM_BWD_EARLY 2       // with AM_BWD_PREFETCH == 2: how many of the four tiles (Q, K, dO, O) are fetched early
#endif
#ifndef AM_BWD_PREFETCH
#define AM_BWD_PREFETCH 0    // backward: 1 = fetch the next head into registers before the main loop (costs ~80 VGPRs there: spills),
                             //           2 = fetch AM_BWD_EARLY of its four tiles at the start of the dQ phase, where the main loop's accumulators are dead
                             //           (round 4; measured at the bench size, same box: 289-291 us with 1, 2 or 3 early tiles against 291 without
                             //           and 288.5 for round 3's kernel -- the backward is not bound by its load latency either; off)
#endif

__device__ __forceinline__ bf16x8 zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// 8 consecutive bf16 of one row (row-major tile, RS bytes per row): fragment of a k-contiguous operand
__device__ __forceinline__ bf16x8 ld_row(const char* tile, int RS, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * RS + chunk * 16);
}
// transposed fragment: element j of lane half h = tile[kbase + 8*(j>>2) + 4*h + (j&3)][cbase + (lane&31)]
__device__ __forceinline__ bf16x8 ld_tr(const char* tile, int RS, int kbase, int cbase, int lane) {
  const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
  const int off = (kbase + 4 * h + q) * RS + (cbase + 16 * (G & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) short4v* lds_p;
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + off));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + off + 8 * RS));
  bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  return bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {   // registers 8s..8s+7 -> bf16x8 (s compile-time)
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)a[8 * s + j];
  return r;
}
// Column sums across the 32 lanes of a lane half, 32 partial sums per lane (index v): lane (c, h) returns the total
// of v = c (transposing butterfly, 31 exchanges; same helper as in gemm_pp.hip)
// (colsum_butterfly32: common.h)
// bias-gradient contribution of one wave's 32 rows x D columns held as DB transposed accumulators (register 4g+e of
// block db = column 32db + 8g + 4h + e of row r0): dst[col] += sum over the wave's valid rows (D = 64: one atomic
// per lane; D = 32: the 16 values per lane are padded to the 32-value butterfly)
template <int DB>
__device__ __forceinline__ void acc_colsum_atomic(const f32x16 (&a)[DB], bool row_ok, float* __restrict__ dst, int lane) {
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0.f;
  if (row_ok) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[16 * db + r] = a[db][r];
  }
  const float tot = colsum_butterfly32(v, lane);
  const int c = lane & 31, h = lane >> 5, db = c >> 4, r = c & 15;
  if (db < DB) atomicAdd(dst + 32 * db + 8 * (r >> 2) + 4 * h + (r & 3), tot);
}
// Store one transposed 32x32 accumulator block (lane = output row, register 4g+e = column 8g + 4h + e) as bf16: the two
// lane halves exchange 4-column groups (v_permlane32_swap) so every lane owns 8 consecutive columns -- two 16-byte stores
// (columns 8h.. and 16+8h..) instead of four 8-byte ones.  Must be called by the whole wave; `ok` masks the store.
// fp8 mode (uc2_attn_fwd_q / uc2_attn_bwd_q): the kernel also writes the e4m3 copy of its output that the next GEMM reads (ctx -> the
// output projection, dqkv -> the q|k|v input gradient), with delayed scaling and the three cell groups of uc2_fp8_quant_delayed.
// q == NULL: off.  The copy is taken from the bf16-ROUNDED values, like the stand-alone quantisation pass it replaces.
struct AttnQ { uint8_t* q; const unsigned* prev; unsigned* next; unsigned* clear; float* scale_out; };
// e4m3 bytes of the lane's eight values (+ their maximum)
__device__ __forceinline__ uint2 attn_q8(const bf16x8& o, float qs, float& qmax, bool ok) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)o[e];
  if (ok) {
#pragma unroll
    for (int e = 0; e < 8; ++e) qmax = fmaxf(qmax, fabsf(f[e]));
  }
  return make_uint2(fp8_pack4_sat(f[0] * qs, f[1] * qs, f[2] * qs, f[3] * qs), fp8_pack4_sat(f[4] * qs, f[5] * qs, f[6] * qs, f[7] * qs));
}
// A lane owns columns 8h .. 8h+7 (piece 0) and 16+8h .. 16+8h+7 (piece 1) of a 32-column block of its row; the two lanes of a row
// (h = 0 / 1) trade one piece each (v_permlane32_swap: piece 0 of the upper half <-> piece 1 of the lower half), so that each stores
// ONE 16-byte run -- columns 0..15 (h = 0) or 16..31 (h = 1) -- instead of two 8-byte ones: half the partial-line write requests.
__device__ __forceinline__ void attn_q8_store(uint2 k0, uint2 k1, uint8_t* __restrict__ row32, int h, bool ok) {
  const auto sx = __builtin_amdgcn_permlane32_swap(k0.x, k1.x, false, false);
  const auto sy = __builtin_amdgcn_permlane32_swap(k0.y, k1.y, false, false);
  if (ok) *reinterpret_cast<uint4*>(row32 + 16 * h) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
}
// the workgroup's max |value| -> one atomic into the role's next cell group (wm: NW floats of LDS nobody else uses any more)
template <int NW>
__device__ __forceinline__ void attn_q_amax(const AttnQ& aq, float qmax, float* wm, int tid) {
  qmax = wave_max(qmax);
  __syncthreads();
  if ((tid & 63) == 0) wm[tid >> 6] = qmax;
  __syncthreads();
  if (tid == 0) {
    float m = wm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fmaxf(m, wm[i]);
    amax_cell_raise(aq.next + (blockIdx.x & (UC2_AMAX_CELLS - 1)), m);
  }
}
__device__ __forceinline__ void store_acc_block(bf16* __restrict__ row_ptr, const f32x16& a, float scale, int h, bool ok,
                                                uint8_t* __restrict__ q_ptr = nullptr, float qs = 0.f, float* qmax = nullptr) {
  uint2 qk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float fx = a[8 * k + e] * scale, fy = a[8 * k + 4 + e] * scale;
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fx), __float_as_uint(fy), false, false);
      o[e] = (bf16)__uint_as_float(sw[0]);
      o[4 + e] = (bf16)__uint_as_float(sw[1]);
    }
    if (ok) *reinterpret_cast<bf16x8*>(row_ptr + 16 * k + 8 * h) = o;
    if (q_ptr) qk[k] = attn_q8(o, qs, *qmax, ok);                       // (uniform branch)
  }
  if (q_ptr) attn_q8_store(qk[0], qk[1], q_ptr, h, ok);
}
// store_acc_block for a workgroup-uniform base + one per-lane byte offset (row and lane half; L == Lp: no mask)
__device__ __forceinline__ void store_acc_block_s(char* __restrict__ sbase, uint32_t voff, const f32x16& a,
                                                  uint8_t* __restrict__ qbase = nullptr, float qs = 0.f, float* qmax = nullptr) {
  uint2 qk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * k + e]), __float_as_uint(a[8 * k + 4 + e]), false, false);
      o[e] = (bf16)__uint_as_float(sw[0]);
      o[4 + e] = (bf16)__uint_as_float(sw[1]);
    }
    *reinterpret_cast<bf16x8*>(sbase + voff + 32 * k) = o;
    if (qbase) qk[k] = attn_q8(o, qs, *qmax, true);
  }
  // (voff = row offset + 16 h bytes of bf16 = element offset 8 h: the e4m3 row starts at (voff >> 1) - 8 h, the lane's run at + 16 h)
  if (qbase) {
    const auto sx = __builtin_amdgcn_permlane32_swap(qk[0].x, qk[1].x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(qk[0].y, qk[1].y, false, false);
    const uint32_t hb = (voff & 16u) >> 1;                              // 8 h
    *reinterpret_cast<uint4*>(qbase + (voff >> 1) - hb + 2 * hb) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  }
}
// The same block set stored as WHOLE 128-byte lines (round 4): store_acc_block writes 16-byte pieces of 32 different rows per
// wave-instruction, i.e. 64 partial-line requests -- the write pattern the GEMM epilogue got rid of in round 2 (+7..19 % there).
// A wave's [32 rows x D] bf16 output goes through a wave-private LDS image (32 rows x 2 D bytes; 16-byte chunk q of row r at
// r 2D + ((q ^ (r & (2D/16 - 1))) << 4), 8-byte pieces written with a 2-way bank conflict, read back conflict-free) and leaves in
// the line layout: lane l owns chunk l % (D/8) of row l / (D/8) + (512/D) it -- D/8 lanes cover one row segment of 2 D bytes.
// LDS executes a wave's operations in order, so the buffer is reused back to back without barriers.
template <int D, int DB>
__device__ __forceinline__ void store_acc_lines(bf16* __restrict__ gbase, size_t ld, int row0, int L, const f32x16 (&a)[DB],
                                                char* __restrict__ tbuf, int lane) {
  constexpr int RB = 2 * D, CPR = D / 8, RPI = 64 / CPR, NIT = 32 / RPI;      // bytes per row, chunks per row, rows per instruction
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)a[db][4 * g + e];
      const int q = 4 * db + g;                                               // 16-byte chunk of the row; this lane's half = h
      *reinterpret_cast<bf16x4*>(tbuf + c * RB + ((q ^ (c & (CPR - 1))) << 4) + 8 * h) = v;
    }
  const int lr = lane / CPR, lc = lane % CPR;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = RPI * it + lr;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(tbuf + r * RB + ((lc ^ (r & (CPR - 1))) << 4));
    if (row0 + r < L) *reinterpret_cast<bf16x8*>(gbase + (size_t)(row0 + r) * ld + 8 * lc) = v;
  }
}
// row of accumulator register i for lane half h
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

template <int D>
__device__ __forceinline__ void stage_tile(char* dst, const bf16* __restrict__ src, int ld, int L, int Lp, int tid,
                                           int nthr) {
  constexpr int RS = D * 2 + 16, CPR = D / 8;
  for (int c = tid; c < Lp * CPR; c += nthr) {
    const int row = c / CPR, c8 = c - row * CPR;
    bf16x8 v = zero8();
    if (row < L) v = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld + c8 * 8);
    *reinterpret_cast<bf16x8*>(dst + row * RS + c8 * 16) = v;
  }
}

// Register-staged variant (backward): a workgroup walks over several heads and fetches the NEXT head's tiles
// into registers while it computes the current one, so the global-load latency (67 % of the wave time of the
// one-head-per-workgroup version, rocprofv3 SQ_WAIT_ANY) is hidden behind the MFMA/softmax work.
// Every thread owns NCH = Lp*CPR/nthr (= 4 for D = 64, 2 for D = 32) 16-byte chunks of each tile.
template <int D, int NCH>
__device__ __forceinline__ void load_tile_regs(bf16x8 (&r)[NCH], const bf16* __restrict__ src, int ld, int L, int tid, int nthr) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * nthr, row = c / CPR, c8 = c - row * CPR;
    r[i] = zero8();
    if (row < L) r[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld + c8 * 8);
  }
}
template <int D, int NCH>
__device__ __forceinline__ void store_tile_regs(char* dst, const bf16x8 (&r)[NCH], int tid, int nthr) {
  constexpr int RS = D * 2 + 16, CPR = D / 8;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * nthr, row = c / CPR, c8 = c - row * CPR;
    *reinterpret_cast<bf16x8*>(dst + row * RS + c8 * 16) = r[i];
  }
}

// Lean forms for L == Lp (no row of the padded tile is past the sequence: L = 96 = 3 x 32, the bench geometry) -- round 4.
// load_tile_regs computes, per 16-byte load, a signed division of the chunk index, a 64-bit multiply-add for the row and a bounds
// branch with four zero moves: ~18 vector instructions per load, 20 loads per head in the backward (a fifth of its vector
// instructions; both kernels are bound by instruction issue, profiles/r04_experiments.md section 3b).  Here the workgroup-uniform
// part of the address (tensor, batch, head, chunk step i) stays in scalar registers and the per-lane part is ONE 32-bit byte offset
// computed once per kernel (global_load saddr + voffset form): thread t owns chunk t % CPR of rows t / CPR + i NTHR / CPR.
template <int D, int NTHR>
__device__ __forceinline__ uint32_t tile_lane_offset(int ld, unsigned tid) {      // byte offset of this thread's chunk in rows of `ld` elements
  constexpr unsigned CPR = D / 8;
  return ((tid / CPR) * (unsigned)ld + (tid % CPR) * 8u) * 2u;
}
template <int D, int NCH, int NTHR>
__device__ __forceinline__ void load_tile_lean(bf16x8 (&r)[NCH], const char* __restrict__ sbase, uint32_t voff, int ld) {
  constexpr int CPR = D / 8;
  static_assert(NTHR % CPR == 0, "a thread keeps its column chunk across steps");
  const size_t step = (size_t)(NTHR / CPR) * (size_t)ld * 2u;                    // uniform
#pragma unroll
  for (int i = 0; i < NCH; ++i) r[i] = *reinterpret_cast<const bf16x8*>(sbase + i * step + voff);
}
template <int D, int NCH, int NTHR>
__device__ __forceinline__ void store_tile_lean(char* dst, const bf16x8 (&r)[NCH], unsigned tid) {
  constexpr int RS = D * 2 + 16, CPR = D / 8;
  char* p = dst + (tid / CPR) * RS + (tid % CPR) * 16;
#pragma unroll
  for (int i = 0; i < NCH; ++i) *reinterpret_cast<bf16x8*>(p + i * (NTHR / CPR) * RS) = r[i];
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// (Round 4 measured a persistent form of this kernel -- a resident round of workgroups walking heads with the next head's Q / K / V
//  prefetched into registers under the current head's compute -- at the bench size: 139.0-139.5 us against 138.2 for this one
//  (scratch/ab_attn_libs.py, same box).  The forward is not bound by the latency of its per-workgroup chain; what moved it in
//  round 3 was the access pattern (head-major buffers: 147 -> 121 us).  Not kept.)
template <int D, int NW, bool FULL, bool DROP, bool Q = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd_mfma_kernel(int L, int nh, const bf16* __restrict__ qkv,
                                                                const float* __restrict__ mask, float scale,
                                                                uint32_t thresh, float keep_scale,
                                                                const uint64_t* __restrict__ seed_ptr,
                                                                uint64_t seed_imm, bf16* __restrict__ ctx,
                                                                float* __restrict__ lse, int ilv, AttnQ aq) {
  constexpr int KS = D / 16, DB = D / 32, RS = D * 2 + 16, Lp = NW * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const float qs = Q ? fp8_delayed_scale(aq.prev) : 0.f;
  float qmax = 0.f;
  if (Q && blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { aq.clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *aq.scale_out = qs; }
  char* Ks = smem;
  char* Vs = smem + Lp * RS;
  float* Ms = reinterpret_cast<float*>(smem + 2 * Lp * RS);
  uint32_t* Hk = reinterpret_cast<uint32_t*>(Ms + Lp);          // per-key dropout hashes
  const int bh = blockIdx.x, b = bh / nh, head = bh - b * nh;
  const int H = nh * D, ld = 3 * H;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const bf16* base = qkv + (size_t)b * L * ld + head * AM_HS(D, H);
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int q = 32 * w + c;
  // this lane's Q fragments: issued together with the K / V tile loads (after the barrier below they were a second, fully
  // exposed trip to memory per workgroup)
  bf16x8 qf[KS];
  if (FULL) {                                        // L == Lp: no bounds, scalar bases + one lane offset (load_tile_lean)
    const char* qb = reinterpret_cast<const char*>(base);
    const uint32_t qoff = ((unsigned)q * (unsigned)ld + 8u * (unsigned)h) * 2u;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qb + qoff + 32 * s);
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[s] = (q < L) ? *reinterpret_cast<const bf16x8*>(base + (size_t)q * ld + 16 * s + 8 * h) : zero8();
  }
  {   // K and V tiles: all 16-byte loads of the thread in flight together, then the LDS stores (a load -> wait -> store
      // loop costs eight dependent memory round trips per workgroup)
    constexpr int NCH = Lp * (D / 8) / (NW * 64);
    bf16x8 rk[NCH], rv[NCH];
    if (FULL) {
      const uint32_t toff = tile_lane_offset<D, NW * 64>(ld, tid);
      load_tile_lean<D, NCH, NW * 64>(rk, reinterpret_cast<const char*>(base + AM_WS(D, H)), toff, ld);
      load_tile_lean<D, NCH, NW * 64>(rv, reinterpret_cast<const char*>(base + 2 * AM_WS(D, H)), toff, ld);
    } else {
      load_tile_regs<D, NCH>(rk, base + AM_WS(D, H), ld, L, tid, NW * 64);
      load_tile_regs<D, NCH>(rv, base + 2 * AM_WS(D, H), ld, L, tid, NW * 64);
    }
    for (int k = tid; k < Lp; k += NW * 64) {
      Ms[k] = (FULL || k < L) ? (mask ? mask[(size_t)b * L + k] * AM_LOG2E : 0.f) : -1e30f;        // base-2 domain: p = exp2(s c2 + m' - max')
      Hk[k] = attn_line_hash(seed, bh, k, UC2_ATTN_SALT_K);
    }
    if (FULL) {
      store_tile_lean<D, NCH, NW * 64>(Ks, rk, tid);
      store_tile_lean<D, NCH, NW * 64>(Vs, rv, tid);
    } else {
      store_tile_regs<D, NCH>(Ks, rk, tid, NW * 64);
      store_tile_regs<D, NCH>(Vs, rv, tid, NW * 64);
    }
  }
  __syncthreads();

  f32x16 sc[NW];
  const float c2 = scale * AM_LOG2E;                   // scores are kept times log2(e): one v_exp_f32 per element, no multiply before it
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NW; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Ks, RS, 32 * kb + c, 2 * s + h), qf[s], sc[kb], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 mv = *reinterpret_cast<const float4*>(Ms + 32 * kb + 8 * g + 4 * h);
      const float m4[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(sc[kb][4 * g + e], c2, m4[e]);
        sc[kb][4 * g + e] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
  const uint32_t hq = attn_line_hash(seed, bh, q, UC2_ATTN_SALT_Q);
#pragma unroll
  for (int kb = 0; kb < NW; ++kb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bool kp[4] = {true, true, true, true};
      if (DROP) {                                      // (DROP = thresh != 0, a launch-time constant: as a run-time test hipcc branches per element)
        const uint4 hk4 = *reinterpret_cast<const uint4*>(Hk + 32 * kb + 8 * g + 4 * h);
        kp[0] = attn_keep(hq, hk4.x, thresh); kp[1] = attn_keep(hq, hk4.y, thresh);
        kp[2] = attn_keep(hq, hk4.z, thresh); kp[3] = attn_keep(hq, hk4.w, thresh);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        const float p = __builtin_amdgcn_exp2f(sc[kb][i] - mx);
        sum += p;
        sc[kb][i] = kp[e] ? p : 0.f;                     // (the 1 / (1 - p_drop) of the kept entries is folded into the output scale)
        // (select in fp32, then convert pairs: left alone, hipcc converts first and selects on the 16-bit halves -- shift, two
        //  selects and a v_perm_b32 per pair instead of two selects)
        asm volatile("" : "+v"(sc[kb][i]));
      }
    }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;

  bf16* out = ctx + ((size_t)b * L + q) * H + head * D;
  uint8_t* out8 = Q ? aq.q + ((size_t)b * L + q) * H + head * D : nullptr;
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NW; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Vs, RS, 32 * kb + 16 * s, 32 * db, lane), pack8(sc[kb], s), o,
                                                    0, 0, 0);
    store_acc_block(out + 32 * db, o, DROP ? inv * keep_scale : inv, h, FULL || q < L, Q ? out8 + 32 * db : nullptr, qs, &qmax);
  }
  if (lse && (FULL || q < L) && h == 0) lse[(size_t)bh * L + q] = (mx + __builtin_amdgcn_logf(sum)) * 0.69314718056f;     // natural-log lse
  if (Q) attn_q_amax<NW>(aq, qmax, Ms, tid);           // (Ms: the mask row, dead after the scores)
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// One orientation only (key on the lane): wave w owns keys 32w .. 32w+31 of the head.  Per 32-query block it forms
// S = Q K^T and dP = dO V^T with the key on the lane (K and V rows of the wave are its B fragments, held in registers),
// so P and dS come out as accumulator tiles that ARE the B operands of dV^T += dO^T P and dK^T += Q^T dS; only dS
// crosses LDS, once, as a bf16 [key][query] image that the dQ phase reads back with the transposing read
// (dQ^T = K^T dS^T, each wave its own 32 queries over all keys).  Against the two-orientation version of round 1
// (S, dP and the softmax / dropout arithmetic evaluated twice): 60 instead of 84 MFMAs and about half the VALU work per
// wave and head, and no V tile in LDS.
// (launch bound: 2 waves per SIMD = 256 registers per lane.  Unbounded, hipcc took 254 VGPRs + 96 AGPRs, which admits ONE
// wave per SIMD: one 3-wave workgroup per CU with a SIMD idle -- rocprofv3 showed wave lifetimes of half the kernel.)
template <int D, int NW, bool FULL, bool DROP, bool Q = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_mfma_kernel(int L, int nh, const bf16* __restrict__ qkv,
                                                                const float* __restrict__ mask, float scale,
                                                                uint32_t thresh, float keep_scale,
                                                                const uint64_t* __restrict__ seed_ptr,
                                                                uint64_t seed_imm, const bf16* __restrict__ ctx,
                                                                const bf16* __restrict__ dctx,
                                                                const float* __restrict__ lse,
                                                                bf16* __restrict__ dqkv, float* __restrict__ dbias, int nbh, int hpw,
                                                                int* __restrict__ queue, int ilv, AttnQ aq) {
  constexpr int KS = D / 16, DB = D / 32, RS = D * 2 + 16, Lp = NW * 32;
  const float qs = Q ? fp8_delayed_scale(aq.prev) : 0.f;
  float qmax = 0.f;
  if (Q && blockIdx.x == 0 && threadIdx.x < UC2_AMAX_CELLS) { aq.clear[threadIdx.x] = 0u; if (threadIdx.x == 0) *aq.scale_out = qs; }
  constexpr int RSD = Lp * 2 + 16;                     // dS^T image: [key][query] bf16
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Ks = smem + Lp * RS;
  char* Gs = smem + 2 * Lp * RS;                       // dO
  char* Ss = smem + 3 * Lp * RS;                       // dS^T
  float* Ms = reinterpret_cast<float*>(Ss + Lp * RSD);
  float* Ls = Ms + Lp;
  float* Ds = Ls + Lp;
  uint32_t* Hq = reinterpret_cast<uint32_t*>(Ds + Lp);          // per-query / per-key dropout hashes
  uint32_t* Hk = Hq + Lp;
  int* Tk = reinterpret_cast<int*>(Hk + Lp);           // [4] (queue) index of the workgroup's next chunk of heads
  float* Cs = reinterpret_cast<float*>(Tk + 4);        // [3H] column sums of dQ | dK | dV over this workgroup's heads (dbias only)
  const int H = nh * D, ld = 3 * H;
  const uint64_t seed = seed_imm + (seed_ptr ? *seed_ptr : 0ull);
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int r0 = 32 * w + c;                           // this lane's key (main loop) and query (dQ phase)
  constexpr int NTHR = NW * 64, CPR = D / 8, NCH = Lp * CPR / NTHR;
  // (FULL) the per-lane byte offsets of the lean addressing: this thread's chunk in a q|k|v tile / a ctx tile, this lane's row
  const uint32_t toff_qkv = tile_lane_offset<D, NTHR>(ld, tid), toff_ctx = tile_lane_offset<D, NTHR>(H, tid);
  const uint32_t roff_qkv = ((unsigned)r0 * (unsigned)ld + 8u * (unsigned)h) * 2u;
  if (dbias) {                                         // (the head loop's first barrier orders this before any accumulation)
    for (int i = tid; i < 3 * H; i += NTHR) Cs[i] = 0.f;
  }
  // Chunks of `hpw` consecutive heads.  Static partition (queue == nullptr): one chunk per workgroup, the grid is one resident
  // round.  With a caller-owned queue (two zeroed ints; uc2_attn_bwd_queued) a workgroup's first chunk is blockIdx.x and every
  // later one comes from an atomic counter: when another kernel (an overlapped all-reduce) holds CUs, the workgroups that do run
  // take over the work of those that are placed late instead of the kernel waiting a second round for them.  The ticket is
  // fetched at the start of a chunk and read at its end; the last workgroup to leave zeroes the queue.
  const int nchunk = (nbh + hpw - 1) / hpw;
  int chunk = blockIdx.x;
  int ticket = 0;
  for (;;) {
  if (chunk >= nchunk) break;
  if (queue && tid == 0) ticket = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (AM_BWD_STRIDED, static partition only: the workgroup's heads are chunk, chunk + nchunk, chunk + 2 nchunk, ... instead of hpw
  //  consecutive ones -- at any moment the resident workgroups then work on CONSECUTIVE (batch, head) pairs, i.e. all heads of a
  //  token row are read at about the same time by neighbouring workgroups)
  const bool strided = AM_BWD_STRIDED && !queue;
  const int hstep = strided ? nchunk : 1;
  const int bh_end = strided ? nbh : min(nbh, (chunk + 1) * hpw);
  int bh = strided ? chunk : chunk * hpw;
  // ---- fetch of one head into registers: Q, K, dO, O tiles (NCH chunks each), this lane's V fragments, mask / lse of row tid
  bf16x8 rq[NCH], rk[NCH], rg[NCH], ro[NCH], rvf[KS];
  float rmask = 0.f, rlse = 0.f;
  // (the thread index goes through an opaque copy: hipcc otherwise hoists the ~20 per-lane 64-bit load addresses out of the head
  //  loop, where they live -- and spill -- across the main loop)
  auto fetch_tiles = [&](int hd, int lo, int hi) __attribute__((always_inline)) {      // tiles lo .. hi-1 of (0 Q, 1 K, 2 dO, 3 O)
    const int fb = hd / nh, fh = hd - fb * nh;
    const bf16* fbase = qkv + (size_t)fb * L * ld + fh * AM_HS(D, H);
    if (FULL) {
      if (lo <= 0 && 0 < hi) load_tile_lean<D, NCH, NTHR>(rq, reinterpret_cast<const char*>(fbase), toff_qkv, ld);
      if (lo <= 1 && 1 < hi) load_tile_lean<D, NCH, NTHR>(rk, reinterpret_cast<const char*>(fbase + AM_WS(D, H)), toff_qkv, ld);
      if (lo <= 2 && 2 < hi) load_tile_lean<D, NCH, NTHR>(rg, reinterpret_cast<const char*>(dctx + (size_t)fb * L * H + fh * D), toff_ctx, H);
      if (lo <= 3 && 3 < hi) load_tile_lean<D, NCH, NTHR>(ro, reinterpret_cast<const char*>(ctx + (size_t)fb * L * H + fh * D), toff_ctx, H);
      return;
    }
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (lo <= 0 && 0 < hi) load_tile_regs<D, NCH>(rq, fbase, ld, L, t_, NTHR);
    if (lo <= 1 && 1 < hi) load_tile_regs<D, NCH>(rk, fbase + AM_WS(D, H), ld, L, t_, NTHR);
    if (lo <= 2 && 2 < hi) load_tile_regs<D, NCH>(rg, dctx + (size_t)fb * L * H + fh * D, H, L, t_, NTHR);
    if (lo <= 3 && 3 < hi) load_tile_regs<D, NCH>(ro, ctx + (size_t)fb * L * H + fh * D, H, L, t_, NTHR);
  };
  auto fetch_rest = [&](int hd) __attribute__((always_inline)) {                       // V fragments, mask / lse of row tid
    const int fb = hd / nh, fh = hd - fb * nh;
    const bf16* fbase = qkv + (size_t)fb * L * ld + fh * AM_HS(D, H);
    if (FULL) {
      const char* vb = reinterpret_cast<const char*>(fbase + 2 * AM_WS(D, H));
#pragma unroll
      for (int s = 0; s < KS; ++s) rvf[s] = *reinterpret_cast<const bf16x8*>(vb + roff_qkv + 32 * s);
    } else {
      int r0_ = r0;
      asm volatile("" : "+v"(r0_));
#pragma unroll
      for (int s = 0; s < KS; ++s)
        rvf[s] = (r0 < L) ? *reinterpret_cast<const bf16x8*>(fbase + 2 * AM_WS(D, H) + (size_t)r0_ * ld + 16 * s + 8 * h) : zero8();
    }
    if (tid < Lp && tid < L) {
      rmask = mask ? mask[(size_t)fb * L + tid] : 0.f;
      rlse = lse[(size_t)hd * L + tid];
    }
  };
  auto fetch = [&](int hd) __attribute__((always_inline)) { fetch_tiles(hd, 0, 4); fetch_rest(hd); };
  if (AM_BWD_PREFETCH == 1 && bh < bh_end) fetch(bh);
  if (AM_BWD_PREFETCH == 2 && bh < bh_end) fetch_tiles(bh, 0, AM_BWD_EARLY);      // (only the early tiles are carried around the loop)
  for (; bh < bh_end; bh += hstep) {
  const int b = bh / nh, head = bh - b * nh;
  bf16* dbase = dqkv + (size_t)b * L * ld + head * AM_HS(D, H);
  uint8_t* qbase = Q ? aq.q + (size_t)b * L * ld + head * AM_HS(D, H) : nullptr;        // (the e4m3 copy of dqkv, same element offsets)
  if (!AM_BWD_PREFETCH) fetch(bh);                     // no register prefetch: the CU's other workgroup covers the latency
  if (AM_BWD_PREFETCH == 2) { fetch_tiles(bh, AM_BWD_EARLY, 4); fetch_rest(bh); }     // what the dQ phase of the previous head did not fetch
  // ---- registers -> LDS (the previous head's readers are past the barrier at the end of the loop body)
  if (FULL) {
    store_tile_lean<D, NCH, NTHR>(Qs, rq, tid);
    store_tile_lean<D, NCH, NTHR>(Ks, rk, tid);
    store_tile_lean<D, NCH, NTHR>(Gs, rg, tid);
  } else {
    store_tile_regs<D, NCH>(Qs, rq, tid, NTHR);
    store_tile_regs<D, NCH>(Ks, rk, tid, NTHR);
    store_tile_regs<D, NCH>(Gs, rg, tid, NTHR);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {                      // delta[row] = sum_d dO*O: 8 values per chunk, CPR consecutive lanes per row
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)rg[i][e] * (float)ro[i][e];
#pragma unroll
    for (int o = 1; o < CPR; o <<= 1) dl += __shfl_xor(dl, o);
    // chunk tid + i NTHR = column chunk tid % CPR of row tid / CPR + i NTHR / CPR (NTHR is a multiple of CPR)
    if (((unsigned)tid & (CPR - 1)) == 0) Ds[(unsigned)tid / CPR + i * (NTHR / CPR)] = dl * scale;       // delta, pre-multiplied by the softmax scale
  }
  if (tid < Lp) {
    Ms[tid] = (tid < L) ? rmask * AM_LOG2E : -1e30f;          // base-2 domain, like the forward
    Ls[tid] = (tid < L) ? rlse * AM_LOG2E : 0.f;
    Hq[tid] = attn_line_hash(seed, bh, tid, UC2_ATTN_SALT_Q);
    Hk[tid] = attn_line_hash(seed, bh, tid, UC2_ATTN_SALT_K);
  }
  bf16x8 vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) vf[s] = rvf[s];
  __syncthreads();
  if (AM_BWD_PREFETCH == 1 && bh + hstep < bh_end) fetch(bh + hstep);        // in flight during the compute below

  // ---------------- rows = query, cols = key (this wave's 32 keys) -> dK, dV, and dS^T into LDS ----------------
  f32x16 dk[DB], dv[DB];
  {
    bf16x8 kf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kf[s] = ld_row(Ks, RS, r0, 2 * s + h);
    const float mk = Ms[r0];
    const uint32_t hk2 = Hk[r0];
    const float c2 = scale * AM_LOG2E, ks_scale = keep_scale * scale;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
#pragma unroll
    for (int qb = 0; qb < NW; ++qb) {
      f32x16 sa, pa;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; pa[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Qs, RS, 32 * qb + c, 2 * s + h), kf[s], sa, 0, 0, 0);
        pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Gs, RS, 32 * qb + c, 2 * s + h), vf[s], pa, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 lv = *reinterpret_cast<const float4*>(Ls + 32 * qb + 8 * g + 4 * h);
        const float4 dv4 = *reinterpret_cast<const float4*>(Ds + 32 * qb + 8 * g + 4 * h);
        const float l4[4] = {lv.x, lv.y, lv.z, lv.w}, d4[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
        const uint4 hq4 = *reinterpret_cast<const uint4*>(Hq + 32 * qb + 8 * g + 4 * h);
        const uint32_t hqv[4] = {hq4.x, hq4.y, hq4.z, hq4.w};
        bf16x4 ds4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          // per element: fma, sub, exp2; two selects for the dropout multipliers (1/(1-p) and scale/(1-p)); mul, fma, mul
          const float p = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, mk) - l4[e]);
          float m1 = 1.0f, m2 = scale;
          if (DROP) {
            const bool keep = attn_keep(hqv[e], hk2, thresh);
            m1 = keep ? keep_scale : 0.f;
            m2 = keep ? ks_scale : 0.f;
          }
          const float dsv = p * fmaf(pa[i], m2, -d4[e]);      // dS = P (dP_dropped - delta) scale, delta pre-scaled
          pa[i] = DROP ? p * m1 : p;                 // dropped P
          sa[i] = dsv;                               // dS
          ds4[e] = (bf16)dsv;
        }
        // dS^T image: row = this lane's key, columns = queries 32 qb + 8 g + 4 h .. +3 (registers 4g .. 4g+3)
        *reinterpret_cast<bf16x4*>(Ss + r0 * RSD + (32 * qb + 8 * g + 4 * h) * 2) = ds4;
      }
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Gs, RS, 32 * qb + 16 * s, 32 * db, lane),
                                                           pack8(pa, s), dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Qs, RS, 32 * qb + 16 * s, 32 * db, lane),
                                                           pack8(sa, s), dk[db], 0, 0, 0);
        }
    }
    if (dbias) {                                                               // d(key bias), d(value bias)
      acc_colsum_atomic<DB>(dk, FULL || r0 < L, Cs + H + head * D, lane);
      acc_colsum_atomic<DB>(dv, FULL || r0 < L, Cs + 2 * H + head * D, lane);
    }
    if (!AM_LINE_STORES) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (FULL) {
          store_acc_block_s(reinterpret_cast<char*>(dbase + AM_WS(D, H) + 32 * db), roff_qkv, dk[db], Q ? qbase + AM_WS(D, H) + 32 * db : nullptr, qs, &qmax);
          store_acc_block_s(reinterpret_cast<char*>(dbase + 2 * AM_WS(D, H) + 32 * db), roff_qkv, dv[db], Q ? qbase + 2 * AM_WS(D, H) + 32 * db : nullptr, qs, &qmax);
        } else {
          store_acc_block(dbase + (size_t)r0 * ld + AM_WS(D, H) + 32 * db, dk[db], 1.0f, h, r0 < L,
                          Q ? qbase + (size_t)r0 * ld + AM_WS(D, H) + 32 * db : nullptr, qs, &qmax);
          store_acc_block(dbase + (size_t)r0 * ld + 2 * AM_WS(D, H) + 32 * db, dv[db], 1.0f, h, r0 < L,
                          Q ? qbase + (size_t)r0 * ld + 2 * AM_WS(D, H) + 32 * db : nullptr, qs, &qmax);
        }
      }
    }
  }
  __syncthreads();                                     // every wave's dS^T columns are in LDS (and Qs / Gs are dead)
  if (AM_LINE_STORES) {                                // dK, dV as whole lines through this wave's slice of the dead Q tile
    store_acc_lines<D, DB>(dbase + AM_WS(D, H), ld, 32 * w, L, dk, Qs + w * 32 * 2 * D, lane);
    store_acc_lines<D, DB>(dbase + 2 * AM_WS(D, H), ld, 32 * w, L, dv, Qs + w * 32 * 2 * D, lane);
  }
  // the next head's tiles: issued here, where dk / dv / the score tiles are dead (the prefetch then costs no register at the
  // kernel's peak); they are in flight during the dQ phase, its stores and the barrier, and the CU's other workgroup
  if (AM_BWD_PREFETCH == 2 && bh + hstep < bh_end) fetch_tiles(bh + hstep, 0, AM_BWD_EARLY);
  // ---------------- dQ^T = K^T dS^T for this wave's 32 queries, contracted over all keys ----------------
  {
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2 * NW; ++ks) {              // 16 keys per step; A (K^T) and B (dS^T) read with the same k permutation
      const bf16x8 bfrag = ld_tr(Ss, RSD, 16 * ks, 32 * w, lane);
#pragma unroll
      for (int db = 0; db < DB; ++db)
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Ks, RS, 16 * ks, 32 * db, lane), bfrag, dq[db], 0, 0, 0);
    }
    if (dbias) acc_colsum_atomic<DB>(dq, FULL || r0 < L, Cs + head * D, lane);          // d(query bias)
    if (AM_LINE_STORES) {
      store_acc_lines<D, DB>(dbase, ld, 32 * w, L, dq, Gs + w * 32 * 2 * D, lane);
    } else {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (FULL) store_acc_block_s(reinterpret_cast<char*>(dbase + 32 * db), roff_qkv, dq[db], Q ? qbase + 32 * db : nullptr, qs, &qmax);
        else store_acc_block(dbase + (size_t)r0 * ld + 32 * db, dq[db], 1.0f, h, r0 < L, Q ? qbase + (size_t)r0 * ld + 32 * db : nullptr, qs, &qmax);
      }
    }
  }
  __syncthreads();                                     // every wave is done with this head's LDS tiles
  }
  if (!queue) break;
  if (tid == 0) Tk[0] = (int)gridDim.x + ticket;       // (the next write to Tk[0] is at least one head = several barriers away)
  __syncthreads();
  chunk = __builtin_amdgcn_readfirstlane(Tk[0]);       // (uniform: the heads' base addresses stay in scalar registers)
  }
  if (queue && tid == 0) {
    const int old = __hip_atomic_fetch_add(queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (int)gridDim.x - 1) { atomicExch(queue, 0); atomicExch(queue + 1, 0); }
  }
  // d(q|k|v bias): the workgroup's heads were summed in LDS (ds_add_f32); one global atomic per column and workgroup
  // (2 x CUs workgroups) instead of one per column, wave and head
  if (dbias) {
    for (int i = tid; i < 3 * H; i += NTHR) atomicAdd(dbias + i, Cs[i]);
  }
  if (Q) attn_q_amax<NW>(aq, qmax, Ms, tid);           // (Ms: no head is in flight any more)
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
extern "C" int uc2_attn_mfma_supported(int L, int D) { return (D == 32 || D == 64) && L >= 1 && L <= 32 * AM_MAXW; }

template <int D, int NW>
static int launch_fwd(int B, int L, int nh, const void* qkv, const float* mask, float scale, float drop_p,
                      const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, int ilv, AttnQ aq, hipStream_t st) {
  constexpr int RS = D * 2 + 16, Lp = NW * 32;
  const size_t smem = 2 * Lp * RS + 2 * Lp * sizeof(float);
  const uint32_t th = drop_thresh(drop_p);
  const bool full = L == Lp && AM_LEAN;
  auto kern = aq.q ? (full ? (th ? attn_fwd_mfma_kernel<D, NW, true, true, true> : attn_fwd_mfma_kernel<D, NW, true, false, true>)
                           : (th ? attn_fwd_mfma_kernel<D, NW, false, true, true> : attn_fwd_mfma_kernel<D, NW, false, false, true>))
                   : (full ? (th ? attn_fwd_mfma_kernel<D, NW, true, true> : attn_fwd_mfma_kernel<D, NW, true, false>)
                           : (th ? attn_fwd_mfma_kernel<D, NW, false, true> : attn_fwd_mfma_kernel<D, NW, false, false>));
  hipLaunchKernelGGL(kern, dim3(B * nh), dim3(NW * 64), smem, st, L, nh, (const bf16*)qkv, mask, scale, th, 1.0f / (1.0f - drop_p),
                     seed_ptr, seed_imm, (bf16*)ctx, lse, ilv, aq);
  UC2_LAUNCH_CHECK();
  return 0;
}
template <int D, int NW>
static int launch_bwd(int B, int L, int nh, const void* qkv, const float* mask, float scale, float drop_p,
                      const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx,
                      const float* lse, void* dqkv, float* dbias, int* queue, int ilv, AttnQ aq, hipStream_t st) {
  constexpr int RS = D * 2 + 16, Lp = NW * 32, RSD = Lp * 2 + 16;
  const size_t smem = 3 * Lp * RS + Lp * RSD + 5 * Lp * sizeof(float) + 16 + (dbias ? 3 * (size_t)nh * D * sizeof(float) : 0);
  const uint32_t th = drop_thresh(drop_p);
  const bool full = L == Lp && AM_LEAN;
  auto kern = aq.q ? (full ? (th ? attn_bwd_mfma_kernel<D, NW, true, true, true> : attn_bwd_mfma_kernel<D, NW, true, false, true>)
                           : (th ? attn_bwd_mfma_kernel<D, NW, false, true, true> : attn_bwd_mfma_kernel<D, NW, false, false, true>))
                   : (full ? (th ? attn_bwd_mfma_kernel<D, NW, true, true> : attn_bwd_mfma_kernel<D, NW, true, false>)
                           : (th ? attn_bwd_mfma_kernel<D, NW, false, true> : attn_bwd_mfma_kernel<D, NW, false, false>));
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  }
  const int nbh = B * nh;
  // heads per workgroup (next head prefetched into registers): two workgroups are resident per CU (LDS, registers), so
  // the grid is ONE full round of 2 x CUs workgroups that walk ceil(nbh / grid) heads each (12 288 heads on 256 CUs:
  // 512 workgroups x 24 heads; a cap of 16 heads made it 768 workgroups = 1.5 rounds, a quarter of the time half empty)
  static int cus_cached = 0;
  if (!cus_cached) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    cus_cached = cus;
  }
  const int slots = 2 * cus_cached;
  int hpw = nbh > slots ? (nbh + slots - 1) / slots : 1;
  int grid = (nbh + hpw - 1) / hpw;
  if (queue && hpw > 4) {                              // dynamic chunks of a sixth of a static share (at least 4 heads), one resident round
    hpw = hpw >= 24 ? hpw / 6 : 4;
    const int nchunk = (nbh + hpw - 1) / hpw;
    grid = nchunk < slots ? nchunk : slots;
  } else {
    queue = nullptr;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, L, nh, (const bf16*)qkv, mask, scale,
                     th, 1.0f / (1.0f - drop_p), seed_ptr, seed_imm, (const bf16*)ctx,
                     (const bf16*)dctx, lse, (bf16*)dqkv, dbias, nbh, hpw, queue, ilv, aq);
  UC2_LAUNCH_CHECK();
  return 0;
}

#define AM_DISPATCH(FN, ...)                                                   \
  do {                                                                         \
    const int nw = (L + 31) / 32;                                              \
    if (D == 64) {                                                             \
      switch (nw) {                                                            \
        case 1: return FN<64, 1>(__VA_ARGS__);                                 \
        case 2: return FN<64, 2>(__VA_ARGS__);                                 \
        case 3: return FN<64, 3>(__VA_ARGS__);                                 \
        case 4: return FN<64, 4>(__VA_ARGS__);                                 \
        default: return FN<64, 5>(__VA_ARGS__);                                \
      }                                                                        \
    } else {                                                                   \
      switch (nw) {                                                            \
        case 1: return FN<32, 1>(__VA_ARGS__);                                 \
        case 2: return FN<32, 2>(__VA_ARGS__);                                 \
        case 3: return FN<32, 3>(__VA_ARGS__);                                 \
        case 4: return FN<32, 4>(__VA_ARGS__);                                 \
        default: return FN<32, 5>(__VA_ARGS__);                                \
      }                                                                        \
    }                                                                          \
  } while (0)

static int attn_fwd_mfma_impl(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                              float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse,
                              int ilv, AttnQ aq, void* stream) {
  UC2_CHECK_ARG(uc2_attn_mfma_supported(L, D));
  UC2_CHECK_ARG(B >= 0 && nh >= 1 && drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG(((nh * D) % 8) == 0);
  if (B == 0) return 0;
  UC2_CHECK_ARG(qkv && ctx);
  hipStream_t st = (hipStream_t)stream;
  AM_DISPATCH(launch_fwd, B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, ilv, aq, st);
}
extern "C" int uc2_attn_fwd_mfma(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse,
                                 int ilv, void* stream) {
  return attn_fwd_mfma_impl(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, ilv, AttnQ{}, stream);
}
// fp8 mode: uc2_attn_fwd (MFMA kernels, plain q|k|v layout) that also writes q_out = sat_e4m3(ctx * scale) [B L, nh D] for the output
// projection's GEMM -- delayed scaling, cell groups as uc2_fp8_quant_delayed.  -2 (nothing launched) when the MFMA kernels do not
// take (L, D).
extern "C" int uc2_attn_fwd_q(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale, float drop_p,
                              const uint64_t* seed_ptr, uint64_t seed_imm, void* ctx, float* lse, void* q_out,
                              const void* amax_prev, void* amax_next, void* amax_clear, float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(q_out && amax_prev && amax_next && amax_clear && q_scale_out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  if (!uc2_attn_mfma_supported(L, D) || ((uintptr_t)q_out & 7)) return -2;
  return attn_fwd_mfma_impl(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, lse, 0,
                            AttnQ{(uint8_t*)q_out, (const unsigned*)amax_prev, (unsigned*)amax_next, (unsigned*)amax_clear, q_scale_out}, stream);
}

static int attn_bwd_mfma_impl(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                              float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx,
                              const void* dctx, const float* lse, void* dqkv, float* dbias, int* queue, int ilv, AttnQ aq, void* stream) {
  UC2_CHECK_ARG(uc2_attn_mfma_supported(L, D));
  UC2_CHECK_ARG(B >= 0 && nh >= 1 && drop_p >= 0.f && drop_p < 1.f);
  UC2_CHECK_ARG(((nh * D) % 8) == 0);
  if (B == 0) return 0;
  UC2_CHECK_ARG(qkv && ctx && dctx && lse && dqkv);
  hipStream_t st = (hipStream_t)stream;
  AM_DISPATCH(launch_bwd, B, L, nh, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, dbias, queue, ilv, aq, st);
}
extern "C" int uc2_attn_bwd_mfma(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale,
                                 float drop_p, const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx,
                                 const void* dctx, const float* lse, void* dqkv, float* dbias, int* queue, int ilv, void* stream) {
  return attn_bwd_mfma_impl(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, dbias, queue, ilv, AttnQ{}, stream);
}
// fp8 mode: uc2_attn_bwd (MFMA kernels, plain q|k|v layout) that also writes q_out = sat_e4m3(dqkv * scale) [B L, 3 nh D] for the
// q|k|v projection's input-gradient GEMM.  -2 (nothing launched) when the MFMA kernels do not take (L, D).
extern "C" int uc2_attn_bwd_q(int B, int L, int nh, int D, const void* qkv, const float* mask, float scale, float drop_p,
                              const uint64_t* seed_ptr, uint64_t seed_imm, const void* ctx, const void* dctx, const float* lse,
                              void* dqkv, float* dbias_qkv, int* queue, void* q_out, const void* amax_prev, void* amax_next,
                              void* amax_clear, float* q_scale_out, void* stream) {
  UC2_CHECK_ARG(q_out && amax_prev && amax_next && amax_clear && q_scale_out);
  UC2_CHECK_ARG(amax_prev != amax_next && amax_next != amax_clear && amax_prev != amax_clear);
  if (!uc2_attn_mfma_supported(L, D) || ((uintptr_t)q_out & 7) || AM_LINE_STORES) return -2;
  return attn_bwd_mfma_impl(B, L, nh, D, qkv, mask, scale, drop_p, seed_ptr, seed_imm, ctx, dctx, lse, dqkv, dbias_qkv, queue, 0,
                            AttnQ{(uint8_t*)q_out, (const unsigned*)amax_prev, (unsigned*)amax_next, (unsigned*)amax_clear, q_scale_out}, stream);
}
