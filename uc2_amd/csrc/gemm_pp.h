// Helpers shared by the persistent ping-pong GEMM kernels (gemm_pp.hip, gemm_roll.hip): staging sources, fragment
// addresses and reads, the wave-private LDS transposition buffer of the epilogues, the exposed epilogue itself.
#pragma once
#include "gemm_tile.h"

#define PP_UNIT 16384

template <int J, int HI> __device__ __forceinline__ int pp_map(int ur) {     // unit row -> row/column of the tile
  if (J == 0) return (ur >> 6) * (64 + 32 * HI) + (ur & 63);                        // A0: 64 rows per wave row
  else if (J == 3) return HI == 2 ? (ur >> 6) * 128 + 64 + (ur & 63) : (HI == 1 ? (ur >> 5) * 96 + 64 + (ur & 31) : 0);   // A1: 32*HI rows per wave row (HI = 0: no such unit)
  else return (ur >> 5) * 64 + (J == 2 ? 32 : 0) + (ur & 31);
}

// per-lane source of wave-instruction wi (0..15) of a unit; same LDS images and swizzles as gf_src<TR,128,64>
template <bool TR, int J, int HI>
__device__ __forceinline__ const bf16* pp_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int wi,
                                              int l) {
  if (!TR) {
    const int row = wi * 8 + (l >> 3), cp = l & 7;
    const int c = cp ^ ((row >> 1) & 7);
    const int gr = min(r0 + pp_map<J, HI>(row), rows - 1);
    return X + (size_t)gr * ld + kbeg + c * 8;
  } else {
    const int krow = wi * 4 + (l >> 4), cp = l & 15;
    const int c = cp ^ ((krow & 3) << 2);
    const int col = min(r0 + pp_map<J, HI>(c * 8), rows - 8);
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

// Direct epilogue in two parts (loads + arithmetic, then stores only):
//   pp_epi_compute: aux_in loads (EPI_ADD / EPI_DGELU), lane-half exchange, activation, conversion.  It leaves
//                   the wave's 128x64 outputs (and the second stream of the GELU kinds) as packed bf16x8 registers.  Its
//                   loads queue behind the next item's staged units (vmcnt retires in order; those were issued four
//                   or more phases earlier by the main loop's tail), so their exposed cost is their own latency.
//                   (The bias is not added here: the accumulators START at the bias, see the kernel.)
//   pp_epi_store  : the stores only; full tiles, so their number is a compile-time function of the epilogue kind
//                   (the item barrier's counted vmcnt relies on it).
// Layout after the exchange (v_permlane32_swap, see gemm_common.h): lane (row m, half h) holds columns
// 32j + 16g + 8h .. +7 of the wave's 64 for j, g in {0,1}.
// Column sums across the 32 lanes of a lane half, 32 partial sums per lane (index v) -> lane (c31, h) returns the
// total of v = c31.  Transposing butterfly: at step k a lane keeps the half of its values whose bit (4-k) of v
// matches its own lane bit and adds the partner's copies of those -- 31 exchanges instead of 5 x 32.
// (colsum_butterfly32: common.h)

// ---- full-line epilogue I/O through a wave-private 4 KiB LDS transposition buffer ---------------------------------
// After the lane-half exchange a lane owns (row r = lane & 31, half h) of a 32-row block: four 16-byte pieces of one
// row, 32 bytes apart.  Stored (or loaded) like that, one wave-instruction touches 64 different 16-byte fragments of
// 32 rows -- 64 separate L2 requests -- and the epilogue's store tail was request-bound (measured: the same bytes in a
// full-line pattern ran the GELU GEMM 19 % faster, the plain ones 7 %).  So every global access of the epilogue is made
// in the LINE layout: lane l owns the 16-byte chunk (l & 7) of row 8*it + (l >> 3), eight lanes cover one 128-byte line,
// one wave-instruction = 8 whole lines.  The two layouts are exchanged through 32 rows x 128 B of LDS per wave (the
// 32 KiB the 128 KiB ring leaves free), chunk c of row r at r*128 + ((c ^ (r & 7)) << 4): conflict-free for the
// ds_write_b128 / ds_read_b128 of both layouts.  LDS executes a wave's operations in issue order, so the buffer is
// reused back to back without waits; only the consumer of a read waits (lgkmcnt).  All of it is inline asm: hipcc
// would otherwise order plain LDS accesses behind the LDS-DMA in flight (vmcnt(0)).
__device__ __forceinline__ void tp_write(unsigned addr, const bf16x8& v) {
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
template <int OFF> __device__ __forceinline__ void tp_write_o(unsigned addr, const bf16x8& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void tp_read(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
}
template <int OFF> __device__ __forceinline__ void tp_read_o(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
#define TP_WAIT4(A, B, C, D) do { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

struct TpAddr {
  unsigned line;         // LINE layout: + it * 1024
  unsigned rh[4];        // (row, half) layout: piece k = 2j + g  (chunk 2k + h)
};
__device__ __forceinline__ TpAddr tp_addr(unsigned tb, int lane) {
  TpAddr t;
  const int lr = lane >> 3, lc = lane & 7, r = lane & 31, h = lane >> 5;
  t.line = tb + lr * 128 + ((lc ^ lr) << 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) t.rh[k] = tb + r * 128 + (((2 * k + h) ^ (r & 7)) << 4);
  return t;
}

struct PpOut {
  bf16x8 o[2][2][4];          // [A half][i][it], LINE layout: row 8*it + (lane >> 3) of the 32-row block, columns 8*(lane & 7) .. +7
  bf16x8 pre[2][2][4];        // second output stream of the GELU epilogues (aux_out), same layout; unused otherwise
};

template <int EPI, int HI>
__device__ __forceinline__ void pp_epi_compute(const GemmArgs& p, const f32x16 (&acc)[2][2][2], PpOut& out, int mb0, int nb,
                                               int lane, const TpAddr& ta) {
  const int h = lane >> 5, c31 = lane & 31, lr = lane >> 3, lc = lane & 7;
  constexpr bool has_aux = (EPI == EPI_DGELU || EPI == EPI_ADD || EPI == EPI_MUL);
  // EPI_DGELU: aux_out (fp32 [N]) += column sums of the result = bias gradient of the layer whose pre-activation
  // gradient this GEMM produces (saves a separate pass over the M x N result)
  const bool want_cs = (EPI == EPI_DGELU || EPI == EPI_MUL) && p.aux_out != nullptr;
  float cs[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) cs[v] = 0.f;
  bf16x8 ax[2][2][4];
  if (has_aux) {                                       // whole 128-byte lines, all 16 loads in flight at once
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (hh == 1 && i >= HI) continue;
          const int m = mb0 + hh * 64 + i * 32 + 8 * it + lr;
          ax[hh][i][it] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + (size_t)m * p.ldaux + nb + 8 * lc));
        }
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (hh == 1 && i >= HI) continue;
      bf16x8 axr[4];                                   // aux tile of this block in the (row, half) layout
      if (has_aux) {
        tp_write_o<0>(ta.line, ax[hh][i][0]); tp_write_o<1024>(ta.line, ax[hh][i][1]);
        tp_write_o<2048>(ta.line, ax[hh][i][2]); tp_write_o<3072>(ta.line, ax[hh][i][3]);
        tp_read(axr[0], ta.rh[0]); tp_read(axr[1], ta.rh[1]); tp_read(axr[2], ta.rh[2]); tp_read(axr[3], ta.rh[3]);
        TP_WAIT4(axr[0], axr[1], axr[2], axr[3]);
      }
      bf16x8 pre[4], o[4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int k = 2 * j + g;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float fx = acc[hh][i][j][8 * g + e], fy = acc[hh][i][j][8 * g + 4 + e];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fx), __float_as_uint(fy), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
          }
          if (EPI == EPI_GELU || EPI == EPI_GELU_NOAUX) {
            if (EPI == EPI_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) pre[k][e] = (bf16)v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_bf(v[e]);
          } else if (EPI == EPI_GELU_D) {               // aux_out <- gelu'(pre): the backward multiplies by it (EPI_MUL)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float gg, dd;
              gelu_and_dgelu_bf(v[e], gg, dd);
              pre[k][e] = (bf16)dd;
              v[e] = gg;
            }
          } else if (EPI == EPI_DGELU || EPI == EPI_MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= (EPI == EPI_MUL) ? (float)axr[k][e] : dgelu_bf((float)axr[k][e]);
            if (want_cs) {
#pragma unroll
              for (int e = 0; e < 8; ++e) cs[16 * j + 8 * g + e] += v[e];
            }
          } else if (EPI == EPI_ADD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)axr[k][e];
          } else if (EPI == EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanh_bf(v[e]);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) o[k][e] = (bf16)v[e];
        }
      if (EPI == EPI_GELU || EPI == EPI_GELU_D) {
        // the second stream is kept in registers (the accumulators are dead by now) and stored AFTER the next item's
        // LDS-DMA has been issued, like the main output: stores issued before it are older than the DMA in the in-order
        // vmcnt queue, and the next tile's first wait then has to sit through their write acknowledgements
        tp_write(ta.rh[0], pre[0]); tp_write(ta.rh[1], pre[1]); tp_write(ta.rh[2], pre[2]); tp_write(ta.rh[3], pre[3]);
        tp_read_o<0>(out.pre[hh][i][0], ta.line); tp_read_o<1024>(out.pre[hh][i][1], ta.line);
        tp_read_o<2048>(out.pre[hh][i][2], ta.line); tp_read_o<3072>(out.pre[hh][i][3], ta.line);
        TP_WAIT4(out.pre[hh][i][0], out.pre[hh][i][1], out.pre[hh][i][2], out.pre[hh][i][3]);
      }
      tp_write(ta.rh[0], o[0]); tp_write(ta.rh[1], o[1]); tp_write(ta.rh[2], o[2]); tp_write(ta.rh[3], o[3]);
      tp_read_o<0>(out.o[hh][i][0], ta.line); tp_read_o<1024>(out.o[hh][i][1], ta.line);
      tp_read_o<2048>(out.o[hh][i][2], ta.line); tp_read_o<3072>(out.o[hh][i][3], ta.line);
      TP_WAIT4(out.o[hh][i][0], out.o[hh][i][1], out.o[hh][i][2], out.o[hh][i][3]);
    }
  if (EPI == EPI_DGELU || EPI == EPI_MUL) {
    if (want_cs) {                                     // (wave-uniform) one 64-lane atomic per wave and tile
      const float tot = colsum_butterfly32(cs, lane);
      const int vv = c31;                              // value index v = 16 j + 8 g + e  ->  column 32 j + 16 g + 8 h + e
      atomicAdd(reinterpret_cast<float*>(p.aux_out) + nb + 32 * (vv >> 4) + 16 * ((vv >> 3) & 1) + 8 * h + (vv & 7), tot);
    }
  }
}

// The same epilogue without the lane-half exchange.  Register group 4c .. 4c+3 of accumulator block j is four consecutive
// columns 32 j + 8 c + 4 h .. + 3 of row (lane & 31): converted, they are one 8-byte piece = half h of 16-byte chunk 4 j + c of
// the row in the transposition buffer, stored with ds_write_b64 (2-way bank conflict, still a quarter of the issue time of the
// 16 v_permlane32_swap per block); the aux tile is read back from the buffer in the same 8-byte pieces.  Bit-identical
// results: only the route of the values through the wave changes.
template <int EPI, int HI>
__device__ __forceinline__ void pp_epi_compute_q(const GemmArgs& p, const f32x16 (&acc)[2][2][2], PpOut& out, int mb0, int nb,
                                                 int lane, const TpAddr& ta) {
  const int h = lane >> 5, c31 = lane & 31, lr = lane >> 3, lc = lane & 7;
  constexpr bool has_aux = (EPI == EPI_DGELU || EPI == EPI_ADD || EPI == EPI_MUL);
  constexpr bool two = (EPI == EPI_GELU || EPI == EPI_GELU_D);
  const bool want_cs = (EPI == EPI_DGELU || EPI == EPI_MUL) && p.aux_out != nullptr;
  const unsigned q0 = (ta.rh[0] & ~127u) + ((unsigned)(c31 & 7) << 4) + 8u * (unsigned)h;     // piece (j, c): q0 ^ ((4 j + c) << 4)
  float cs[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) cs[v] = 0.f;
  bf16x8 ax[2][2][4];
  if (has_aux) {                                       // whole 128-byte lines, all 16 loads in flight at once
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (hh == 1 && i >= HI) continue;
          const int m = mb0 + hh * 64 + i * 32 + 8 * it + lr;
          ax[hh][i][it] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + (size_t)m * p.ldaux + nb + 8 * lc));
        }
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (hh == 1 && i >= HI) continue;
      bf16x4 axq[8];                                   // aux tile of this block, piece (j, c) at [4 j + c]
      if (has_aux) {
        tp_write_o<0>(ta.line, ax[hh][i][0]); tp_write_o<1024>(ta.line, ax[hh][i][1]);
        tp_write_o<2048>(ta.line, ax[hh][i][2]); tp_write_o<3072>(ta.line, ax[hh][i][3]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned qa = q0 ^ (unsigned)(k << 4);
          asm volatile("ds_read_b64 %0, %1" : "=v"(axq[k]) : "v"(qa) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(axq[0]), "+v"(axq[1]), "+v"(axq[2]), "+v"(axq[3]), "+v"(axq[4]), "+v"(axq[5]), "+v"(axq[6]), "+v"(axq[7]) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x4 pre[8];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = 4 * j + c;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[hh][i][j][4 * c + e];
          if (EPI == EPI_GELU || EPI == EPI_GELU_NOAUX) {
            if (EPI == EPI_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) pre[k][e] = (bf16)v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_bf(v[e]);
          } else if (EPI == EPI_GELU_D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float gg, dd;
              gelu_and_dgelu_bf(v[e], gg, dd);
              pre[k][e] = (bf16)dd;
              v[e] = gg;
            }
          } else if (EPI == EPI_DGELU || EPI == EPI_MUL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (EPI == EPI_MUL) ? (float)axq[k][e] : dgelu_bf((float)axq[k][e]);
            if (want_cs) {
#pragma unroll
              for (int e = 0; e < 4; ++e) cs[16 * j + 4 * c + e] += v[e];
            }
          } else if (EPI == EPI_ADD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)axq[k][e];
          } else if (EPI == EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanh_bf(v[e]);
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
          const unsigned qa = q0 ^ (unsigned)(k << 4);
          asm volatile("ds_write_b64 %0, %1" :: "v"(qa), "v"(o) : "memory");
        }
      tp_read_o<0>(out.o[hh][i][0], ta.line); tp_read_o<1024>(out.o[hh][i][1], ta.line);
      tp_read_o<2048>(out.o[hh][i][2], ta.line); tp_read_o<3072>(out.o[hh][i][3], ta.line);
      if (two) {                                       // second stream: same route, after the reads of the first have been issued
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned qa = q0 ^ (unsigned)(k << 4);
          asm volatile("ds_write_b64 %0, %1" :: "v"(qa), "v"(pre[k]) : "memory");
        }
        tp_read_o<0>(out.pre[hh][i][0], ta.line); tp_read_o<1024>(out.pre[hh][i][1], ta.line);
        tp_read_o<2048>(out.pre[hh][i][2], ta.line); tp_read_o<3072>(out.pre[hh][i][3], ta.line);
        TP_WAIT4(out.pre[hh][i][0], out.pre[hh][i][1], out.pre[hh][i][2], out.pre[hh][i][3]);
      }
      TP_WAIT4(out.o[hh][i][0], out.o[hh][i][1], out.o[hh][i][2], out.o[hh][i][3]);
    }
  if (EPI == EPI_DGELU || EPI == EPI_MUL) {
    if (want_cs) {                                     // (wave-uniform) one 64-lane atomic per wave and tile
      const float tot = colsum_butterfly32(cs, lane);
      const int vv = c31;                              // value index v = 16 j + 4 c + e  ->  column 32 j + 8 c + 4 h + e
      atomicAdd(reinterpret_cast<float*>(p.aux_out) + nb + 32 * (vv >> 4) + 8 * ((vv >> 2) & 3) + 4 * h + (vv & 3), tot);
    }
  }
}

template <int EPI, int HI>
__device__ __forceinline__ void pp_epi_store(const GemmArgs& p, const PpOut& out, int mb0, int nb, int lane) {
  const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (hh == 1 && i >= HI) continue;
      bf16* c0 = reinterpret_cast<bf16*>(p.C) + (size_t)(mb0 + hh * 64 + i * 32 + lr) * p.ldc + nb + 8 * lc;
#pragma unroll
      for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(out.o[hh][i][it], reinterpret_cast<bf16x8*>(c0 + (size_t)(8 * it) * p.ldc));
      if (EPI == EPI_GELU || EPI == EPI_GELU_D) {          // (the host only selects these kinds with aux_out set)
        bf16* a0 = reinterpret_cast<bf16*>(p.aux_out) + (size_t)(mb0 + hh * 64 + i * 32 + lr) * p.ldaux + nb + 8 * lc;
#pragma unroll
        for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(out.pre[hh][i][it], reinterpret_cast<bf16x8*>(a0 + (size_t)(8 * it) * p.ldaux));
      }
    }
}

// fp32 partial tile of one split-K item (two-stage reduction): transposed accumulators, lane = row m with 4
// consecutive columns per register group; through the same LDS transposition -> whole 128-byte lines,
// 32 stores per wave (compile-time count, full tiles)
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int HI, bool ACCUM = false>
__device__ __forceinline__ void pp_partial_store(float* __restrict__ dst, int ldn, const f32x16 (&acc)[2][2][2], int mb0, int nb,
                                                 int lane, const TpAddr& ta) {
  const int lr = lane >> 3, lc = lane & 7;
  // ACCUM: dst += tile (a weight gradient that is not split: the vocabulary-long dE of the tied decoder).  The 16-byte pieces a
  // lane will store are also the pieces it loads, two groups (8 loads) ahead of their use, so the read of C hides behind the
  // transposition and the stores of the groups before it.
  constexpr int NG = 2 * (2 + HI);                     // groups (hh, i, j), in store order
  auto rowp = [&](int g) __attribute__((always_inline)) {
    const int hh = g >> 2, i = (g >> 1) & 1, j = g & 1;
    return dst + (size_t)(mb0 + hh * 64 + i * 32 + lr) * ldn + nb + 32 * j + 4 * lc;
  };
  f32x4v cin[2][4];
  if (ACCUM) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int it = 0; it < 4; ++it) cin[g][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rowp(g) + (size_t)(8 * it) * ldn));
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int hh = g >> 2, i = (g >> 1) & 1, j = g & 1;
    bf16x8 w[4], t[4];                                  // 16 bytes = 4 floats (columns 32j + 8c + 4h .. +3  ->  chunk 2c + h)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      w[c] = __builtin_bit_cast(bf16x8, make_float4(acc[hh][i][j][4 * c], acc[hh][i][j][4 * c + 1], acc[hh][i][j][4 * c + 2], acc[hh][i][j][4 * c + 3]));
    tp_write(ta.rh[0], w[0]); tp_write(ta.rh[1], w[1]); tp_write(ta.rh[2], w[2]); tp_write(ta.rh[3], w[3]);
    tp_read_o<0>(t[0], ta.line); tp_read_o<1024>(t[1], ta.line); tp_read_o<2048>(t[2], ta.line); tp_read_o<3072>(t[3], ta.line);
    TP_WAIT4(t[0], t[1], t[2], t[3]);
    float* row = rowp(g);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4v v = __builtin_bit_cast(f32x4v, t[it]);
      if (ACCUM) v += cin[g & 1][it];
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(row + (size_t)(8 * it) * ldn));
    }
    if (ACCUM && g + 2 < NG) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        cin[g & 1][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rowp(g + 2) + (size_t)(8 * it) * ldn));
    }
  }
}

template <> __device__ __forceinline__ void wait_vmcnt<40>() { asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<1>() { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<7>() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<19>() { asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<31>() { asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); }
// n in 0..8 (wave-uniform; a compile-time constant in the steady-state loop)
__device__ __forceinline__ void pp_wait_small(int n) {
  if (n >= 8) wait_vmcnt<8>(); else if (n == 7) wait_vmcnt<7>(); else if (n == 6) wait_vmcnt<6>(); else if (n == 5) wait_vmcnt<5>();
  else if (n == 4) wait_vmcnt<4>(); else if (n == 3) wait_vmcnt<3>(); else if (n == 2) wait_vmcnt<2>(); else if (n == 1) wait_vmcnt<1>();
  else wait_vmcnt<0>();
}

