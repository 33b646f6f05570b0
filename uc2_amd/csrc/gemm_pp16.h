// Helpers of the ping-pong GEMM on v_mfma_f32_16x16x32_bf16 (gemm_pp16.hip, variant 12).  Same tile, ring, phases and
// epilogue I/O as gemm_pp.hip; what changes is everything that depends on the MFMA operand / accumulator layout:
//   operand fragment  : lane l holds row (or column) l & 15 of a 16-row block, k = 32 ks + 8 g .. + 7, g = l >> 4
//   accumulator block : D^T[n][m] 16 x 16, lane l holds token row m = l & 15 and columns n = 4 g + r, r = 0..3
// so a lane owns one token row and, per 16-column block, four consecutive columns = one 8-byte bf16 piece (one 16-byte fp32
// piece) of the output row -- the same "pieces" the 32x32 kernel's swap-free epilogue works with, at other addresses.
#pragma once
#include "gemm_pp.h"

// chunk-slot swizzle of the k-strided LDS image as a function of the k row (experiment knob: 1 = variant 8's swizzle, 2 = bit 4)
#ifndef UC2_PP16_SWZ
#define UC2_PP16_SWZ 0
#endif
#if UC2_PP16_SWZ == 1
#define PP16_SWZ(krow) (((krow) & 3) << 2)
#elif UC2_PP16_SWZ == 2
#define PP16_SWZ(krow) ((((krow) & 3) << 2) | ((((krow) >> 4) & 1) << 1))
#elif UC2_PP16_SWZ == 3
#define PP16_SWZ(krow) ((((krow) & 3) << 2) | ((((krow) >> 3) & 1) << 1) | (((krow) >> 4) & 1))
#else
#define PP16_SWZ(krow) ((((krow) & 3) << 2) | ((((krow) >> 3) & 1) << 1))
#endif

// per-lane staging source: as pp_src, but the k-strided image gets a second swizzle bit (bit 3 of the k row): the
// transposing read of a 16x16x32 operand takes two 4 x 16 blocks 8 k-rows apart in the same columns per 32 lanes, which land
// on the same banks with the (krow & 3) swizzle alone
template <bool TR, int J, int HI>
__device__ __forceinline__ const bf16* pp16_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int wi, int l) {
  if (!TR) {
    return pp_src<false, J, HI>(X, ld, rows, r0, kbeg, wi, l);
  } else {
    const int krow = wi * 4 + (l >> 4), cp = l & 15;
    const int c = cp ^ PP16_SWZ(krow);
    const int col = min(r0 + pp_map<J, HI>(c * 8), rows - 8);
    return X + (size_t)(kbeg + krow) * ld + col;
  }
}

// byte offset (inside a unit) of the lane's part of fragment (block 0, k-step 0) for rows / columns rbase .. of the unit;
//   k-contiguous image: fragment (blk, ks) at (off ^ (ks << 6)) + blk * 2048
//   k-strided image   : fragment (blk, ks) at (off ^ (blk << 5)) + ks * 8192, second half of the k octet + 1024
//   (pp16_read takes the k-step's base off ^ (ks << 6) resp. the block's base off ^ (blk << 5); the kernel keeps them in
//    registers: every VALU instruction in an L section competes with the partner wave's MFMA issue)
template <bool TR> __device__ __forceinline__ unsigned pp16_frag_off(int rbase, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if (!TR) {
    const int row = rbase + i;
    return row * 128 + ((g ^ ((row >> 1) & 7)) << 4);
  } else {
    const int q = i >> 2, pp = i & 3;
    const int krow = 8 * g + q;
    const int chunk = (rbase >> 3) + (pp >> 1);
    return krow * 256 + ((chunk ^ PP16_SWZ(krow)) << 4) + (pp & 1) * 8;
  }
}
template <bool TR, int BLK, int KS, int OFF> __device__ __forceinline__ void pp16_read(bf16x8& dst, unsigned addr) {
  // addr: the k-step's (k-contiguous image) resp. the block's (k-strided image) base in the k-tile's buffer; OFF: the unit's
  // offset in the buffer.  No address arithmetic here: every VALU instruction of an L section competes with the partner
  // wave's MFMA issue (16 MFMAs per C section hold the SIMD's vector issue for half of it).
  if (!TR) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(BLK * 2048 + OFF));
  } else {
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(KS * 8192 + OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(KS * 8192 + 1024 + OFF));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    dst = bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

// column sums over the 16 lanes of a DPP row (= the lanes with the same g), 16 partial sums per lane: lane i of the row
// returns the total of v = i (the last four stages of colsum_butterfly32)
__device__ __forceinline__ float colsum_butterfly16(float (&v)[16], int lane) {
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = v[i] + dpp_f32<0x128>(v[i]), b = v[i + 8] + dpp_f32<0x128>(v[i + 8]);
      v[i] = up ? b : a;
    }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[i] + dpp_f32<0x104>(v[i]), b = v[i + 4] + dpp_f32<0x114>(v[i + 4]);
      v[i] = up ? b : a;
    }
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a = v[i] + dpp_f32<0x4E>(v[i]), b = v[i + 2] + dpp_f32<0x4E>(v[i + 2]);
      v[i] = up ? b : a;
    }
  }
  {
    const float a = v[0] + dpp_f32<0xB1>(v[0]), b = v[1] + dpp_f32<0xB1>(v[1]);
    v[0] = (lane & 1) ? b : a;
  }
  return v[0];
}

// Epilogue arithmetic (pp_epi_compute_q of gemm_pp.h for the 16 x 16 accumulator blocks).  acc[hh][mb][nb]: 16-row block mb of
// A half hh, 16-column block nb; a 32-row output block (hh, i) is the block pair mb = 2 i, 2 i + 1.  Piece k = 4 mbl + nb of
// the lane = row 16 mbl + (lane & 15) of the 32-row block, columns 16 nb + 4 g .. + 3 = half (g & 1) of 16-byte chunk
// 2 nb + (g >> 1) of the 128-byte row in the transposition buffer: (q0 ^ (nb << 5)) + mbl * 2048.
#ifndef UC2_GELU_PACKED
#define UC2_GELU_PACKED 1
#endif
template <int EPI>
__device__ __forceinline__ void pp16_epi_compute_q(const GemmArgs& p, const f32x4 (&acc)[2][4][4], PpOut& out, int mb0, int nb0,
                                                   int lane, const TpAddr& ta) {
  const int g = lane >> 4, r15 = lane & 15, lr = lane >> 3, lc = lane & 7;
  constexpr bool has_aux = (EPI == EPI_DGELU || EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_DROPADD);
  constexpr bool two = (EPI == EPI_GELU || EPI == EPI_GELU_D);
  const bool want_cs = (EPI == EPI_DGELU || EPI == EPI_MUL) && p.aux_out != nullptr;
  const unsigned tb = ta.line - (unsigned)(lr * 128 + ((lc ^ lr) << 4));
  const unsigned q0 = tb + (unsigned)r15 * 128u + ((unsigned)((g >> 1) ^ (r15 & 7)) << 4) + 8u * (unsigned)(g & 1);
  // EPI_DROPADD: the column part of the dropout variate for the lane's four pieces of the tile (common.h, drop_keep4); the launcher
  // routes p == 0 to EPI_ADD, so no run-time test here
  uint32_t hcol[4];
  uint64_t dseed = 0;
  if (EPI == EPI_DROPADD) {
    dseed = p.drop_seed_imm + (p.drop_seed_ptr ? *p.drop_seed_ptr : 0ull);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hcol[nb] = drop_col_hash(dseed, (uint32_t)((nb0 >> 2) + 4 * nb + g));
  }
  float cs[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) cs[v] = 0.f;
  bf16x8 ax[2][2][4];
  if (has_aux) {                                       // whole 128-byte lines, all 16 loads in flight at once
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = mb0 + hh * 64 + i * 32 + 8 * it + lr;
          ax[hh][i][it] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + (size_t)m * p.ldaux + nb0 + 8 * lc));
        }
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x4 axq[8];                                   // aux tile of this block, piece k
      if (has_aux) {
        tp_write_o<0>(ta.line, ax[hh][i][0]); tp_write_o<1024>(ta.line, ax[hh][i][1]);
        tp_write_o<2048>(ta.line, ax[hh][i][2]); tp_write_o<3072>(ta.line, ax[hh][i][3]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned qa = q0 ^ (unsigned)((k & 3) << 5);
          if (k < 4) asm volatile("ds_read_b64 %0, %1" : "=v"(axq[k]) : "v"(qa) : "memory");
          else asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(axq[k]) : "v"(qa) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(axq[0]), "+v"(axq[1]), "+v"(axq[2]), "+v"(axq[3]), "+v"(axq[4]), "+v"(axq[5]), "+v"(axq[6]), "+v"(axq[7]) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x4 pre[8];
      uint32_t hrow[2];
      if (EPI == EPI_DROPADD) {
        hrow[0] = drop_row_hash(dseed, (uint32_t)(mb0 + hh * 64 + i * 32 + r15));
        hrow[1] = drop_row_hash(dseed, (uint32_t)(mb0 + hh * 64 + i * 32 + 16 + r15));
      }
#pragma unroll
      for (int mbl = 0; mbl < 2; ++mbl)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int k = 4 * mbl + nb;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[hh][2 * i + mbl][nb][e];
          if (EPI == EPI_GELU || EPI == EPI_GELU_NOAUX) {
            if (EPI == EPI_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) pre[k][e] = (bf16)v[e];
            }
#if UC2_GELU_PACKED
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const uc2_f2v gg = gelu_bf2(uc2_f2v{v[e], v[e + 1]});
              v[e] = gg.x; v[e + 1] = gg.y;
            }
#else
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_bf(v[e]);
#endif
          } else if (EPI == EPI_GELU_D) {
#if UC2_GELU_PACKED
#pragma unroll
            for (int e = 0; e < 4; e += 2) {              // pairs: packed fp32 arithmetic (common.h, gelu_and_dgelu_bf2)
              uc2_f2v gg, dd;
              gelu_and_dgelu_bf2(uc2_f2v{v[e], v[e + 1]}, gg, dd);
              pre[k][e] = (bf16)dd.x; pre[k][e + 1] = (bf16)dd.y;
              v[e] = gg.x; v[e + 1] = gg.y;
            }
#else
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float gg, dd;
              gelu_and_dgelu_bf(v[e], gg, dd);
              pre[k][e] = (bf16)dd;
              v[e] = gg;
            }
#endif
          } else if (EPI == EPI_DGELU || EPI == EPI_MUL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (EPI == EPI_MUL) ? (float)axq[k][e] : dgelu_bf((float)axq[k][e]);
            // (unconditional: under the run-time `want_cs` hipcc emitted a packed add AND a select per value -- 128 v_cndmask per
            //  tile and wave; the sums are simply not used when no column-sum output was asked for)
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[4 * nb + e] += v[e];
          } else if (EPI == EPI_ADD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)axq[k][e];
          } else if (EPI == EPI_DROPADD) {
            // dropout(dense output) + residual, the mask of element (row, column) exactly as ln_fwd / ln_bwd derive it
            bool kp[4];
            drop_keep4(hrow[mbl], hcol[nb], p.drop_thresh, kp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (kp[e] ? v[e] * p.drop_scale : 0.f) + (float)axq[k][e];
          } else if (EPI == EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanh_bf(v[e]);
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
          const unsigned qa = q0 ^ (unsigned)(nb << 5);
          if (mbl == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(qa), "v"(o) : "memory");
          else asm volatile("ds_write_b64 %0, %1 offset:2048" :: "v"(qa), "v"(o) : "memory");
        }
      tp_read_o<0>(out.o[hh][i][0], ta.line); tp_read_o<1024>(out.o[hh][i][1], ta.line);
      tp_read_o<2048>(out.o[hh][i][2], ta.line); tp_read_o<3072>(out.o[hh][i][3], ta.line);
      if (two) {                                       // second stream: same route, after the reads of the first have been issued
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned qa = q0 ^ (unsigned)((k & 3) << 5);
          if (k < 4) asm volatile("ds_write_b64 %0, %1" :: "v"(qa), "v"(pre[k]) : "memory");
          else asm volatile("ds_write_b64 %0, %1 offset:2048" :: "v"(qa), "v"(pre[k]) : "memory");
        }
        tp_read_o<0>(out.pre[hh][i][0], ta.line); tp_read_o<1024>(out.pre[hh][i][1], ta.line);
        tp_read_o<2048>(out.pre[hh][i][2], ta.line); tp_read_o<3072>(out.pre[hh][i][3], ta.line);
        TP_WAIT4(out.pre[hh][i][0], out.pre[hh][i][1], out.pre[hh][i][2], out.pre[hh][i][3]);
      }
      TP_WAIT4(out.o[hh][i][0], out.o[hh][i][1], out.o[hh][i][2], out.o[hh][i][3]);
    }
  if (EPI == EPI_DGELU || EPI == EPI_MUL) {
    if (want_cs) {                                     // (wave-uniform) one 64-lane atomic per wave and tile
      const float tot = colsum_butterfly16(cs, lane);
      const int vv = r15;                              // value index v = 4 nb + e  ->  column 16 nb + 4 g + e
      atomicAdd(reinterpret_cast<float*>(p.aux_out) + nb0 + 16 * (vv >> 2) + 4 * g + (vv & 3), tot);
    }
  }
}

// fp32 partial tile (split-K item), or C += tile (ACCUM): per pass one 32-row x 32-column fp32 block (hh, i, j) = block pairs
// mb = 2 i + mbl, nb = 2 j + nbl; the lane's f32x4 of (mbl, nbl) is 16-byte chunk 4 nbl + g of row 16 mbl + (lane & 15)
template <bool ACCUM = false>
__device__ __forceinline__ void pp16_partial_store(float* __restrict__ dst, int ldn, const f32x4 (&acc)[2][4][4], int mb0, int nb0,
                                                   int lane, const TpAddr& ta) {
  const int lr = lane >> 3, lc = lane & 7, g = lane >> 4, r15 = lane & 15;
  const unsigned tb = ta.line - (unsigned)(lr * 128 + ((lc ^ lr) << 4));
  const unsigned w0 = tb + (unsigned)r15 * 128u + ((unsigned)(g ^ (r15 & 7)) << 4);      // piece (mbl, nbl): (w0 ^ (nbl << 6)) + mbl * 2048
  constexpr int NG = 8;                                // groups (hh, i, j), in store order
  auto rowp = [&](int gi) __attribute__((always_inline)) {
    const int hh = gi >> 2, i = (gi >> 1) & 1, j = gi & 1;
    return dst + (size_t)(mb0 + hh * 64 + i * 32 + lr) * ldn + nb0 + 32 * j + 4 * lc;
  };
  f32x4v cin[2][4];
  if (ACCUM) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int it = 0; it < 4; ++it) cin[gi][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rowp(gi) + (size_t)(8 * it) * ldn));
  }
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int hh = gi >> 2, i = (gi >> 1) & 1, j = gi & 1;
    bf16x8 t[4];
#pragma unroll
    for (int mbl = 0; mbl < 2; ++mbl)
#pragma unroll
      for (int nbl = 0; nbl < 2; ++nbl) {
        const bf16x8 w = __builtin_bit_cast(bf16x8, acc[hh][2 * i + mbl][2 * j + nbl]);
        const unsigned wa = w0 ^ (unsigned)(nbl << 6);
        if (mbl == 0) tp_write(wa, w); else tp_write_o<2048>(wa, w);
      }
    tp_read_o<0>(t[0], ta.line); tp_read_o<1024>(t[1], ta.line); tp_read_o<2048>(t[2], ta.line); tp_read_o<3072>(t[3], ta.line);
    TP_WAIT4(t[0], t[1], t[2], t[3]);
    float* row = rowp(gi);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4v v = __builtin_bit_cast(f32x4v, t[it]);
      if (ACCUM) v += cin[gi & 1][it];
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(row + (size_t)(8 * it) * ldn));
    }
    if (ACCUM && gi + 2 < NG) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        cin[gi & 1][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rowp(gi + 2) + (size_t)(8 * it) * ldn));
    }
  }
}
