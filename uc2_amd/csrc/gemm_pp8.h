// Helpers of the ping-pong GEMM on e4m3 operands (gemm_pp8.hip): gemm_pp16.h's, where the MFMA operand layout differs.
//   operand fragment of v_mfma_scale_f32_16x16x128_f8f6f4: lane l holds row (or column) l & 15 of a 16-row block and the 32 bytes
//   k = 32 g .. 32 g + 31, g = l >> 4, of the 128-byte k-tile row = 16-byte chunks 2 g and 2 g + 1 of the k-contiguous LDS image
//   (any k -> (lane, register) map is right as long as A and B use the same one: the MFMA sums over k);
//   accumulator block: as v_mfma_f32_16x16x32_bf16 (lane = token row l & 15, columns 4 g + r).
#pragma once
#include "gemm_pp16.h"
typedef __attribute__((ext_vector_type(8))) int i32x8p;
typedef __attribute__((ext_vector_type(4))) int i32x4p;
// byte offset (inside a unit) of chunk 2 g of the lane's row for rows rbase ..; chunk 2 g + 1 is at off ^ 16; block blk at + blk * 2048
__device__ __forceinline__ unsigned pp8_frag_off(int rbase, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int row = rbase + i;
  return row * 128 + (((2 * g) ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ i32x8p pp8_cat(const bf16x8& lo, const bf16x8& hi) {
  const i32x4p l = __builtin_bit_cast(i32x4p, lo), h = __builtin_bit_cast(i32x4p, hi);
  return __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Epilogue arithmetic (pp_epi_compute_q of gemm_pp.h for the 16 x 16 accumulator blocks).  acc[hh][mb][nb]: 16-row block mb of
// A half hh, 16-column block nb; a 32-row output block (hh, i) is the block pair mb = 2 i, 2 i + 1.  Piece k = 4 mbl + nb of
// the lane = row 16 mbl + (lane & 15) of the 32-row block, columns 16 nb + 4 g .. + 3 = half (g & 1) of 16-byte chunk
// 2 nb + (g >> 1) of the 128-byte row in the transposition buffer: (q0 ^ (nb << 5)) + mbl * 2048.
// QOUT: also an e4m3 copy of the main output (x qs, saturating) as whole 64-byte row segments qo[A half][i][it] (row 16 it + (lane >> 2)
// of the 32-row block, bytes 16 (lane & 3) .. + 15 of the wave's 64 columns), and the running maximum of |output| in qmax
template <int EPI, bool QOUT = false>
__device__ __forceinline__ void pp8_epi_compute_q(const GemmArgs& p, const f32x4 (&acc)[2][4][4], PpOut& out, int mb0, int nb0,
                                                  int lane, const TpAddr& ta, float alpha, float qs, float& qmax) {
  const int g = lane >> 4, r15 = lane & 15, lr = lane >> 3, lc = lane & 7;
  constexpr bool has_aux = (EPI == EPI_DGELU || EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_DROPADD);
  constexpr bool two = (EPI == EPI_GELU || EPI == EPI_GELU_D);
  const bool want_cs = (EPI == EPI_DGELU || EPI == EPI_MUL) && p.aux_out != nullptr;
  const unsigned tb = ta.line - (unsigned)(lr * 128 + ((lc ^ lr) << 4));
  const unsigned q0 = tb + (unsigned)r15 * 128u + ((unsigned)((g >> 1) ^ (r15 & 7)) << 4) + 8u * (unsigned)(g & 1);
  // EPI_DROPADD (as in gemm_pp16.h): the column part of the dropout variate for the lane's four pieces of the tile; p == 0 is routed
  // to EPI_ADD by the launcher
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
      unsigned qv[8];
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
          for (int e = 0; e < 4; ++e) v[e] = acc[hh][2 * i + mbl][nb][e] * alpha;      // de-scale: 1 / (scale_a scale_b), a power of two
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
          if (QOUT) {
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            qv[k] = fp8_pack4_sat(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
          }
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
        if (QOUT) {
          // (with the e4m3 stream the second stream leaves at once, block by block: three streams held to the end do not fit the
          //  register file -- 80 spilled registers; the stores are counted by the kernel either way)
          bf16x8 pl[4];
          tp_read_o<0>(pl[0], ta.line); tp_read_o<1024>(pl[1], ta.line); tp_read_o<2048>(pl[2], ta.line); tp_read_o<3072>(pl[3], ta.line);
          TP_WAIT4(pl[0], pl[1], pl[2], pl[3]);
          bf16* a0 = reinterpret_cast<bf16*>(p.aux_out) + (size_t)(mb0 + hh * 64 + i * 32 + lr) * p.ldaux + nb0 + 8 * lc;
#pragma unroll
          for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(pl[it], reinterpret_cast<bf16x8*>(a0 + (size_t)(8 * it) * p.ldaux));
        } else {
          tp_read_o<0>(out.pre[hh][i][0], ta.line); tp_read_o<1024>(out.pre[hh][i][1], ta.line);
          tp_read_o<2048>(out.pre[hh][i][2], ta.line); tp_read_o<3072>(out.pre[hh][i][3], ta.line);
          TP_WAIT4(out.pre[hh][i][0], out.pre[hh][i][1], out.pre[hh][i][2], out.pre[hh][i][3]);
        }
      }
      TP_WAIT4(out.o[hh][i][0], out.o[hh][i][1], out.o[hh][i][2], out.o[hh][i][3]);
      if (QOUT) {
        // the e4m3 block through the same buffer (every read of it above has completed): 32 rows x 64 bytes, dword 4 nb + g of row
        // 16 mbl + r15 at 4 (nb ^ ((r15 >> 2) & 3)) + g -- conflict-free for the 32-bit writes and for the 16-byte row reads
        const unsigned wq = tb + (unsigned)r15 * 64u + 4u * (unsigned)g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned a_ = wq + (unsigned)(((k & 3) ^ ((r15 >> 2) & 3)) << 4);
          if (k < 4) asm volatile("ds_write_b32 %0, %1" :: "v"(a_), "v"(qv[k]) : "memory");
          else asm volatile("ds_write_b32 %0, %1 offset:1024" :: "v"(a_), "v"(qv[k]) : "memory");
        }
        const int qr = lane >> 2, qc = lane & 3;
        const unsigned rq = tb + (unsigned)qr * 64u + (unsigned)((qc ^ ((qr >> 2) & 3)) << 4);
        bf16x8 q0_, q1_;
        asm volatile("ds_read_b128 %0, %1" : "=v"(q0_) : "v"(rq) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q1_) : "v"(rq) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0_), "+v"(q1_) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // stored at once (64-byte row segments; 8 stores per wave and item -- the kernel's counted vmcnt knows): holding four blocks of
        // them until the other streams' stores spilled 98 registers in the GELU + gelu' instantiation
        unsigned char* d = reinterpret_cast<unsigned char*>(p.q_out) + (size_t)(mb0 + hh * 64 + i * 32 + qr) * p.ldq + nb0 + 16 * qc;
        __builtin_nontemporal_store(q0_, reinterpret_cast<bf16x8*>(d));
        __builtin_nontemporal_store(q1_, reinterpret_cast<bf16x8*>(d + (size_t)16 * p.ldq));
      }
    }
  if (EPI == EPI_DGELU || EPI == EPI_MUL) {
    if (want_cs) {                                     // (wave-uniform) one 64-lane atomic per wave and tile
      const float tot = colsum_butterfly16(cs, lane);
      const int vv = r15;                              // value index v = 4 nb + e  ->  column 16 nb + 4 g + e
      atomicAdd(reinterpret_cast<float*>(p.aux_out) + nb0 + 16 * (vv >> 2) + 4 * g + (vv & 3), tot);
    }
  }
}


