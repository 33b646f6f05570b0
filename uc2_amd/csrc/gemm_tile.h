// Staging / fragment helpers shared by the pipelined bf16 GEMM kernels (gemm_fast.hip, gemm_pp.hip).
#pragma once
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

#define GF_BN 128

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* glb_void_p;

// per-lane source pointer for wave-instruction `wi` (1 KiB of the operand tile) at the first k-tile.
//   !TR: tile [W rows][BK] , row = BK*2 bytes ;  TR: tile [BK k-rows][W], k-row = W*2 bytes
template <bool TR, int W, int BK>
__device__ __forceinline__ const bf16* gf_src(const bf16* __restrict__ X, int ld, int rows, int r0, int kbeg, int wi,
                                              int l) {
  if (!TR) {
    constexpr int CPR = BK / 8;                 // 16-B chunks per row (8 or 4)
    constexpr int RPI = 64 / CPR;               // rows per wave-instruction
    const int row = wi * RPI + l / CPR, cp = l % CPR;
    const int c = cp ^ ((row >> (CPR == 8 ? 1 : 2)) & (CPR - 1));
    const int gr = min(r0 + row, rows - 1);
    return X + (size_t)gr * ld + kbeg + c * 8;
  } else {
    constexpr int CPR = W / 8;                  // chunks per k-row (16 or 32)
    constexpr int RPI = 64 / CPR;               // k-rows per wave-instruction (4 or 2)
    const int krow = wi * RPI + l / CPR, cp = l % CPR;
    const int c = cp ^ ((krow & 3) << 2);
    const int col = min(r0 + c * 8, rows - 8);
    return X + (size_t)(kbeg + krow) * ld + col;
  }
}

// one MFMA operand fragment (32 rows x 16 k) for k16-step s of the tile; rbase = first row of the fragment
template <bool TR, int W, int BK>
__device__ __forceinline__ bf16x8 gf_frag(const char* lds, int rbase, int s, int lane) {
  if (!TR) {
    constexpr int CPR = BK / 8;
    const int row = rbase + (lane & 31), h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(lds + row * (BK * 2) +
                                            ((((s << 1) + h) ^ ((row >> (CPR == 8 ? 1 : 2)) & (CPR - 1))) << 4));
  } else {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = G >> 1;
    const int krow = 16 * s + 8 * h + q;                 // krow & 3 == q for both reads
    const int col = rbase + 16 * (G & 1) + 4 * pp;
    const int off = krow * (W * 2) + ((((col >> 3) ^ (q << 2))) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) short4v* lds_p;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + off + 4 * (W * 2)));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    return bf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<2>() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<9>() { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<18>() { asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }

