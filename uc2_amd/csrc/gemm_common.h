// Shared pieces of the GEMM kernels: argument block, per-element epilogue, bf16 tile epilogue.
#pragma once
#include "common.h"

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_ADD = 3, EPI_TANH = 4,
       EPI_GELU_D = 5, EPI_MUL = 6,       // internal to the ping-pong kernel: EPI_GELU / EPI_DGELU under GemmArgs::aux_deriv
       EPI_GELU_NOAUX = 7,                // ... and EPI_GELU without a second output stream (aux_out == NULL)
       EPI_ACC = 8,                       // ... fp32 C += tile through the line-wide partial-tile store (unsplit weight gradient)
       EPI_GROUP = 9,                     // ... one launch over the split-K items of up to four weight gradients (GemmArgs::grp)
       EPI_DROPADD = 10 };                // ... dropout(acc + bias) + aux_in: the pre-LayerNorm sum of the encoder's dense -> dropout -> add tails (uc2_gemm_drop_residual)

// One problem of a grouped weight-gradient launch (uc2_gemm_wgrad_group): dW[M,N] += A^T B over `ktiles` 64-row k-tiles,
// A = dY [rows][M] and B = X [rows][N] both k-strided; items item0 .. item0 + ntile * split - 1 of the launch, tile-major
// inside a split; fp32 partial tiles [split][M][N] at `partial`.
struct GemmProb {
  const void* A; const void* B; float* partial;
  int M, N, lda, ldb;
  int nbx, mt, col_group, per, ktiles, item0;
};
#define UC2_GEMM_MAX_GROUP 4

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias; const void* aux_in; void* aux_out;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int epi, c_f32, accumulate, split_k, atomic;
  int a_vec, b_vec;      // 16-byte vector loads allowed (alignment checked on the host)
  float* partial;        // split-K partial tiles [split][M][N] fp32 (ping-pong kernel, two-stage reduction), or null
  int skew;              // ping-pong kernel: start delay step in units of s_sleep(127) (~8k cycles) between the 4 phase groups
  // per-call plan (include/uc2_hip.h uc2_gemm): nothing about kernel selection is process-global
  int variant;           // UC2_GEMM_AUTO (-2), 99 = generic register-staged kernel, 0..9 = a specific kernel
  float* ws; size_t ws_bytes;   // caller-owned split-K workspace for THIS call (or null)
  int defer;             // leave the split-K partials in ws (the caller runs uc2_gemm_splitk_reduce)
  int diag;              // diagnostic launch mode (main loop only / epilogue only / stamps), 0 in production
  float alpha;           // accumulator scale applied before bias / epilogue (fp8 GEMM: 1 / (scale_a * scale_b)); 1 otherwise
  const float* alpha_dev; // ... or read from device memory (product of two device-side scales), NULL = use alpha
  const float* alpha_dev2;
  int col_group;         // ping-pong kernel: column tiles per L2 group of the tile order (host: largest divisor of N/256 that is <= 6)
  int aux_deriv;         // UC2_GEMM_AUX_DERIV: EPI_GELU stores gelu'(pre) (not pre) to aux_out, EPI_DGELU multiplies by aux_in as is
  int spare_cus;         // persistent kernels: CUs left without a workgroup (for a kernel running beside this one on another stream)
  int* queue;            // ping-pong kernel: caller-owned item queue (9 zeroed ints: next-item counter per XCD + exit count), or null = static partition
  int ngroup, grp_items; // EPI_GROUP: problems and total work items of the launch
  GemmProb grp[UC2_GEMM_MAX_GROUP];
  // fp8 ping-pong kernel (gemm_pp8.hip), optional: an e4m3 copy of the MAIN output for the next GEMM (delayed scaling: quantised with
  // half the scale of *q_amax_prev; max |output| accumulated into *q_amax_next; *q_amax_clear zeroed; the scale used -> *q_scale_out)
  // EPI_DROPADD: the dropout of nn.Dropout behind the dense layer, same counter-based mask as the LayerNorm kernels (common.h drop_keep4)
  unsigned drop_thresh; float drop_scale; const uint64_t* drop_seed_ptr; uint64_t drop_seed_imm;
  void* q_out; int ldq;
  const unsigned* q_amax_prev; unsigned* q_amax_next; unsigned* q_amax_clear; float* q_scale_out;
};

// ------------------------------------------------------------------------------------------
// epilogue on one element
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void epi_store(const GemmArgs& p, int m, int n, float v) {
  if (m >= p.M || n >= p.N) return;
  v *= p.alpha;
  if (p.bias) v += p.bias[n];
  const size_t ia = (size_t)m * p.ldaux + n;
  if (p.epi == EPI_GELU) {
    if (p.aux_out) reinterpret_cast<T*>(p.aux_out)[ia] = from_f<T>(p.aux_deriv ? dgelu_t<T>(v) : v);
    v = gelu_t<T>(v);
  } else if (p.epi == EPI_DGELU) {
    const float x = to_f<T>(reinterpret_cast<const T*>(p.aux_in)[ia]);
    v *= p.aux_deriv ? x : dgelu_t<T>(x);
  } else if (p.epi == EPI_ADD) {
    v += to_f<T>(reinterpret_cast<const T*>(p.aux_in)[ia]);
  } else if (p.epi == EPI_TANH) {
    v = tanh_t<T>(v);
  }
  const size_t ic = (size_t)m * p.ldc + n;
  if (p.c_f32) {
    float* c = reinterpret_cast<float*>(p.C);
    if (p.atomic & 1) atomicAdd(c + ic, v);
    else if (p.accumulate) c[ic] += v;
    else c[ic] = v;
  } else {
    T* c = reinterpret_cast<T*>(p.C);
    if (p.accumulate) v += to_f<T>(c[ic]);
    c[ic] = from_f<T>(v);
  }
}


// Write one wave's 64x64 block of the workgroup tile through the epilogue.
// TACC = true : acc holds D^T (lane = one output row m, 4 consecutive n per register group).  The block is
//               staged through a wave-private LDS area (EPI_LDS_PER_WAVE bytes, fp32, 32 rows at a time) and
//               leaves as whole 128-byte row segments, 16 bytes per lane: bias, aux reads/writes and the C
//               stores are all full-line accesses.  (Per-lane 8-byte stores straight from the accumulator
//               layout touch 32 different lines per instruction and cost more than the whole K loop at K=768.)
// TACC = false: acc holds D (one register = two 128-B row segments): f32 atomics / accumulation, direct.
// The caller must have passed a workgroup barrier after its last LDS read before calling this.
#define EPI_ROW_F32 68                       /* 64 + 4 floats of padding: conflict-free b128 rows */
#define EPI_LDS_PER_WAVE (32 * EPI_ROW_F32 * 4)

template <bool TACC>
__device__ __forceinline__ void bf16_tile_epilogue(const GemmArgs& p, const f32x16 (&acc)[2][2], int m0, int n0,
                                                   int wm, int wn, int lane, char* lds_wave) {
  // 32x32 C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int h = lane >> 5, c31 = lane & 31;
  if (TACC) {
    float* st = reinterpret_cast<float*>(lds_wave);
    const int nb = n0 + wn * 64;
    const int cg = lane & 7, rr = lane >> 3;            // store phase: lane -> (row rr + 8*it, 8 columns at 8*cg)
    const int n = nb + 8 * cg;
    const bool vec_ok = !p.c_f32 && !p.accumulate && ((p.ldc & 7) == 0) && ((p.ldaux & 7) == 0) && (n + 7 < p.N) &&
                        ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.aux_in) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.aux_out) & 15) == 0);
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // ---- accumulator layout -> LDS (fp32): row = c31, columns 32*j + 8*c + 4*h .. +3
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<float4*>(st + c31 * EPI_ROW_F32 + 32 * j + 8 * c + 4 * h) =
              make_float4(acc[i][j][4 * c], acc[i][j][4 * c + 1], acc[i][j][4 * c + 2], acc[i][j][4 * c + 3]);
      __builtin_amdgcn_wave_barrier();
      // ---- LDS -> global: 8 rows x 128 B per wave-instruction
      const int mb = m0 + wm * 64 + i * 32;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = rr + 8 * it, m = mb + r;
        const float4 lo = *reinterpret_cast<const float4*>(st + r * EPI_ROW_F32 + 8 * cg);
        const float4 hi = *reinterpret_cast<const float4*>(st + r * EPI_ROW_F32 + 8 * cg + 4);
        const float raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (m >= p.M) continue;
        if (vec_ok) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = raw[e] * p.alpha + bv[e];
          const size_t ia = (size_t)m * p.ldaux + n;
          if (p.epi == EPI_GELU) {
            if (p.aux_out) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (bf16)(p.aux_deriv ? dgelu_bf(v[e]) : v[e]);
              *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.aux_out) + ia) = o;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_bf(v[e]);
          } else if (p.epi == EPI_DGELU) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + ia);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.aux_deriv ? (float)x[e] : dgelu_bf((float)x[e]);
          } else if (p.epi == EPI_ADD) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + ia);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)x[e];
          } else if (p.epi == EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanh_bf(v[e]);
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n));   // streamed output: keep the operands in L2
        } else {
          // ragged / unaligned / fp32-output tiles: element-wise path (bias is applied inside epi_store)
#pragma unroll
          for (int e = 0; e < 8; ++e) epi_store<bf16>(p, m, n + e, raw[e]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int mb = m0 + wm * 64 + i * 32, nb = n0 + wn * 64 + j * 32;
        const int n = nb + c31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
          epi_store<bf16>(p, m, n, acc[i][j][r]);
        }
      }
  }
}

// Direct (no LDS) epilogue for one wave's 64x64 block held transposed (TACC layout: lane = output row m,
// register 4c+e = column 8c + 4*(lane>>5) + e of a 32-wide block).  v_permlane32_swap exchanges the 4-column
// groups between the two lane halves so that every lane ends up with 8 consecutive columns
// (lane half h: columns 8h..8h+7 and 16+8h..16+8h+7 of each 32-wide block), i.e. one 16-byte bf16 store per
// lane with 32 contiguous bytes per output row and instruction.  Requirements (checked by the caller):
// bf16 output, no accumulate, N % 16 == 0, ldc/ldaux % 8 == 0, 16-byte aligned C / aux pointers.
__device__ __forceinline__ void bf16_tile_epilogue_direct(const GemmArgs& p, const f32x16 (&acc)[2][2], int mb0,
                                                          int nb, int lane) {
  const int h = lane >> 5, c31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = nb + 32 * j + 16 * g + 8 * h;
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.N) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // X = group c = 2g, Y = group c = 2g+1: afterwards the low half holds (own X, partner's X), the high half (partner's Y... see above)
          // (copy the vector elements to scalars first: __builtin_bit_cast on an ext-vector element reads element 0)
          const float fx = acc[i][j][8 * g + e], fy = acc[i][j][8 * g + 4 + e];
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(fx), __float_as_uint(fy), false, false);
          v[e] = __uint_as_float(r[0]);
          v[4 + e] = __uint_as_float(r[1]);
        }
        const int m = mb0 + i * 32 + c31;
        if (m >= p.M || n >= p.N) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
        const size_t ia = (size_t)m * p.ldaux + n;
        if (p.epi == EPI_GELU) {
          if (p.aux_out) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(p.aux_deriv ? dgelu_bf(v[e]) : v[e]);
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.aux_out) + ia) = o;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_bf(v[e]);
        } else if (p.epi == EPI_DGELU) {
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + ia);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.aux_deriv ? (float)x[e] : dgelu_bf((float)x[e]);
        } else if (p.epi == EPI_ADD) {
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.aux_in) + ia);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)x[e];
        } else if (p.epi == EPI_TANH) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tanh_bf(v[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
      }
    }
}
