// HBM-bound helper kernels of the uc2 hot path: embedding lookups, the txt|img gather into the
// compact sequence, row selection for the masked heads, bias-gradient column sums, the loss
// heads (cross entropy with online softmax, KL, MSE, triplet) and casts.
// Reference lines: model/model.py:280-335 (text embeddings), :412-425 (gather), :653-657
// (masked rows), :583-596 (MLM CE), :697-732 (ITM CE), :668-688 (MRFR), :738-775 (MRC),
// model/itm.py:45-53 (triplet).
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t n, int per_thread = 1) {
  size_t g = (n + (size_t)EW_BLOCK * per_thread - 1) / ((size_t)EW_BLOCK * per_thread);
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}

// ---------------------------------------------------------------------------------------
// position ids: pos = cumsum(id != pad) * (id != pad) + pad          (model/model.py:280-290)
// ---------------------------------------------------------------------------------------
__global__ void position_ids_kernel(int B, int T, const int64_t* __restrict__ ids, int64_t pad,
                                    int64_t* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t c = 0;
  for (int t = 0; t < T; ++t) {
    const bool nz = ids[(size_t)b * T + t] != pad;
    c += nz ? 1 : 0;
    out[(size_t)b * T + t] = (nz ? c : 0) + pad;
  }
}
extern "C" int uc2_position_ids(int B, int T, const int64_t* ids, int64_t pad, int64_t* out, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  UC2_CHECK_ARG(ids && out);
  hipLaunchKernelGGL(position_ids_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, T, ids, pad, out);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// text embedding sum: out[r] = word[ids[r]] + pos[pos_ids[r]] + type[type_ids[r] or type_const]
// one wave per row, 4 elements per lane per step; tables are the fp32 master weights
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(int rows, int H, const int64_t* __restrict__ ids,
                                                        const int64_t* __restrict__ pos_ids,
                                                        const int64_t* __restrict__ type_ids, int type_const,
                                                        const float* __restrict__ word, const float* __restrict__ pos,
                                                        const float* __restrict__ type, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* w = word + (size_t)ids[r] * H;
  const float* p = pos + (size_t)pos_ids[r] * H;
  const float* ty = type + (size_t)(type_ids ? type_ids[r] : type_const) * H;
  for (int c = lane * 4; c < H; c += 256) {
    float a[4], b[4], d[4], o[4];
    Vec4<float>::load(w + c, a);
    Vec4<float>::load(p + c, b);
    Vec4<float>::load(ty + c, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] + b[e] + d[e];
    Vec4<T>::store(out + (size_t)r * H + c, o);
  }
}
// backward: scatter-add rows into the fp32 gradient tables (word, pos; type too when per-row ids)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(int rows, int H, const int64_t* __restrict__ ids,
                                                        const int64_t* __restrict__ pos_ids,
                                                        const int64_t* __restrict__ type_ids,
                                                        const T* __restrict__ dpre, float* __restrict__ dword,
                                                        float* __restrict__ dpos, float* __restrict__ dtype,
                                                        int64_t word_pad, int64_t pos_pad) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  // nn.Embedding(padding_idx=...) rows receive no gradient (model/model.py:296-298)
  float* w = (dword && ids[r] != word_pad) ? dword + (size_t)ids[r] * H : nullptr;
  float* p = (dpos && pos_ids[r] != pos_pad) ? dpos + (size_t)pos_ids[r] * H : nullptr;
  float* ty = (dtype && type_ids) ? dtype + (size_t)type_ids[r] * H : nullptr;
  for (int c = lane; c < H; c += 64) {           // one dword per lane: 256 contiguous bytes per wave-instruction
    const float g = to_f<T>(dpre[(size_t)r * H + c]);
    if (w) atomicAdd(w + c, g);
    if (p) atomicAdd(p + c, g);
    if (ty) atomicAdd(ty + c, g);
  }
}
// The same for ids laid out [B, T] (the embedding call of model/model.py:296-333): a wave walks ONE sequence position t down a chunk of
// the batch.  Position ids (and type ids) then repeat from row to row -- arange + offset per sequence, one type per modality -- so
// their gradient rows are accumulated in registers and flushed with one atomic per column when the id changes and at the end of
// the chunk: 6144 pairs x 60 tokens piled 368 k atomic adds onto each element of ~60 position rows and ONE type row in
// embed_bwd_kernel (1.88 ms per step).  Any id pattern is handled (a change of id is a flush), only the speed assumes repetition.
// The word rows stay direct atomics (distinct ids, no pile-up).
template <typename T, int NC>                         // NC = H / 64 columns per lane (column = lane + 64 k: one dword per lane and
__global__ __launch_bounds__(256) void embed_bwd_seq_kernel(int B, int Tn, int H, int bchunk, const int64_t* __restrict__ ids,   // instruction, 256 contiguous bytes per atomic)
                                                            const int64_t* __restrict__ pos_ids, const int64_t* __restrict__ type_ids,
                                                            const T* __restrict__ dpre, float* __restrict__ dword,
                                                            float* __restrict__ dpos, float* __restrict__ dtype, int64_t word_pad,
                                                            int64_t pos_pad) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nchunk = (B + bchunk - 1) / bchunk;
  if (w >= Tn * nchunk) return;
  const int t = w % Tn, b0 = (w / Tn) * bchunk, b1 = min(B, b0 + bchunk);
  float ap[NC], at[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { ap[k] = 0.f; at[k] = 0.f; }
  int64_t curp = -1, curt = -1;                        // ids whose sums are in ap / at (-1: none)
  auto flush = [&](float (&a)[NC], float* tab, int64_t id) {
    if (id < 0) return;
#pragma unroll
    for (int k = 0; k < NC; ++k) { atomicAdd(tab + (size_t)id * H + lane + 64 * k, a[k]); a[k] = 0.f; }
  };
  for (int b = b0; b < b1; ++b) {
    const size_t r = (size_t)b * Tn + t;
    const int64_t wid = ids[r];
    const int64_t pid = (dpos && pos_ids[r] != pos_pad) ? pos_ids[r] : -1;
    const int64_t tid = (dtype && type_ids) ? type_ids[r] : -1;
    if (pid != curp) { flush(ap, dpos, curp); curp = pid; }
    if (tid != curt) { flush(at, dtype, curt); curt = tid; }
    float* wrow = (dword && wid != word_pad) ? dword + (size_t)wid * H : nullptr;
    float g[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) g[k] = to_f<T>(dpre[r * H + lane + 64 * k]);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (wrow) atomicAdd(wrow + lane + 64 * k, g[k]);
      if (pid >= 0) ap[k] += g[k];                     // (a padding position / no type table: nothing to sum)
      if (tid >= 0) at[k] += g[k];
    }
  }
  flush(ap, dpos, curp);
  flush(at, dtype, curt);
}

extern "C" int uc2_embed_fwd(int dtype, int rows, int H, const int64_t* ids, const int64_t* pos_ids,
                             const int64_t* type_ids, int type_const, const float* word, const float* pos,
                             const float* type, void* out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(H > 0 && (H % 4) == 0);
  if (rows <= 0) return 0;
  UC2_CHECK_ARG(ids && pos_ids && word && pos && type && out);
  dim3 grid((rows + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, rows, H, ids, pos_ids, type_ids, type_const, word, pos, type, (float*)out);
  else hipLaunchKernelGGL(embed_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, rows, H, ids, pos_ids, type_ids, type_const, word, pos, type, (bf16*)out);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_embed_bwd(int dtype, int rows, int H, const int64_t* ids, const int64_t* pos_ids,
                             const int64_t* type_ids, const void* dpre, float* dword, float* dpos, float* dtype_tab,
                             int64_t word_pad, int64_t pos_pad, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (rows <= 0) return 0;
  UC2_CHECK_ARG(ids && pos_ids && dpre);
  dim3 grid((rows + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, rows, H, ids, pos_ids, type_ids, (const float*)dpre, dword, dpos, dtype_tab, word_pad, pos_pad);
  else hipLaunchKernelGGL(embed_bwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, rows, H, ids, pos_ids, type_ids, (const bf16*)dpre, dword, dpos, dtype_tab, word_pad, pos_pad);
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_embed_bwd_seq(int dtype, int B, int T, int H, const int64_t* ids, const int64_t* pos_ids,
                                 const int64_t* type_ids, const void* dpre, float* dword, float* dpos, float* dtype_tab,
                                 int64_t word_pad, int64_t pos_pad, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (B <= 0 || T <= 0) return 0;
  UC2_CHECK_ARG(ids && pos_ids && dpre);
  if (H <= 0 || (H % 64) != 0 || H > 1024) return -2;        // (the caller falls back to uc2_embed_bwd)
  // batch chunk per wave: long enough to amortise the flush (H atomics per table), short enough for ~8 waves per SIMD over the chip
  int bchunk = (int)(((long long)B * T + 8191) / 8192);
  if (bchunk < 8) bchunk = 8;
  if (bchunk > B) bchunk = B;
  const int nchunk = (B + bchunk - 1) / bchunk;
  const int waves = T * nchunk;
  dim3 grid((waves + 3) / 4);
#define EB_LAUNCH(TT, NCC) hipLaunchKernelGGL((embed_bwd_seq_kernel<TT, NCC>), grid, dim3(256), 0, (hipStream_t)stream, B, T, H, bchunk, ids, pos_ids, type_ids, (const TT*)dpre, dword, dpos, dtype_tab, word_pad, pos_pad)
#define EB_NC(TT) do { switch (H / 64) { case 1: EB_LAUNCH(TT, 1); break; case 2: EB_LAUNCH(TT, 2); break; case 4: EB_LAUNCH(TT, 4); break; case 8: EB_LAUNCH(TT, 8); break; \
                                         case 12: EB_LAUNCH(TT, 12); break; case 16: EB_LAUNCH(TT, 16); break; default: return -2; } } while (0)
  if (dtype == 0) EB_NC(float); else EB_NC(bf16);
#undef EB_NC
#undef EB_LAUNCH
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// gather along dim 1: out[b, j, :] = src[b, index[b, j], :]      (model/model.py:420-425)
// backward is a deterministic per-source scan (index may repeat in the padded tail)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(int B, int S, int L, int H, const T* __restrict__ src,
                                                         const int64_t* __restrict__ index, T* __restrict__ out,
                                                         const T* __restrict__ src2 = nullptr, int S1 = 0) {
  // (src2 != NULL: the source is the concatenation [src (S1 rows) | src2 (S - S1 rows)] along dim 1 -- text and image
  //  embeddings, model/model.py:412-425 -- read in place, the concatenated tensor never exists)
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B * L) return;
  const int b = r / L;
  const int64_t si = index[r];
  const T* s = (src2 && si >= S1) ? src2 + ((size_t)b * (S - S1) + (si - S1)) * H
                                  : src + ((size_t)b * (src2 ? S1 : S) + si) * H;
  for (int c = lane * 4; c < H; c += 256) {
    float v[4];
    Vec4<T>::load(s + c, v);
    Vec4<T>::store(out + (size_t)r * H + c, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_kernel(int B, int S, int L, int H, const T* __restrict__ dout,
                                                         const int64_t* __restrict__ index, T* __restrict__ dsrc,
                                                         T* __restrict__ dsrc2 = nullptr, int S1 = 0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);          // r = b*S + s
  if (r >= B * S) return;
  const int b = r / S, s = r - b * S;
  // the positions j with index[b][j] == s, found 64 at a time (one per lane, ballot) instead of every lane scanning
  // all L entries for every column chunk; up to 4 column chunks (H <= 1024) are accumulated in registers per match
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  for (int cbase = 0; cbase < H; cbase += 1024) {
    for (int j0 = 0; j0 < L; j0 += 64) {
      const int j = j0 + lane;
      unsigned long long m = __ballot(j < L && index[(size_t)b * L + j] == s);
      while (m) {
        const int jj = j0 + __builtin_ctzll(m);
        m &= m - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = cbase + i * 256 + lane * 4;
          if (c < H) {
            float v[4];
            Vec4<T>::load(dout + ((size_t)b * L + jj) * H + c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] += v[e];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = cbase + i * 256 + lane * 4;
      if (c < H) {
        T* d = (dsrc2 && s >= S1) ? dsrc2 + ((size_t)b * (S - S1) + (s - S1)) * H
                                  : dsrc + ((size_t)b * (dsrc2 ? S1 : S) + s) * H;
        Vec4<T>::store(d + c, acc[i]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    }
  }
}
extern "C" int uc2_gather_rows_fwd(int dtype, int B, int S, int L, int H, const void* src, const int64_t* index,
                                   void* out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG((H % 4) == 0);
  if (B * L <= 0) return 0;
  UC2_CHECK_ARG(src && index && out);
  dim3 grid((B * L + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(gather_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, B, S, L, H, (const float*)src, index, (float*)out);
  else hipLaunchKernelGGL(gather_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, B, S, L, H, (const bf16*)src, index, (bf16*)out);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_gather_rows_bwd(int dtype, int B, int S, int L, int H, const void* dout, const int64_t* index,
                                   void* dsrc, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG((H % 4) == 0);
  if (B * S <= 0) return 0;
  UC2_CHECK_ARG(dout && index && dsrc);
  dim3 grid((B * S + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(gather_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, B, S, L, H, (const float*)dout, index, (float*)dsrc);
  else hipLaunchKernelGGL(gather_bwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, B, S, L, H, (const bf16*)dout, index, (bf16*)dsrc);
  UC2_LAUNCH_CHECK();
  return 0;
}

// the same over the concatenation [src1 [B,S1,H] | src2 [B,S2,H]] along dim 1 without building it: forward reads
// the two sources in place, backward writes the two gradients as separate contiguous tensors
extern "C" int uc2_gather_rows2_fwd(int dtype, int B, int S1, int S2, int L, int H, const void* src1, const void* src2,
                                    const int64_t* index, void* out, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG((H % 4) == 0 && S1 >= 0 && S2 >= 0);
  if (B * L <= 0) return 0;
  UC2_CHECK_ARG(src1 && src2 && index && out);
  dim3 grid((B * L + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(gather_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, B, S1 + S2, L, H, (const float*)src1, index, (float*)out, (const float*)src2, S1);
  else hipLaunchKernelGGL(gather_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, B, S1 + S2, L, H, (const bf16*)src1, index, (bf16*)out, (const bf16*)src2, S1);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_gather_rows2_bwd(int dtype, int B, int S1, int S2, int L, int H, const void* dout, const int64_t* index,
                                    void* dsrc1, void* dsrc2, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG((H % 4) == 0 && S1 >= 0 && S2 >= 0);
  if (B * (S1 + S2) <= 0) return 0;
  UC2_CHECK_ARG(dout && index && dsrc1 && dsrc2);
  dim3 grid((B * (S1 + S2) + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(gather_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, B, S1 + S2, L, H, (const float*)dout, index, (float*)dsrc1, (float*)dsrc2, S1);
  else hipLaunchKernelGGL(gather_bwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, B, S1 + S2, L, H, (const bf16*)dout, index, (bf16*)dsrc1, (bf16*)dsrc2, S1);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// row selection for the masked heads: out[i] = src[rows[i]]  /  dsrc[rows[i]] = dsel[i]
// (rows are unique: they come from a boolean mask; dsrc must be zero-filled by the caller)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void select_rows_kernel(int n, int H, const T* __restrict__ src, int ld_src,
                                                          const int64_t* __restrict__ rows, T* __restrict__ dst,
                                                          int ld_dst, int scatter) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t r = rows[i];
  // A negative index is a padding entry of a fixed-size index list (torch.nonzero_static with a count hint larger than the number
  // of set entries, uc2_amd/model/model.py::_masked_rows): gathered as a zero row, skipped when scattering -- never dereferenced.
  if (r < 0 && scatter) return;
  const T* s = scatter ? src + (size_t)i * ld_src : src + (size_t)(r < 0 ? 0 : r) * ld_src;
  T* d = scatter ? dst + (size_t)r * ld_dst : dst + (size_t)i * ld_dst;
  for (int c = lane * 4; c < H; c += 256) {
    float v[4];
    Vec4<T>::load(s + c, v);
    if (r < 0) { v[0] = v[1] = v[2] = v[3] = 0.f; }
    if (scatter == 2) {                 // accumulate: dst[rows[i]] += src[i]  (rows unique)
      float o[4];
      Vec4<T>::load(d + c, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    Vec4<T>::store(d + c, v);
  }
}
extern "C" int uc2_select_rows(int dtype, int n, int H, const void* src, int ld_src, const int64_t* rows, void* dst,
                               int ld_dst, int scatter, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG((H % 4) == 0 && (ld_src % 4) == 0 && (ld_dst % 4) == 0);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(src && rows && dst);
  dim3 grid((n + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(select_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, n, H, (const float*)src, ld_src, rows, (float*)dst, ld_dst, scatter);
  else hipLaunchKernelGGL(select_rows_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, n, H, (const bf16*)src, ld_src, rows, (bf16*)dst, ld_dst, scatter);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// rank of a target among candidates (retrieval recall, reference eval/itm.py:6-53 and itm.py:461-470): for query q
//   rank[q] = #{c : s(q,c) > s(q,t_q)} + #{c < t_q : s(q,c) == s(q,t_q)},   s(q,c) = scores[off[q] + c * stride]
// i.e. the position of the target in a stable descending sort -- recall@k = mean(rank < k) -- without materialising
// a top-k.  One wave per query; scores fp16 (the reference's score matrix dtype), bf16 or fp32.
// ---------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float rk_f(T v) { return (float)v; }
template <typename T>
__global__ __launch_bounds__(256) void rank_of_target_kernel(int nq, int nc, const T* __restrict__ scores,
                                                             const int64_t* __restrict__ off, int64_t stride,
                                                             const int64_t* __restrict__ target, int32_t* __restrict__ rank) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const T* s = scores + off[q];
  const int64_t t = target[q];
  const float st = rk_f<T>(s[t * stride]);
  int cnt = 0;
  for (int c = lane; c < nc; c += 64) {
    const float v = rk_f<T>(s[(int64_t)c * stride]);
    cnt += (v > st) || (v == st && c < t);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) rank[q] = cnt;
}
extern "C" int uc2_rank_of_target(int dtype, int nq, int nc, const void* scores, const int64_t* off, int64_t stride,
                                  const int64_t* target, int32_t* rank, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1 || dtype == 2);
  if (nq <= 0) return 0;
  UC2_CHECK_ARG(nc > 0 && scores && off && target && rank);
  dim3 grid((nq + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(rank_of_target_kernel<float>, grid, dim3(256), 0, st, nq, nc, (const float*)scores, off, stride, target, rank);
  else if (dtype == 1) hipLaunchKernelGGL(rank_of_target_kernel<bf16>, grid, dim3(256), 0, st, nq, nc, (const bf16*)scores, off, stride, target, rank);
  else hipLaunchKernelGGL(rank_of_target_kernel<_Float16>, grid, dim3(256), 0, st, nq, nc, (const _Float16*)scores, off, stride, target, rank);
  UC2_LAUNCH_CHECK();
  return 0;
}

// element gather / scatter-add on an fp32 vector: mode 0: dst[i] = src[idx[i]]; mode 1: dst[idx[i]] += src[i]
// (idx unique) -- bias entries of a column subset of the tied decoder (model/model.py:639-642)
__global__ void gather_f32_kernel(int n, const float* __restrict__ src, const int64_t* __restrict__ idx,
                                  float* __restrict__ dst, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) dst[i] = src[idx[i]];
  else dst[idx[i]] += src[i];
}
extern "C" int uc2_gather_f32(int n, const float* src, const int64_t* idx, float* dst, int mode, void* stream) {
  UC2_CHECK_ARG(mode == 0 || mode == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(src && idx && dst);
  hipLaunchKernelGGL(gather_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, src, idx, dst, mode);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// column sum (bias gradients): out[n] += sum_{m : rowmask[m]} X[m, n]   (rowmask optional)
// ---------------------------------------------------------------------------------------
// One wave reads 256 contiguous columns of a row (4 per lane, 8-byte loads for bf16); the 4 waves of a
// workgroup take 4 different rows per step; partials are combined through LDS, one atomic per column
// per workgroup.  HBM-bound: algorithmic bytes = M * N * sizeof(T).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const T* __restrict__ X, int ldx,
                                                     const uint8_t* __restrict__ rowmask, float* __restrict__ out,
                                                     int rows_per_blk, int vec) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = blockIdx.x * 256 + lane * 4;
  const int m0 = blockIdx.y * rows_per_blk, m1 = min(M, m0 + rows_per_blk);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    if (vec && n + 3 < N) {
      int m = m0 + wv;
      for (; m + 4 < m1; m += 8) {                 // two rows in flight per wave
        float v[4], u[4];
        const bool k0 = !rowmask || rowmask[m], k1 = !rowmask || rowmask[m + 4];
        Vec4<T>::load(X + (size_t)m * ldx + n, v);
        Vec4<T>::load(X + (size_t)(m + 4) * ldx + n, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (k0 ? v[e] : 0.f) + (k1 ? u[e] : 0.f);
      }
      for (; m < m1; m += 4) {
        if (rowmask && !rowmask[m]) continue;
        float v[4];
        Vec4<T>::load(X + (size_t)m * ldx + n, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
      }
    } else {
      for (int m = m0 + wv; m < m1; m += 4) {
        if (rowmask && !rowmask[m]) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) s[e] += to_f<T>(X[(size_t)m * ldx + n + e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][lane * 4 + e] = s[e];
  __syncthreads();
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col < N) atomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int uc2_colsum_accum(int dtype, int M, int N, const void* X, int ldx, const uint8_t* rowmask, float* out,
                                void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (M <= 0 || N <= 0) return 0;
  UC2_CHECK_ARG(X && out);
  const int colblk = (N + 255) / 256;
  int splits = (M + 63) / 64;
  const int want = (2048 + colblk - 1) / colblk;         // ~8 workgroups per CU in total
  if (splits > want) splits = want;
  if (splits < 1) splits = 1;
  const int rpb = ((M + splits - 1) / splits + 3) / 4 * 4;
  dim3 grid(colblk, (M + rpb - 1) / rpb);
  const int esz = dtype == 0 ? 4 : 2;
  const int vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & (4 * esz - 1)) == 0);
  if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, M, N, (const float*)X, ldx, rowmask, out, rpb, vec);
  else hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, M, N, (const bf16*)X, ldx, rowmask, out, rpb, vec);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// out[r] = a[r] + b[r] + (rowmask[r] ? vec : 0)     (b, vec, rowmask optional; vec is fp32 [H])
// image embeddings: feat + mask_embedding (model/model.py:355-356), ti + tp + type row (:360)
// ---------------------------------------------------------------------------------------
template <typename TI, typename T>
__global__ __launch_bounds__(256) void add_rowvec_kernel(int rows, int H, const TI* __restrict__ a,
                                                         const T* __restrict__ b, const float* __restrict__ vec,
                                                         const uint8_t* __restrict__ rowmask, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bool use_vec = vec && (!rowmask || rowmask[r]);
  for (int c = lane * 4; c < H; c += 256) {
    float v[4];
    Vec4<TI>::load(a + (size_t)r * H + c, v);
    if (b) {
      float w[4];
      Vec4<T>::load(b + (size_t)r * H + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    if (use_vec) {
      float w[4];
      Vec4<float>::load(vec + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    Vec4<T>::store(out + (size_t)r * H + c, v);
  }
}
extern "C" int uc2_add_rowvec(int a_dtype, int dtype, int rows, int H, const void* a, const void* b, const float* vec,
                              const uint8_t* rowmask, void* out, void* stream) {
  UC2_CHECK_ARG((a_dtype == 0 || a_dtype == 1) && (dtype == 0 || dtype == 1));
  UC2_CHECK_ARG((H % 4) == 0);
  if (rows <= 0) return 0;
  UC2_CHECK_ARG(a && out);
  dim3 grid((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == 0 && dtype == 0) hipLaunchKernelGGL((add_rowvec_kernel<float, float>), grid, dim3(256), 0, st, rows, H, (const float*)a, (const float*)b, vec, rowmask, (float*)out);
  else if (a_dtype == 0 && dtype == 1) hipLaunchKernelGGL((add_rowvec_kernel<float, bf16>), grid, dim3(256), 0, st, rows, H, (const float*)a, (const bf16*)b, vec, rowmask, (bf16*)out);
  else if (a_dtype == 1 && dtype == 1) hipLaunchKernelGGL((add_rowvec_kernel<bf16, bf16>), grid, dim3(256), 0, st, rows, H, (const bf16*)a, (const bf16*)b, vec, rowmask, (bf16*)out);
  else hipLaunchKernelGGL((add_rowvec_kernel<bf16, float>), grid, dim3(256), 0, st, rows, H, (const bf16*)a, (const float*)b, vec, rowmask, (float*)out);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// cross entropy over rows of logits (F.cross_entropy(reduction='none'), ignore_index):
// one 256-thread workgroup per row, online (max, sum) so the row is read once; argmax is a
// side output (first index of the maximum).  Backward rewrites logits in place:
//   dlogits = (softmax - onehot) * gout[row]     (zero for ignored rows)
// ---------------------------------------------------------------------------------------
// online-softmax merge of (max, sum, first argmax) pairs
__device__ __forceinline__ void ce_merge(float& m, float& s, int& am, float m2, float s2, int a2) {
  const float mn = fmaxf(m, m2);
  const float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
  const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mn);
  am = (m2 > m || (m2 == m && a2 < am)) ? a2 : am;
  s = sa + sb; m = mn;
}
// 16 bytes per lane and load: VEC = 8 bf16 / 4 fp32 consecutive columns (the 250 002-column MLM rows are HBM-bound:
// scalar 2-byte loads ran at 1.8 TB/s); rows must start 16-byte aligned (ld a multiple of VEC), else VEC = 1
template <typename T> struct CeVec;
template <> struct CeVec<float> { static constexpr int N = 4; };
template <> struct CeVec<bf16> { static constexpr int N = 8; };
template <typename T, int VEC>
__device__ __forceinline__ void ce_load(const T* p, float (&v)[VEC]) {
  if constexpr (VEC == 1) { v[0] = to_f<T>(p[0]); }
  else if constexpr (sizeof(T) == 4) { Vec4<float>::load(reinterpret_cast<const float*>(p), reinterpret_cast<float (&)[4]>(v)); }
  else {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
  }
}
template <typename T, int VEC>
__device__ __forceinline__ void ce_store(T* p, const float (&v)[VEC]) {
  if constexpr (VEC == 1) { p[0] = from_f<T>(v[0]); }
  else if constexpr (sizeof(T) == 4) { Vec4<float>::store(reinterpret_cast<float*>(p), reinterpret_cast<const float (&)[4]>(v)); }
  else {
    bf16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = x;
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_fwd_kernel(int V, const T* __restrict__ logits, int ld,
                                                     const int64_t* __restrict__ labels, int64_t ignore_index,
                                                     float* __restrict__ loss, float* __restrict__ lse_o,
                                                     int64_t* __restrict__ argmax_o) {
  __shared__ float sm_m[4], sm_s[4];
  __shared__ int sm_i[4];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const T* x = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f;
  int am = 0x7fffffff;
  for (int c0 = t * VEC; c0 < V; c0 += 256 * VEC) {
    float v[VEC];
    ce_load<T, VEC>(x + c0, v);                      // (the row is padded to a multiple of VEC columns by its leading dimension)
    float cm = -INFINITY;
    int ci = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (c0 + e >= V) v[e] = -INFINITY;
      if (v[e] > cm) { cm = v[e]; ci = c0 + e; }       // strict >: the first maximum wins (torch.argmax tie rule)
    }
    if (cm > m) { s *= __expf(m - cm); m = cm; am = ci; }     // exp(-inf) = 0 the first time
    if (m > -INFINITY) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += __expf(v[e] - m);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ce_merge(m, s, am, __shfl_xor(m, o), __shfl_xor(s, o), __shfl_xor(am, o));
  if (lane == 0) { sm_m[wv] = m; sm_s[wv] = s; sm_i[wv] = am; }
  __syncthreads();
  if (t == 0) {
    float M = sm_m[0], S = sm_s[0];
    int A = sm_i[0];
    for (int i = 1; i < 4; ++i) ce_merge(M, S, A, sm_m[i], sm_s[i], sm_i[i]);
    const float lse = M + __logf(S);
    if (lse_o) lse_o[row] = lse;
    if (argmax_o) argmax_o[row] = A;
    if (loss) {
      const int64_t lab = labels[row];
      loss[row] = (lab == ignore_index) ? 0.f : lse - to_f<T>(x[lab]);
    }
  }
}
// in place: logits -> dlogits = (softmax - onehot) * g; the padding columns [V, ld) of a row are set to zero (they feed
// the decoder's weight-gradient and input-gradient GEMMs, which run over whole padded rows)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_bwd_kernel(int V, T* __restrict__ logits, int ld,
                                                     const int64_t* __restrict__ labels, int64_t ignore_index,
                                                     const float* __restrict__ lse, const float* __restrict__ gout) {
  const int row = blockIdx.y;
  const int64_t lab = labels[row];
  const float g = (lab == ignore_index) ? 0.f : gout[row];
  const float l = lse[row];
  T* x = logits + (size_t)row * ld;
  const int cend = VEC == 1 ? V : ld;
  for (int c0 = (blockIdx.x * 256 + threadIdx.x) * VEC; c0 < cend; c0 += gridDim.x * 256 * VEC) {
    float v[VEC];
    ce_load<T, VEC>(x + c0, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = c0 + e;
      v[e] = (c < V && g != 0.f) ? (__expf(v[e] - l) - (c == lab ? 1.f : 0.f)) * g : 0.f;
    }
    ce_store<T, VEC>(x + c0, v);
  }
  if (VEC == 1) {                                    // unaligned rows: scalar tail zeroing
    for (int c = V + blockIdx.x * 256 + threadIdx.x; c < ld; c += gridDim.x * 256) x[c] = from_f<T>(0.f);
  }
}
// The same with the column sums of dlogits (the decoder-bias gradient) produced on the way: given lse the backward is
// element-wise, so it can be tiled by column strips -- a workgroup owns 32 * VEC columns (32 column vectors x 8 row lanes)
// of a slice of the rows, keeps per-column partial sums in registers and adds them to dbias once (gridDim.y atomics per
// column).  Saves the separate column-sum pass over the [rows, 250 112] buffer (4.6 GB per 1024 pairs).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_bwd_colsum_kernel(int n, int V, T* __restrict__ logits, int ld,
                                                            const int64_t* __restrict__ labels, int64_t ignore_index,
                                                            const float* __restrict__ lse, const float* __restrict__ gout,
                                                            float* __restrict__ dbias, int ncol_bias) {
  __shared__ float red[8][32 * VEC + 1];
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + cv) * VEC;
  const int rows_per = (n + gridDim.y - 1) / gridDim.y;
  const int r_beg = blockIdx.y * rows_per, r_end = min(n, r_beg + rows_per);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (c0 < ld) {
    constexpr int U = 4;                               // rows in flight per thread
    for (int row0 = r_beg + rl; row0 < r_end; row0 += 8 * U) {
      float v[U][VEC], g[U], l[U];
      int64_t lab[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + 8 * u, r_end - 1);
        ce_load<T, VEC>(logits + (size_t)row * ld + c0, v[u]);
        lab[u] = labels[row];
        g[u] = (lab[u] == ignore_index) ? 0.f : gout[row];
        l[u] = lse[row];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + 8 * u;
        if (row >= r_end) break;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int c = c0 + e;
          v[u][e] = (c < V && g[u] != 0.f) ? (__expf(v[u][e] - l[u]) - (c == lab[u] ? 1.f : 0.f)) * g[u] : 0.f;
          acc[e] += v[u][e];
        }
        ce_store<T, VEC>(logits + (size_t)row * ld + c0, v[u]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[rl][cv * VEC + e] = acc[e];
  __syncthreads();
  for (int t = threadIdx.x; t < 32 * VEC; t += 256) {
    const int c = blockIdx.x * 32 * VEC + t;
    if (c < ncol_bias) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r][t];
      atomicAdd(dbias + c, sum);
    }
  }
}
template <typename T> static bool ce_vec_ok(const void* p, int ld) {
  return (ld % CeVec<T>::N) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
extern "C" int uc2_ce_fwd(int dtype, int n, int V, const void* logits, int ld, const int64_t* labels, int64_t ignore_index,
                          float* loss, float* lse, int64_t* argmax, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(V > 0 && ld >= V && logits && (labels || !loss));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    if (ce_vec_ok<float>(logits, ld)) hipLaunchKernelGGL((ce_fwd_kernel<float, 4>), dim3(n), dim3(256), 0, st, V, (const float*)logits, ld, labels, ignore_index, loss, lse, argmax);
    else hipLaunchKernelGGL((ce_fwd_kernel<float, 1>), dim3(n), dim3(256), 0, st, V, (const float*)logits, ld, labels, ignore_index, loss, lse, argmax);
  } else {
    if (ce_vec_ok<bf16>(logits, ld)) hipLaunchKernelGGL((ce_fwd_kernel<bf16, 8>), dim3(n), dim3(256), 0, st, V, (const bf16*)logits, ld, labels, ignore_index, loss, lse, argmax);
    else hipLaunchKernelGGL((ce_fwd_kernel<bf16, 1>), dim3(n), dim3(256), 0, st, V, (const bf16*)logits, ld, labels, ignore_index, loss, lse, argmax);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_ce_bwd(int dtype, int n, int V, void* logits, int ld, const int64_t* labels, int64_t ignore_index,
                          const float* lse, const float* gout, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(V > 0 && ld >= V && logits && labels && lse && gout);
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == 0 ? (ce_vec_ok<float>(logits, ld) ? 4 : 1) : (ce_vec_ok<bf16>(logits, ld) ? 8 : 1);
  int gx = (ld / vec + 255) / 256;
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  dim3 grid(gx, n);
  if (dtype == 0) {
    if (vec == 4) hipLaunchKernelGGL((ce_bwd_kernel<float, 4>), grid, dim3(256), 0, st, V, (float*)logits, ld, labels, ignore_index, lse, gout);
    else hipLaunchKernelGGL((ce_bwd_kernel<float, 1>), grid, dim3(256), 0, st, V, (float*)logits, ld, labels, ignore_index, lse, gout);
  } else {
    if (vec == 8) hipLaunchKernelGGL((ce_bwd_kernel<bf16, 8>), grid, dim3(256), 0, st, V, (bf16*)logits, ld, labels, ignore_index, lse, gout);
    else hipLaunchKernelGGL((ce_bwd_kernel<bf16, 1>), grid, dim3(256), 0, st, V, (bf16*)logits, ld, labels, ignore_index, lse, gout);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}
// dbias[0 .. ncol_bias) += column sums of dlogits (ncol_bias = V, or ld when the bias vector is padded like the rows).
// Needs the vectorised layout (ld % 8 == 0 for bf16 / % 4 for fp32, 16-byte aligned rows): returns -2 otherwise so that
// the caller can run uc2_ce_bwd + uc2_colsum_accum instead.
extern "C" int uc2_ce_bwd_colsum(int dtype, int n, int V, void* logits, int ld, const int64_t* labels, int64_t ignore_index,
                                 const float* lse, const float* gout, float* dbias, int ncol_bias, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(V > 0 && ld >= V && logits && labels && lse && gout && dbias && ncol_bias >= V && ncol_bias <= ld);
  const bool ok = dtype == 0 ? ce_vec_ok<float>(logits, ld) : ce_vec_ok<bf16>(logits, ld);
  if (!ok) return -2;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == 0 ? 4 : 8;
  const int gx = (ld + 32 * vec - 1) / (32 * vec);
  int gy = (2048 + gx - 1) / gx;                     // >= 2048 workgroups, each with >= 64 rows
  if (gy > (n + 63) / 64) gy = (n + 63) / 64;
  if (gy < 1) gy = 1;
  dim3 grid(gx, gy);
  if (dtype == 0) hipLaunchKernelGGL((ce_bwd_colsum_kernel<float, 4>), grid, dim3(256), 0, st, n, V, (float*)logits, ld, labels, ignore_index, lse, gout, dbias, ncol_bias);
  else hipLaunchKernelGGL((ce_bwd_colsum_kernel<bf16, 8>), grid, dim3(256), 0, st, n, V, (bf16*)logits, ld, labels, ignore_index, lse, gout, dbias, ncol_bias);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// KL(target || softmax(pred)) per element (F.kl_div(log_softmax(pred), target, 'none')),
// model/model.py:764-768.  fwd needs lse per row (from ce_fwd with loss=null).
//   loss[i,j] = t * (log t - (pred - lse))   (0 where t == 0)
//   dpred[i,k] = p_k * sum_j g_j t_j - g_k t_k
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kl_fwd_kernel(int V, const T* __restrict__ pred, int ld,
                                                     const float* __restrict__ target, const float* __restrict__ lse,
                                                     float* __restrict__ loss) {
  const int row = blockIdx.y;
  const float l = lse[row];
  for (int c = blockIdx.x * 256 + threadIdx.x; c < V; c += gridDim.x * 256) {
    const float t = target[(size_t)row * V + c];
    const float logp = to_f<T>(pred[(size_t)row * ld + c]) - l;
    loss[(size_t)row * V + c] = (t > 0.f) ? t * (__logf(t) - logp) : 0.f;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void kl_bwd_kernel(int V, const T* __restrict__ pred, int ld,
                                                     const float* __restrict__ target, const float* __restrict__ lse,
                                                     const float* __restrict__ gout, T* __restrict__ dpred) {
  __shared__ float red[4];
  const int row = blockIdx.x, t = threadIdx.x;
  float s = 0.f;
  for (int c = t; c < V; c += 256) s += gout[(size_t)row * V + c] * target[(size_t)row * V + c];
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float l = lse[row];
  for (int c = t; c < V; c += 256) {
    const float p = __expf(to_f<T>(pred[(size_t)row * ld + c]) - l);
    dpred[(size_t)row * ld + c] = from_f<T>(p * tot - gout[(size_t)row * V + c] * target[(size_t)row * V + c]);
  }
}
extern "C" int uc2_kl_fwd(int dtype, int n, int V, const void* pred, int ld, const float* target, const float* lse,
                          float* loss, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(pred && target && lse && loss);
  dim3 grid(min(8, (V + 255) / 256), n);
  if (dtype == 0) hipLaunchKernelGGL(kl_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, V, (const float*)pred, ld, target, lse, loss);
  else hipLaunchKernelGGL(kl_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, V, (const bf16*)pred, ld, target, lse, loss);
  UC2_LAUNCH_CHECK();
  return 0;
}
extern "C" int uc2_kl_bwd(int dtype, int n, int V, const void* pred, int ld, const float* target, const float* lse,
                          const float* gout, void* dpred, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(pred && target && lse && gout && dpred);
  if (dtype == 0) hipLaunchKernelGGL(kl_bwd_kernel<float>, dim3(n), dim3(256), 0, (hipStream_t)stream, V, (const float*)pred, ld, target, lse, gout, (float*)dpred);
  else hipLaunchKernelGGL(kl_bwd_kernel<bf16>, dim3(n), dim3(256), 0, (hipStream_t)stream, V, (const bf16*)pred, ld, target, lse, gout, (bf16*)dpred);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// MSE per element (F.mse_loss(reduction='none'), model/model.py:684-686)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void mse_kernel(size_t n, const T* __restrict__ pred, const float* __restrict__ target,
                           const float* __restrict__ gout, float* __restrict__ loss, T* __restrict__ dpred) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = to_f<T>(pred[i]) - target[i];
    if (loss) loss[i] = d * d;
    if (dpred) dpred[i] = from_f<T>(2.f * d * gout[i]);
  }
}
extern "C" int uc2_mse(int dtype, size_t n, const void* pred, const float* target, const float* gout, float* loss,
                       void* dpred, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n == 0) return 0;
  UC2_CHECK_ARG(pred && target && (loss || (dpred && gout)));
  if (dtype == 0) hipLaunchKernelGGL(mse_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const float*)pred, target, gout, loss, (float*)dpred);
  else hipLaunchKernelGGL(mse_kernel<bf16>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const bf16*)pred, target, gout, loss, (bf16*)dpred);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// triplet ranking loss (model/itm.py:45-53): s = sigmoid(score).view(-1, ss);
//   loss[i, j-1] = max(margin + s[i,j] - s[i,0], 0),  j = 1..ss-1
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void triplet_kernel(int n, int ss, float margin, const T* __restrict__ score,
                               const float* __restrict__ gout, float* __restrict__ loss, T* __restrict__ dscore) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sp = 1.f / (1.f + __expf(-to_f<T>(score[(size_t)i * ss])));
  float dpos = 0.f;
  for (int j = 1; j < ss; ++j) {
    const float sn = 1.f / (1.f + __expf(-to_f<T>(score[(size_t)i * ss + j])));
    const float l = margin + sn - sp;
    if (loss) loss[(size_t)i * (ss - 1) + j - 1] = l > 0.f ? l : 0.f;
    if (dscore) {
      const float g = (l > 0.f) ? gout[(size_t)i * (ss - 1) + j - 1] : 0.f;
      dscore[(size_t)i * ss + j] = from_f<T>(g * sn * (1.f - sn));
      dpos -= g;
    }
  }
  if (dscore) dscore[(size_t)i * ss] = from_f<T>(dpos * sp * (1.f - sp));
}
extern "C" int uc2_triplet(int dtype, int n, int ss, float margin, const void* score, const float* gout, float* loss,
                           void* dscore, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  UC2_CHECK_ARG(ss >= 2);
  if (n <= 0) return 0;
  UC2_CHECK_ARG(score && (loss || (dscore && gout)));
  if (dtype == 0) hipLaunchKernelGGL(triplet_kernel<float>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, ss, margin, (const float*)score, gout, loss, (float*)dscore);
  else hipLaunchKernelGGL(triplet_kernel<bf16>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, ss, margin, (const bf16*)score, gout, loss, (bf16*)dscore);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// dtanh for the pooler backward: dx = dy * (1 - y^2)   (model/layer.py:179-185)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void dtanh_kernel(size_t n, const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = to_f<T>(y[i]);
    dx[i] = from_f<T>(to_f<T>(dy[i]) * (1.f - v * v));
  }
}
extern "C" int uc2_dtanh(int dtype, size_t n, const void* y, const void* dy, void* dx, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n == 0) return 0;
  UC2_CHECK_ARG(y && dy && dx);
  if (dtype == 0) hipLaunchKernelGGL(dtanh_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const float*)y, (const float*)dy, (float*)dx);
  else hipLaunchKernelGGL(dtanh_kernel<bf16>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const bf16*)y, (const bf16*)dy, (bf16*)dx);
  UC2_LAUNCH_CHECK();
  return 0;
}

// gelu as a stand-alone activation (model/layer.py:31-37; the GELU module of the head Sequentials)
template <typename T>
__global__ void gelu_kernel(size_t n, const T* __restrict__ x, T* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = from_f<T>(gelu_t<T>(to_f<T>(x[i])));
}
extern "C" int uc2_gelu(int dtype, size_t n, const void* x, void* y, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n == 0) return 0;
  UC2_CHECK_ARG(x && y);
  if (dtype == 0) hipLaunchKernelGGL(gelu_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const float*)x, (float*)y);
  else hipLaunchKernelGGL(gelu_kernel<bf16>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const bf16*)x, (bf16*)y);
  UC2_LAUNCH_CHECK();
  return 0;
}

// dgelu for head transforms: dx = dy * gelu'(pre)   (model/layer.py:31-37)
template <typename T>
__global__ void dgelu_kernel(size_t n, const T* __restrict__ pre, const T* __restrict__ dy, T* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = from_f<T>(to_f<T>(dy[i]) * dgelu_t<T>(to_f<T>(pre[i])));
}
extern "C" int uc2_dgelu(int dtype, size_t n, const void* pre, const void* dy, void* dx, void* stream) {
  UC2_CHECK_ARG(dtype == 0 || dtype == 1);
  if (n == 0) return 0;
  UC2_CHECK_ARG(pre && dy && dx);
  if (dtype == 0) hipLaunchKernelGGL(dgelu_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const float*)pre, (const float*)dy, (float*)dx);
  else hipLaunchKernelGGL(dgelu_kernel<bf16>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, (const bf16*)pre, (const bf16*)dy, (bf16*)dx);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// casts between fp32 and bf16 (compute copies of the master weights; inputs)
// ---------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void cast_kernel(size_t n, const TI* __restrict__ in, TO* __restrict__ out) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float v[4];
    Vec4<TI>::load(in + i * 4, v);
    Vec4<TO>::store(out + i * 4, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    out[i] = from_f<TO>(to_f<TI>(in[i]));
  }
}
extern "C" int uc2_cast(int from_dtype, int to_dtype, size_t n, const void* in, void* out, void* stream) {
  UC2_CHECK_ARG((from_dtype == 0 || from_dtype == 1) && (to_dtype == 0 || to_dtype == 1));
  if (n == 0) return 0;
  UC2_CHECK_ARG(in && out);
  UC2_CHECK_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
  dim3 grid(ew_grid(n, 4)), block(EW_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  if (from_dtype == 0 && to_dtype == 1) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, st, n, (const float*)in, (bf16*)out);
  else if (from_dtype == 1 && to_dtype == 0) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, st, n, (const bf16*)in, (float*)out);
  else if (from_dtype == 0) hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, n, (const float*)in, (float*)out);
  else hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, block, 0, st, n, (const bf16*)in, (bf16*)out);
  UC2_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Batched bf16 transposes: dst[off_i .. ] viewed as [cols_i][rows_i] = transpose of src[off_i ..] viewed as [rows_i][cols_i], for
// up to UC2_TR_MAX matrices in ONE launch.  Keeps k-contiguous copies W^T of the layer weights beside the bf16 shadow arena, so
// that the input-gradient GEMMs dX = dY W read both operands k-contiguously (the ping-pong kernel on the 16x16x32 MFMA is 4-10 %
// faster on that layout than with W read through the transposing LDS read); refreshed once per optimizer step: 170 MB each way.
// ---------------------------------------------------------------------------------------
#define UC2_TR_MAX 64
struct TrBatch { int n; int tile0[UC2_TR_MAX + 1]; int rows[UC2_TR_MAX]; int cols[UC2_TR_MAX]; unsigned long long off[UC2_TR_MAX]; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(TrBatch b, const bf16* __restrict__ src, bf16* __restrict__ dst) {
  __shared__ bf16 tile[64][64 + 4];
  int i = 0;
  for (int k = 1; k < UC2_TR_MAX; ++k) if (k < b.n && (int)blockIdx.x >= b.tile0[k]) i = k;     // (uniform)
  const int t = blockIdx.x - b.tile0[i];
  const int tc = b.cols[i] / 64, tr_ = t / tc, tcx = t - tr_ * tc;
  const bf16* s = src + b.off[i] + (size_t)(tr_ * 64) * b.cols[i] + tcx * 64;
  bf16* d = dst + b.off[i] + (size_t)(tcx * 64) * b.rows[i] + tr_ * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 x 4 elements each
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ty + 16 * r;
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(s + (size_t)row * b.cols[i] + tx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row][tx * 4 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = ty + 16 * r;                                     // source column = destination row
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tile[tx * 4 + e][col];
    *reinterpret_cast<bf16x4*>(d + (size_t)col * b.rows[i] + tx * 4) = v;
  }
}
// Head-interleaved copies of fused QKV projections: for every item a bf16 weight block [3 nh D, cols] (rows q | k | v, head h at
// row h D of each third) and its fp32 bias [3 nh D] are copied with row w nh D + h D + d -> row h 3D + w D + d.  A GEMM on the copy
// writes a head's q | k | v as ONE 384-byte segment per token (D = 64) instead of three 128-byte segments 1536 bytes apart -- the
// attention kernels' only access pattern (4.2 -> 4.7 TB/s forward).  One launch for all layers, once per optimizer step.
#define UC2_ILV_MAX 64
struct IlvBatch { int n, nh, D, cols; unsigned long long w_src[UC2_ILV_MAX], w_dst[UC2_ILV_MAX], b_src[UC2_ILV_MAX], b_dst[UC2_ILV_MAX]; };
__global__ __launch_bounds__(256) void qkv_interleave_kernel(IlvBatch b, const bf16* __restrict__ wsrc, bf16* __restrict__ wdst,
                                                             const float* __restrict__ bsrc, float* __restrict__ bdst) {
  const int item = blockIdx.y, rows = 3 * b.nh * b.D, cpr = b.cols / 8;       // 16-byte chunks per row
  const bf16* s = wsrc + b.w_src[item];
  bf16* d = wdst + b.w_dst[item];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cpr; i += gridDim.x * blockDim.x) {
    const int r = i / cpr, c = i - r * cpr;                                  // destination row r = h 3D + w D + dd
    const int h = r / (3 * b.D), rr = r - h * 3 * b.D, w = rr / b.D, dd = rr - w * b.D;
    const int sr = w * b.nh * b.D + h * b.D + dd;
    *reinterpret_cast<bf16x8*>(d + (size_t)r * b.cols + c * 8) = *reinterpret_cast<const bf16x8*>(s + (size_t)sr * b.cols + c * 8);
    if (c == 0 && bsrc) bdst[b.b_dst[item] + r] = bsrc[b.b_src[item] + sr];
  }
}
struct Uc2IlvItem { size_t w_src, w_dst, b_src, b_dst; };            // mirrors include/uc2_hip.h (element offsets)
extern "C" int uc2_qkv_interleave_batch(int n, const Uc2IlvItem* items, int nh, int D, int cols, const void* w_src_base,
                                        void* w_dst_base, const float* b_src_base, float* b_dst_base, void* stream) {
  UC2_CHECK_ARG(n >= 0 && (n == 0 || items));
  if (n == 0) return 0;
  UC2_CHECK_ARG(nh > 0 && D > 0 && cols > 0 && (cols % 8) == 0 && w_src_base && w_dst_base && w_src_base != w_dst_base);
  UC2_CHECK_ARG((b_src_base == nullptr) == (b_dst_base == nullptr));
  for (int i0 = 0; i0 < n; i0 += UC2_ILV_MAX) {
    IlvBatch b{};
    b.n = n - i0 < UC2_ILV_MAX ? n - i0 : UC2_ILV_MAX; b.nh = nh; b.D = D; b.cols = cols;
    for (int k = 0; k < b.n; ++k) {
      const Uc2IlvItem& it = items[i0 + k];
      UC2_CHECK_ARG((it.w_src % 8) == 0 && (it.w_dst % 8) == 0);
      b.w_src[k] = it.w_src; b.w_dst[k] = it.w_dst; b.b_src[k] = it.b_src; b.b_dst[k] = it.b_dst;
    }
    const int work = 3 * nh * D * (cols / 8);
    hipLaunchKernelGGL(qkv_interleave_kernel, dim3((work + 255) / 256 < 512 ? (work + 255) / 256 : 512, b.n), dim3(256), 0,
                       (hipStream_t)stream, b, (const bf16*)w_src_base, (bf16*)w_dst_base, b_src_base, b_dst_base);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}

struct Uc2TransposeItem { size_t offset; int rows, cols; };        // mirrors include/uc2_hip.h
extern "C" int uc2_transpose_batch(int n, const Uc2TransposeItem* items, const void* src_base, void* dst_base, void* stream) {
  UC2_CHECK_ARG(n >= 0 && (n == 0 || items));
  if (n == 0) return 0;
  UC2_CHECK_ARG(src_base && dst_base && src_base != dst_base);
  for (int i0 = 0; i0 < n; i0 += UC2_TR_MAX) {
    TrBatch b{};
    b.n = n - i0 < UC2_TR_MAX ? n - i0 : UC2_TR_MAX;
    int tiles = 0;
    for (int k = 0; k < b.n; ++k) {
      const Uc2TransposeItem& it = items[i0 + k];
      UC2_CHECK_ARG(it.rows > 0 && it.cols > 0 && (it.rows % 64) == 0 && (it.cols % 64) == 0 && (it.offset % 4) == 0);
      b.tile0[k] = tiles; b.rows[k] = it.rows; b.cols[k] = it.cols; b.off[k] = it.offset;
      tiles += (it.rows / 64) * (it.cols / 64);
    }
    b.tile0[b.n] = tiles;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, b, (const bf16*)src_base, (bf16*)dst_base);
  }
  UC2_LAUNCH_CHECK();
  return 0;
}
