// Device-side batch assembly: what the reference's collate functions do on the host with Python loops
// (data/itm.py:205-232 xlmr_itm_collate, data/mrm.py:73-119 xlmr_mrfr_collate, data/mlm.py:761-801 xlmr_mlm_collate;
// helpers data/data.py:360-384 pad_tensors / get_gather_index, data/mrm.py:36-39 _mask_img_feat) and what
// PrefetchLoader (data/loader.py:85-140) then copies to the GPU tensor by tensor.  Here the host only concatenates the
// ragged per-sample arrays into flat pinned buffers; after ONE async copy per buffer two kernels build the padded batch:
//   uc2_collate_regions : [sum nb, D] rows -> [B, maxR, D] zero-padded, masked regions zero-filled, optional bf16 output
//   uc2_collate_index   : input_ids (pad id), attention mask, gather index, padded region masks, img_mask_tgt, txt_labels
#include "common.h"

template <typename TO>
__global__ __launch_bounds__(256) void collate_regions_kernel(int maxR, int D, const float* __restrict__ flat,
                                                              const int64_t* __restrict__ row_off,
                                                              const uint8_t* __restrict__ mask_flat, TO* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);          // row of the padded output of batch element b
  const int b = blockIdx.y;
  if (r >= maxR) return;
  const int64_t o0 = row_off[b], nb = row_off[b + 1] - o0;
  const bool live = r < nb && !(mask_flat && mask_flat[o0 + r]);
  const float* src = flat + (size_t)(o0 + r) * D;
  TO* dst = out + ((size_t)b * maxR + r) * D;
  if ((D & 3) == 0) {
    for (int c = lane * 4; c < D; c += 256) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (live) Vec4<float>::load(src + c, v);
      Vec4<TO>::store(dst + c, v);
    }
  } else {
    for (int c = lane; c < D; c += 64) dst[c] = from_f<TO>(live ? src[c] : 0.f);
  }
}

__global__ __launch_bounds__(256) void collate_index_kernel(int maxT, int maxR, int Lout, const int64_t* __restrict__ ids_flat,
                                                            const int64_t* __restrict__ txt_off, const int64_t* __restrict__ row_off,
                                                            int64_t pad_id, const uint8_t* __restrict__ mask_flat,
                                                            const int64_t* __restrict__ labels_flat,
                                                            int64_t* __restrict__ input_ids, int64_t* __restrict__ attn,
                                                            int64_t* __restrict__ gather, uint8_t* __restrict__ img_mask,
                                                            uint8_t* __restrict__ img_mask_tgt, int64_t* __restrict__ txt_labels) {
  const int b = blockIdx.y;
  const int64_t t0 = txt_off[b], tl = txt_off[b + 1] - t0, r0 = row_off[b], nb = row_off[b + 1] - r0;
  const int n = max(max(maxT, maxR), Lout);
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    if (j < maxT) {
      input_ids[(size_t)b * maxT + j] = j < tl ? ids_flat[t0 + j] : pad_id;               // pad_sequence(padding_value = 1)
      if (txt_labels) txt_labels[(size_t)b * maxT + j] = (j < tl && labels_flat) ? labels_flat[t0 + j] : -1;
    }
    if (j < maxR && img_mask) img_mask[(size_t)b * maxR + j] = (j < nb && mask_flat) ? mask_flat[r0 + j] : 0;
    if (j < Lout) {
      attn[(size_t)b * Lout + j] = j < tl + nb ? 1 : 0;
      // get_gather_index (data/data.py:376-384): arange, with [tl, tl+nb) pointing at the regions behind the padded text
      gather[(size_t)b * Lout + j] = (j >= tl && j < tl + nb) ? (int64_t)maxT + (j - tl) : (int64_t)j;
      if (img_mask_tgt) img_mask_tgt[(size_t)b * Lout + j] = (j >= tl && j < tl + nb && mask_flat) ? mask_flat[r0 + (j - tl)] : 0;
    }
  }
}

extern "C" int uc2_collate_regions(int out_dtype, int B, int maxR, int D, const float* flat, const int64_t* row_off,
                                   const uint8_t* mask_flat, void* out, void* stream) {
  UC2_CHECK_ARG(out_dtype == 0 || out_dtype == 1);
  if (B <= 0 || maxR <= 0) return 0;
  UC2_CHECK_ARG(D > 0 && flat && row_off && out);
  dim3 grid((maxR + 3) / 4, B);
  if (out_dtype == 0) hipLaunchKernelGGL(collate_regions_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, maxR, D, flat, row_off, mask_flat, (float*)out);
  else hipLaunchKernelGGL(collate_regions_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, maxR, D, flat, row_off, mask_flat, (bf16*)out);
  UC2_LAUNCH_CHECK();
  return 0;
}

extern "C" int uc2_collate_index(int B, int maxT, int maxR, int Lout, const int64_t* ids_flat, const int64_t* txt_off,
                                 const int64_t* row_off, int64_t pad_id, const uint8_t* mask_flat, const int64_t* labels_flat,
                                 int64_t* input_ids, int64_t* attn_masks, int64_t* gather_index, uint8_t* img_masks,
                                 uint8_t* img_mask_tgt, int64_t* txt_labels, void* stream) {
  if (B <= 0) return 0;
  UC2_CHECK_ARG(maxT > 0 && Lout > 0 && ids_flat && txt_off && row_off && input_ids && attn_masks && gather_index);
  const int n = maxT > Lout ? maxT : Lout;
  dim3 grid((n + 255) / 256, B);
  hipLaunchKernelGGL(collate_index_kernel, grid, dim3(256), 0, (hipStream_t)stream, maxT, maxR, Lout, ids_flat, txt_off, row_off,
                     pad_id, mask_flat, labels_flat, input_ids, attn_masks, gather_index, img_masks, img_mask_tgt, txt_labels);
  UC2_LAUNCH_CHECK();
  return 0;
}
