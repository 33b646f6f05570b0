// MFMA attention kernels (bf16).  Placeholder until the MFMA kernels land: reports "unsupported"
// so the auto dispatch in attention.hip uses the fp32-math kernels.
#include "common.h"
extern "C" int uc2_attn_mfma_supported(int L, int D) { (void)L; (void)D; return 0; }
extern "C" int uc2_attn_fwd_mfma(int, int, int, int, const void*, const float*, float, float, const uint64_t*, uint64_t,
                                 void*, float*, void*) {
  uc2_set_error(__FILE__, __LINE__, "MFMA attention not built");
  return -1;
}
extern "C" int uc2_attn_bwd_mfma(int, int, int, int, const void*, const float*, float, float, const uint64_t*, uint64_t,
                                 const void*, const void*, const float*, void*, void*) {
  uc2_set_error(__FILE__, __LINE__, "MFMA attention not built");
  return -1;
}
