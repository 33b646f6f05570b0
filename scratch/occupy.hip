// experiment helper (not part of the library): n workgroups that each hold `lds_bytes` of LDS and spin for `cycles`
// shader clocks -- emulates a concurrent communication kernel occupying compute units
#include <hip/hip_runtime.h>
extern "C" __global__ void occupy_kernel(long long cycles, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < cycles) { acc += lds[(threadIdx.x + acc) & 63]; __builtin_amdgcn_s_sleep(20); }
  if (acc == 0x7fffffff) *sink = acc;
}
extern "C" int occupy(int n_wg, int lds_bytes, long long cycles, int* sink, void* stream) {
  hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(256), lds_bytes, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
