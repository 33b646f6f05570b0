// experiment: does gfx950 execute scalar memory atomics (s_atomic_add with return, lgkmcnt)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* q, int* out) {
  int v = 1;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(q) : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  int *q, *out; const int n = 4096;
  hipMalloc(&q, 4); hipMalloc(&out, n * 4); hipMemset(q, 0, 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, q, out);
  hipError_t e = hipDeviceSynchronize();
  int hq = -1; static int ho[4096];
  hipMemcpy(&hq, q, 4, hipMemcpyDeviceToHost); hipMemcpy(ho, out, n * 4, hipMemcpyDeviceToHost);
  static char seen[4096]; int dup = 0, oor = 0;
  for (int i = 0; i < n; ++i) { if (ho[i] < 0 || ho[i] >= n) ++oor; else if (seen[ho[i]]++) ++dup; }
  printf("sync %s; counter %d (want %d); returned values: %d duplicates, %d out of range\n", hipGetErrorString(e), hq, n, dup, oor);
  return 0;
}
