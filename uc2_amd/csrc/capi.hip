// Library-level entry points of libuc2_hip.so: ABI version and the last-error string.
// Error convention for every uc2_* call: 0 = ok, < 0 = argument error, > 0 = hipError_t.
// Nothing throws across the boundary; the Python host raises (uc2_amd/_lib.py).
#include "common.h"
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void uc2_set_error(const char* file, int line, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, what);
}
extern "C" const char* uc2_last_error(void) { return g_err; }
extern "C" int uc2_abi_version(void) { return 13; }

// number of compute units / arch string of the current device (diagnostics for bench.py)
extern "C" int uc2_device_info(int* cu_count, int* clock_khz, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  return 0;
}
