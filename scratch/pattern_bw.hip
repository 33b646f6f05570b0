// What HBM delivers for the ATTENTION access pattern with no arithmetic at all: one workgroup of 192 threads per (batch, head)
// loads the head's q, k, v tiles (96 rows x 128 B each, 16-byte loads, the kernels' thread -> chunk map) and stores a 96 x 128 B
// ctx tile that depends on them.  Layouts: 0 = [B L][3][nh][64] (the projection's natural order: 128-B pieces 1536 B apart, row
// stride 4608 B), 1 = [B L][nh][3][64] (head-interleaved: one 384-B piece per token and head), 2 = head-major compact
// ([B][nh][3][L][64]: 36 KB per head), 3 = plain streaming copy of the same bytes.   hipcc --offload-arch=gfx950 -O3 pattern_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(192) void pattern_kernel(const u32x4* __restrict__ qkv, u32x4* __restrict__ ctx, int L, int nh, int layout) {
  const int bh = blockIdx.x, b = bh / nh, h = bh % nh, t = threadIdx.x;
  // offsets in 16-byte units
  size_t row_stride, base[3], out_base = ((size_t)b * L) * nh * 8 + (size_t)h * 8, out_stride = (size_t)nh * 8;
  if (layout == 0) { row_stride = 3 * nh * 8; for (int w = 0; w < 3; ++w) base[w] = (size_t)b * L * row_stride + ((size_t)w * nh + h) * 8; }
  else if (layout == 1) { row_stride = 3 * nh * 8; for (int w = 0; w < 3; ++w) base[w] = (size_t)b * L * row_stride + ((size_t)h * 3 + w) * 8; }
  else { row_stride = 8; for (int w = 0; w < 3; ++w) base[w] = (((size_t)b * nh + h) * 3 + w) * L * 8; out_base = ((size_t)b * nh + h) * L * 8; out_stride = 8; }
  u32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + i * 192, row = c >> 3, c8 = c & 7;
    u32x4 v = qkv[base[0] + row * row_stride + c8];
    const u32x4 k = qkv[base[1] + row * row_stride + c8], w = qkv[base[2] + row * row_stride + c8];
    v.x ^= k.x ^ w.x; v.y ^= k.y ^ w.y; v.z ^= k.z ^ w.z; v.w ^= k.w ^ w.w;
    acc[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + i * 192, row = c >> 3, c8 = c & 7;
    ctx[out_base + row * out_stride + c8] = acc[i];
  }
}
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n3, size_t n1) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = src[i]; const u32x4 k = src[n1 + i], w = src[2 * n1 + i];
    v.x ^= k.x ^ w.x; v.y ^= k.y ^ w.y; v.z ^= k.z ^ w.z; v.w ^= k.w ^ w.w;
    dst[i] = v;
  }
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, L = 96, nh = 12;
  const size_t n1 = (size_t)B * L * nh * 8;              // 16-byte units of one of q / k / v / ctx
  u32x4 *qkv, *ctx;
  hipMalloc(&qkv, 3 * n1 * 16); hipMalloc(&ctx, n1 * 16);
  hipMemset(qkv, 1, 3 * n1 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"natural [BL][3][nh][64] (128-B pieces)", "head-interleaved [BL][nh][3][64] (384-B pieces)", "head-major compact (36 KB per head)", "streaming copy"};
  for (int rep = 0; rep < 2; ++rep)
    for (int layout = 0; layout < 4; ++layout) {
      for (int it = 0; it < 23; ++it) {
        if (it == 3) hipEventRecord(e0);
        if (layout < 3) hipLaunchKernelGGL(pattern_kernel, dim3(B * nh), dim3(192), 0, 0, qkv, ctx, L, nh, layout);
        else hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, qkv, ctx, 3 * n1, n1);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("%-52s %7.1f us  %.2f TB/s\n", names[layout], ms / 20 * 1e3, 4.0 * n1 * 16 / (ms / 20 * 1e-3) / 1e12);
    }
  return 0;
}
