// What writing the dense observation costs as a pure write, by pattern (tools/experiments/README.md):
//   rows64   one 64-lane wave per agent streams its own 123 008-byte row (k_observe_dense's pattern)
//   rows256  one 256-thread workgroup per agent streams its row (k_observe's pattern)
//   front    the same bytes written as one moving front over the whole buffer (what a fill does)
// hipcc --offload-arch=gfx950 -O3 -o row_write row_write.hip && ./row_write
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROW4 = 30752 / 4, AGENTS = 4096;

template <bool NT>
__device__ inline void st(f32x4 *p) {
  if (NT)
    __builtin_nontemporal_store((f32x4)(0.f), p);
  else
    *p = (f32x4)(0.f);
}
template <bool NT>
__global__ __launch_bounds__(64) void rows64(f32x4 *out) {
  f32x4 *o = out + (size_t)blockIdx.x * ROW4;
  for (int i = threadIdx.x; i < ROW4; i += 64) st<NT>(o + i);
}
template <bool NT>
__global__ __launch_bounds__(256) void rows256(f32x4 *out) {
  f32x4 *o = out + (size_t)blockIdx.x * ROW4;
  for (int i = threadIdx.x; i < ROW4; i += 256) st<NT>(o + i);
}
template <bool NT>
__global__ __launch_bounds__(256) void front(f32x4 *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<NT>(out + i);
}

int main() {
  f32x4 *d;
  const size_t n = (size_t)AGENTS * ROW4;
  if (hipMalloc(&d, n * sizeof(f32x4)) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s %.1f us = %.2f TB/s\n", name, ms / 20 * 1e3, n * 16.0 / (ms / 20) / 1e9);
  };
  time("rows64, non-temporal", [&] { hipLaunchKernelGGL(rows64<true>, dim3(AGENTS), dim3(64), 0, 0, d); });
  time("rows64, plain", [&] { hipLaunchKernelGGL(rows64<false>, dim3(AGENTS), dim3(64), 0, 0, d); });
  time("rows256, non-temporal", [&] { hipLaunchKernelGGL(rows256<true>, dim3(AGENTS), dim3(256), 0, 0, d); });
  time("rows256, plain", [&] { hipLaunchKernelGGL(rows256<false>, dim3(AGENTS), dim3(256), 0, 0, d); });
  time("front (2048 wg), non-temp.", [&] { hipLaunchKernelGGL(front<true>, dim3(2048), dim3(256), 0, 0, d, n); });
  time("front (2048 wg), plain", [&] { hipLaunchKernelGGL(front<false>, dim3(2048), dim3(256), 0, 0, d, n); });
  return 0;
}
