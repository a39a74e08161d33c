// Where does the dispatcher put the workgroups of a k_step-shaped launch (4096 workgroups of one wavefront, 8.3 KB of LDS
// each: 16 per CU, all resident at once)?  Every workgroup records its XCC / SE / CU / SIMD ids.  tools/micro/wg_map.py
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(64) void k_map(uint32_t *out, int spin) {
  extern __shared__ uint8_t lds[];
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (uint8_t)hw;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);  // stay resident (100 MHz ticks)
  if (threadIdx.x == 0) out[blockIdx.x * 4 + 0] = hw, out[blockIdx.x * 4 + 1] = xcc, out[blockIdx.x * 4 + 2] = (uint32_t)t0, out[blockIdx.x * 4 + 3] = lds[1];
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, lds = argc > 2 ? atoi(argv[2]) : 8512;
  uint32_t *d, *h = (uint32_t *)malloc((size_t)n * 16);
  hipMalloc(&d, (size_t)n * 16);
  hipLaunchKernelGGL(k_map, dim3(n), dim3(64), lds, 0, d, 5000 /* 50 us */);
  hipDeviceSynchronize();
  hipMemcpy(h, d, (size_t)n * 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%d %u %u %u\n", i, h[i * 4], h[i * 4 + 1], h[i * 4 + 2]);
  return 0;
}
