// tools/micro/ramp.hip — what does STARTING a wavefront cost?  One-wave workgroups that do (almost) nothing, timed with
// HIP events over launches of 1024 .. 16384 workgroups, varied in what a start has to set up: nothing, dynamic LDS, a
// large register allocation, a large kernel-argument block.  (round 4: the one-launch-per-step loops are bound by this.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ramp.hip -o tools/micro/ramp && tools/micro/ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big {
  int v[64];
};
__global__ __launch_bounds__(64) void k_empty(unsigned *out) {
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1u;
}
__global__ __launch_bounds__(64) void k_lds(unsigned *out) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = blockIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && out) out[blockIdx.x] = lds[63];
}
__global__ __launch_bounds__(64) void k_regs(unsigned *out) {
  unsigned x;
  asm volatile("v_mov_b32 v120, 1\n\tv_mov_b32 %0, v120" : "=v"(x) : : "v120");  // 121+ registers allocated per lane
  if (threadIdx.x == 0 && out) out[blockIdx.x] = x;
}
__global__ __launch_bounds__(64) void k_args(Big b, unsigned *out) {
  if (threadIdx.x == 0 && out) out[blockIdx.x] = (unsigned)b.v[blockIdx.x & 63];
}
__global__ __launch_bounds__(256) void k_empty256(unsigned *out) {
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1u;
}
// a wave that lives `spin` ticks of the 100 MHz clock, holding 8 KB of LDS: how the starts and the waves' lives add up
// when all 4096 waves are resident at once (16 per CU)
__global__ __launch_bounds__(64) void k_spin(unsigned *out, int spin) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1u;
}

// the same with a register allocation like k_step's (121 registers per lane: four such waves fill a SIMD's register file)
__global__ __launch_bounds__(64) void k_spin_regs(unsigned *out, int spin) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = blockIdx.x;
  unsigned x;
  asm volatile("v_mov_b32 v120, 1\n\tv_mov_b32 %0, v120" : "=v"(x) : : "v120");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0 && out) out[blockIdx.x] = x;
}

template <class F>
static float timed(F launch, int reps = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  unsigned *out;
  hipMalloc(&out, 65536 * 4);
  hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  Big big = {};
  printf("%8s %10s %10s %10s %10s %10s %10s %12s %12s %12s\n", "waves", "empty", "lds 8K", "lds 32K", "regs 121", "args 256B", "256thr/4", "spin 20us", "spin 40us", "spin20 r121");
  for (int n : {1024, 2048, 4096, 8192, 16384}) {
    const float e = timed([&] { hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, out); });
    const float l8 = timed([&] { hipLaunchKernelGGL(k_lds, dim3(n), dim3(64), 8 * 1024, 0, out); });
    const float l32 = timed([&] { hipLaunchKernelGGL(k_lds, dim3(n), dim3(64), 32 * 1024, 0, out); });
    const float r = timed([&] { hipLaunchKernelGGL(k_regs, dim3(n), dim3(64), 0, 0, out); });
    const float a = timed([&] { hipLaunchKernelGGL(k_args, dim3(n), dim3(64), 0, 0, big, out); });
    const float q = timed([&] { hipLaunchKernelGGL(k_empty256, dim3(n / 4), dim3(256), 0, 0, out); });
    const float s20 = timed([&] { hipLaunchKernelGGL(k_spin, dim3(n), dim3(64), 8 * 1024, 0, out, 2000); });
    const float s40 = timed([&] { hipLaunchKernelGGL(k_spin, dim3(n), dim3(64), 8 * 1024, 0, out, 4000); });
    const float s20r = timed([&] { hipLaunchKernelGGL(k_spin_regs, dim3(n), dim3(64), 8 * 1024, 0, out, 2000); });
    printf("%8d %9.1fus %9.1fus %9.1fus %9.1fus %9.1fus %9.1fus %11.1fus %11.1fus %11.1fus\n", n, e, l8, l32, r, a, q, s20, s40, s20r);
  }
  return 0;
}
