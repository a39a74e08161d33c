// Microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 in loops shaped like k_gemm's inner loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float *out, int iters, int rnd) {
  __shared__ float lds[2][288 * 17];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  for (int i = t; i < 2 * 288 * 17; i += 256) (&lds[0][0])[i] = rnd ? (float)((i * 2654435761u) >> 8) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[5];
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = (float)l * 0.01f, b[5] = {1.f, 2.f, 3.f, 4.f, 5.f};
  for (int it = 0; it < iters; ++it) {
    const float *as = &lds[it & 1][(w * 32 + (l & 31)) * 17 + (l >> 5)];
    const float *bs = &lds[it & 1][(128 + (l & 31)) * 17 + (l >> 5)];
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      if (VARIANT >= 1) {
        a = as[kk];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) b[nt] = bs[nt * 32 * 17 + kk];
      }
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[nt], acc[nt], 0, 0, 0);
    }
    if (VARIANT >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}

template <int V>
void run(const char *name, int blocks, size_t dyn, int rnd = 0) {
  float *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  if (dyn) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), dyn, 0, out, iters, rnd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 40 * 4096.0;
    if (rep == 2) printf("%-28s blocks %5d dynLDS %6zu: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, dyn, ms, flop / ms / 1e9);
  }
  (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 768}) {
    run<0>("pure mfma", blocks, 0);
    run<1>("mfma + lds reads", blocks, 0);
    run<2>("mfma + lds reads + barrier", blocks, 0);
  }
  run<1>("random data: mfma + lds", 512, 0, 1);
  run<2>("random data: + barrier", 512, 0, 1);
  run<2>("random data: + barrier", 768, 0, 1);
  run<2>("1 block/CU (dyn lds)", 256, 90000);
  run<2>("1 block/CU x2 rounds", 512, 90000);
  return 0;
}
