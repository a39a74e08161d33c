// Microbenchmark 2: the K-loop of k_gemm (groups of two products, fragments one group ahead, sched_barrier) without
// any global memory traffic, to separate schedule effects from memory effects.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BKT, int FENCE, int LSTORE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters) {
  constexpr int LD = BKT + 1, BM = 128, BN = 160, NT = 5, GS = BKT / 4;
  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BN * LD];
  const int t = threadIdx.x, l = t & 63, wm = t >> 6;
  for (int i = t; i < 2 * BM * LD; i += 256) (&As[0][0])[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 8388608.0f) - 1.0f;
  for (int i = t; i < 2 * BN * LD; i += 256) (&Bs[0][0])[i] = (float)((i * 40503u) & 1023) * (1.0f / 512.0f) - 1.0f;
  __syncthreads();
  f32x16 acc[NT];
  for (int i = 0; i < NT; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float fa[2][2], fb[2][NT][2];
  auto fload = [&](int buf, int grp, int slot) {
    const float *as = &As[buf][(wm * 32 + (l & 31)) * LD + (l >> 5)];
    const float *bs = &Bs[buf][((l & 31)) * LD + (l >> 5)];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      fa[slot][h] = as[4 * grp + 2 * h];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[slot][nt][h] = bs[nt * 32 * LD + 4 * grp + 2 * h];
    }
  };
  float keep[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  fload(0, 0, 0);
  for (int i = 0; i < iters; ++i) {
    const int buf = i & 1;
#pragma unroll
    for (int grp = 0; grp < GS; ++grp) {
      if (LSTORE && grp == GS / 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) As[buf ^ 1][((t / 8) + (j & 3) * 32) * LD + (t & 7) * 4 + (j >> 2)] = keep[j] + (float)i;
      }
      if (grp + 1 < GS) {
        fload(buf, grp + 1, (grp + 1) & 1);
      } else {
        __syncthreads();
        fload(buf ^ 1, 0, 0);
      }
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[grp & 1][h], fb[grp & 1][nt][h], acc[nt], 0, 0, 0);
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NT; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}

template <int BKT, int FENCE, int LSTORE>
void run(const char *name, int blocks) {
  float *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 32000 / BKT;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BKT, FENCE, LSTORE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * (BKT / 2) * 5 * 4096.0;
    if (rep == 2) printf("%-40s BK %2d blocks %4d: %8.3f ms  %7.1f TFLOP/s\n", name, BKT, blocks, ms, flop / ms / 1e9);
  }
  (void)hipFree(out);
}

int main() {
  run<16, 1, 0>("loop, fenced", 512);
  run<16, 0, 0>("loop, compiler-scheduled", 512);
  run<32, 1, 0>("loop, fenced", 512);
  run<32, 0, 0>("loop, compiler-scheduled", 512);
  run<32, 1, 1>("loop, fenced, + lds stores", 512);
  run<32, 0, 1>("loop, compiler-scheduled, + lds stores", 512);
  run<16, 1, 1>("loop, fenced, + lds stores", 512);
  return 0;
}
