// r03 experiment (tools/experiments/README.md): the generator of random.hpp:54-62 for FOUR arenas per wavefront, one
// arena per 16-lane DPP row, as a standalone producer kernel.  Not part of the product.  Used (a) to check the row form
// against the generator's known answers and (b) as the producer side of a timing build: how fast would k_step be if its
// draws came from producer waves running beside it?
//
// Row layout: `lo` lane j = log3(random[j]) for j = 0..15; `hi` lane 14 = log3(random[16]), lane 15 = log3(random[17]),
// lane 13 = a copy of the newest log with seed 1 / us 0: its power is the value of the draw made in the previous round.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOGT_OFF 1024
#define BIAS_LANE (61441u + 12u * 65537u)  // 16 * BIAS_LANE = 207 * 65537 + 1: the generator's `sum = 1`, every row sum positive
static_assert((16ull * BIAS_LANE) % 65537ull == 1ull, "bias");
static_assert(16ull * BIAS_LANE > 18ull * 10ull * 65536ull, "positive row sums");

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ int row_sum(int x) {  // butterfly inside each 16-lane row: every lane ends with the row's total
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);  // row_mirror
  return x;
}
__device__ __forceinline__ int pow3_signed(const uint32_t *xt, uint32_t m) {  // 3^m (mod 65537) as lo16 - hi16 of the table product
  const uint32_t pr = mul24(xt[m & 255u], xt[256u + ((m >> 8) & 255u)]);
  return (int)(pr & 0xffffu) - (int)(pr >> 16);
}

struct RowsArgs {
  const uint16_t *logt;     // [LOGT_OFF + 65537]
  const uint32_t *exptab;   // [512]
  const uint32_t *state;    // [arenas][18] log3(random[i]) | us << 20 | seed << 24 after the warm-up
  uint32_t jomle0;          // 18 + 1024
  uint16_t *out;            // [arenas][rounds] draw values (mode 0) or nullptr
  int arenas, rounds;
};

// one wavefront per workgroup, 4 arenas per wavefront
__global__ __launch_bounds__(64) void k_rng_rows(RowsArgs a) {
  __shared__ uint32_t xt[512];
  __shared__ uint32_t ring[4][128];
  const int l = threadIdx.x, row = l >> 4, j = l & 15;
  for (int i = l; i < 512; i += 64) xt[i] = a.exptab[i];
  __syncthreads();
  const int arena = (int)blockIdx.x * 4 + row;
  const bool live = arena < a.arenas;
  const uint32_t *st = a.state + (size_t)(live ? arena : 0) * 18;
  const uint32_t w_lo = st[j], w_hi = j >= 14 ? st[j + 2] : 0u;
  uint32_t lo = w_lo & 0xffffu, hi = j >= 14 ? (w_hi & 0xffffu) : 0u;
  const uint32_t slo = (w_lo >> 24) & 15u, ulo = (w_lo >> 20) & 15u;
  const uint32_t shi = j >= 14 ? ((w_hi >> 24) & 15u) : (j == 13 ? 1u : 0u), uhi = j >= 14 ? ((w_hi >> 20) & 15u) : 0u;
  uint32_t e = a.jomle0;
  const uint16_t *lt = a.logt + LOGT_OFF;
  uint32_t la = 0;
  for (int k = 0; k <= a.rounds; ++k) {
    if (k) {  // commit the draw whose log arrived: rotate, newest last (+ the output copy on lane 13)
      e += 1u;
      const uint32_t lnew = mul24(la, e & 0xffffu) & 0xffffu;
      const uint32_t lo_s = dpp<0x101>(lo);  // row_shl:1  lane j <- lane j + 1
      const uint32_t hi_r = dpp<0x111>(hi);  // row_shr:1  lane 15 <- hi[14]
      const uint32_t hi_s = dpp<0x101>(hi);  // lane 14 <- hi[15]
      lo = j == 15 ? hi_r : lo_s;
      hi = (j == 15 || j == 13) ? lnew : (j == 14 ? hi_s : 0u);
    }
    const int d_lo = pow3_signed(xt, mul24(lo, slo));
    const int d_hi = pow3_signed(xt, mul24(hi, shi));
    if (k && j == 13) {  // the value of the draw committed in this round's first half
      const uint32_t v = (uint32_t)(d_hi + ((d_hi >> 31) & 65537));
      ring[row][(k - 1) & 127] = (hi << 10) | (v & 1023u);
      if (a.out && live) a.out[(size_t)arena * a.rounds + (k - 1)] = (uint16_t)(v & 1023u);
    }
    const int x = row_sum(d_lo * (int)ulo + d_hi * (int)uhi + (int)BIAS_LANE);
    const int t = (int)((uint32_t)x & 0xffffu) - (int)((uint32_t)x >> 16);
    la = lt[t];
  }
  if (a.out == nullptr && l == 0 && ring[0][5] == 0xdeadbeefu) a.out[0] = 1;  // keep the ring alive
}

extern "C" int rr_run(const uint16_t *logt, const uint32_t *exptab, const uint32_t *state, uint32_t jomle0, uint16_t *out,
                      int arenas, int rounds, void *stream) {
  RowsArgs a{logt, exptab, state, jomle0, out, arenas, rounds};
  hipLaunchKernelGGL(k_rng_rows, dim3((unsigned)((arenas + 3) / 4)), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
