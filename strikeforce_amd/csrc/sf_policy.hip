// sf_policy.hip — batched on-device evaluation of the reference's bot network.
//
// What is computed, per agent, is AgentModel::forward of StrikeForce-client/bots/bot-0.5/Modules.hpp:54-179:
//   GameCNN (4x Conv2d 3x3 stride 2, no bias, no activation; 32->160 channels; 31->15->7->3->1)   :54-72
//   Backbone: L1-style normalisations x*160/(sum|x|+1e-8), gru0, "pov" (5 centre cells x 32 channels + last
//   action one-hot), Linear(329->160), gru1 with a residual                                        :106-134
//   heads: ResB(160, 3) + Linear(160->9) -> softmax + 1e-8; ResB + Linear(160->1) -> sigmoid         :30-49,169-178
// The reference runs it with batch 1 on the host; here a "row" is one agent and all matrix work is f32 MFMA
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain), so results differ from
// libtorch only by summation order.
//
// Kernels
//   k_gemm<WAVES, MODE>   C[M][N] = A[M][K] * W[N][K]^T (+ bias), LDS-tiled, 32 rows x 160 columns per wave
//                         (5 accumulator tiles of 32x32; or 5 waves x 1 tile when M is small), K tile 16 or 32,
//                         global->register prefetch, double-buffered LDS, operand fragments read one product ahead.
//                         MODE selects how a row of A is addressed: a dense row, or the im2col row of a 3x3/stride-2
//                         convolution gathered on the fly from an NHWC activation (conv1..3) or from the NCHW
//                         observation buffer (conv0).  Activations between the convolutions are kept NHWC, which is
//                         simply the row-major C of the previous GEMM; the host permutes conv1..3's weights to the
//                         matching (ky, kx, cin) K-order once at load time.
//   k_*                   one wavefront per agent row for the 160-wide normalisations, the GRU gate math, the
//                         residual blocks' relu+skip and the two heads.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/strikeforce_policy.h"
#include "sf_host.hpp"

namespace sfp {

using sf::fail;

constexpr int HID = SF_POLICY_HIDDEN;    // 160
constexpr int ACT = SF_POLICY_ACTIONS;   // 9
constexpr int G3 = 3 * HID;              // 480 gate rows r,z,n
constexpr int OBS_C = SF_OBS_CHANNELS;   // 32
constexpr int OBS_W = SF_OBS_WINDOW;     // 31
constexpr int OBS_F = SF_OBS_FLOATS;     // 30752
constexpr int POV = SF_POLICY_POV;       // 169
constexpr int COMB = 2 * HID + ACT;      // 329 inputs of combined_processor
constexpr int COMB_PAD = 352;            // padded to a multiple of the GEMM's K tiles (16 and 32)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_DENSE = 0, MODE_NHWC = 1, MODE_NCHW = 2 };

struct Gemm {
  const float *A;
  const float *W;     // [N][K] row-major
  const float *bias;  // [N] or null
  float *C;           // [M][ldc]
  int M, N, K;        // K % 32 == 0, N % 160 == 0
  int lda, ldc;
  int S, Cin, So;     // convolution modes: input side, input channels, output side
  // work split: the (row tile, K tile) units of one 160-column strip, numbered tile-major, are dealt to the
  // gridDim.x blocks in contiguous runs of unit_base (+1 for the first unit_rem blocks) units
  int ntiles, unit_base, unit_rem;
  float *part;        // [gridDim.y][gridDim.x][2][BM*160] partial tiles of runs that start or end inside a tile
  // an independent second product of the same shape, computed by the blocks with blockIdx.z == 1 (one launch for
  // the two gate products of a GRU cell, or for the same layer of the two heads)
  const float *A2, *W2, *bias2;
  float *C2;
  const void *W3;     // k_gemm_b3: W split into bf16 hi / mid / lo parts (split_weights)
};

constexpr int BN = 160;

__device__ inline f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

__device__ inline int run_start(const Gemm &g, int b) { return b * g.unit_base + (b < g.unit_rem ? b : g.unit_rem); }
__device__ inline int run_owner(const Gemm &g, int u) {
  const int big = g.unit_rem * (g.unit_base + 1);
  return u < big ? u / (g.unit_base + 1) : g.unit_rem + (u - big) / g.unit_base;
}

// Block tile (WM*32) x 160, K tile BKT.  WN = 1: a wave owns 32 rows x 160 columns (5 accumulator tiles; the
// large-M shape).  WN = 5: the five 32-column tiles of the same 32 rows go to five waves (the small-M shape: five
// times the waves for the same work, each with a fifth of the dependent MFMA chain).
//
// A block walks a contiguous run of (row tile, K tile) units ("stream-K"): with one tile per run this is the
// classic one-block-per-tile GEMM; with gridDim.x = the number of resident blocks the chip holds, every block gets
// the same number of MFMAs whatever M is, the global->LDS->MFMA pipeline never drains between row tiles, and a tile
// whose K range is cut by a run boundary is finished by k_gemm_fixup, which adds the partial tiles in K order
// (deterministic: no atomics).
template <int WM, int WN, int BKT, int MODE>
__global__ __launch_bounds__(WM *WN * 64, 2) void k_gemm(Gemm g) {
  if (blockIdx.z) g.A = g.A2, g.W = g.W2, g.bias = g.bias2, g.C = g.C2;
  constexpr int T = WM * WN * 64, BM = WM * 32, LD = BKT + 1;
  constexpr int NT = 5 / WN;                    // 32-column tiles per wave
  constexpr int Q = BKT / 4;                    // float4 per tile row
  constexpr int AJ = (BM * Q + T - 1) / T;      // float4 loads of A per thread
  constexpr int BJ = (BN * Q + T - 1) / T;      // float4 loads of W per thread
  constexpr int SJ = (BM * BKT + T - 1) / T;    // scalar loads of A per thread (NCHW gather)
  constexpr int KS = BKT / 2;                   // 32x32x2 products per tile
  constexpr int GS = KS / 2;                    // groups of two products
  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BN * LD];

  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int wm = w / WN, wn = w - wm * WN;
  const int n0 = blockIdx.y * BN;
  const int KT = g.K / BKT;
  const int u0 = run_start(g, blockIdx.x);
  const int nu = g.unit_base + ((int)blockIdx.x < g.unit_rem ? 1 : 0);
  if (nu == 0) return;

  // ---- where this thread's share of an A tile comes from (recomputed when the load cursor enters a new row tile) ----
  const float *arow[AJ];
  // NCHW: consecutive threads walk consecutive rows, one row per thread.  The address is split into a wave-uniform
  // part (the tile's first agent + the (cin, ky, kx) offset of the k being loaded: SALU, lands in the load's saddr)
  // and a 32-bit per-lane part (this row's pixel relative to that agent), so a gathered element costs no VALU.
  const float *abase = nullptr;
  uint32_t avoff = 0;
  auto setrow = [&](int tile) {
    const int m0 = tile * BM;
    if (MODE == MODE_NCHW) {
      int m = m0 + (t % BM);
      if (m >= g.M) m = g.M - 1;
      const int so2 = g.So * g.So;
      const int b0 = m0 / so2;
      const int b = m / so2, r = m - b * so2, oy = r / g.So, ox = r - oy * g.So;
      abase = g.A + (size_t)b0 * g.Cin * g.S * g.S;
      avoff = (uint32_t)(((b - b0) * g.Cin * g.S * g.S + (2 * oy) * g.S + 2 * ox) * 4);
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int f = t + j * T;
        int m = m0 + ((f / Q < BM) ? f / Q : BM - 1);
        if (m >= g.M) m = g.M - 1;
        if (MODE == MODE_DENSE) {
          arow[j] = g.A + (size_t)m * g.lda + (f % Q) * 4;
        } else {
          const int so2 = g.So * g.So;
          const int b = m / so2, r = m - b * so2, oy = r / g.So, ox = r - oy * g.So;
          arow[j] = g.A + ((size_t)(b * g.S + 2 * oy) * g.S + 2 * ox) * g.Cin + (f % Q) * 4;
        }
      }
    }
  };
  const float *brow[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int f = t + j * T;
    const int n = (f / Q < BN) ? f / Q : BN - 1;
    brow[j] = g.W + (size_t)(n0 + n) * g.K + (f % Q) * 4;
  }

  f32x4 ra[AJ], rb[BJ];
  float rs[SJ];

  auto gload = [&](int kt) {
    const int k0 = kt * BKT;
    if (MODE == MODE_NCHW) {
      const int ss = g.S * g.S;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int e = t + j * T;
        int kl = (e / BM < BKT) ? e / BM : BKT - 1;
        if (BM % 64 == 0) kl = __builtin_amdgcn_readfirstlane(kl);  // a wave's 64 rows share k
        const int k = k0 + kl;
        const int cin = k / 9, tap = k - cin * 9, ky = tap / 3, kx = tap - ky * 3;
        const char *sb = reinterpret_cast<const char *>(abase + ((size_t)cin * ss + ky * g.S + kx));
        rs[j] = *reinterpret_cast<const float *>(sb + avoff);
      }
    } else {
      int off = k0;
      if (MODE == MODE_NHWC) {
        const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin, ky = tap / 3, kx = tap - ky * 3;
        off = (ky * g.S + kx) * g.Cin + c0;
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) ra[j] = ldg4(arow[j] + off);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) rb[j] = ldg4(brow[j] + k0);
  };
  auto lstore = [&](int buf) {
    if (MODE == MODE_NCHW) {
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int e = t + j * T;
        if ((BM * BKT) % T == 0 || e < BM * BKT) As[buf][(e % BM) * LD + e / BM] = rs[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int f = t + j * T;
        if ((BM * Q) % T == 0 || f < BM * Q) {
          float *d = &As[buf][(f / Q) * LD + (f % Q) * 4];
          d[0] = ra[j].x, d[1] = ra[j].y, d[2] = ra[j].z, d[3] = ra[j].w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int f = t + j * T;
      if ((BN * Q) % T == 0 || f < BN * Q) {
        float *d = &Bs[buf][(f / Q) * LD + (f % Q) * 4];
        d[0] = rb[j].x, d[1] = rb[j].y, d[2] = rb[j].z, d[3] = rb[j].w;
      }
    }
  };

  f32x16 acc[NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  };
  // The products are issued transposed (W fragment as the MFMA's A operand, activation fragment as its B operand:
  // the two operand lane maps are the same, so the fragments need no change), which puts the output row m on the
  // lane (m = l&31) and four consecutive output columns n = 8*(reg>>2) + 4*(l>>5) + (reg&3) in consecutive
  // registers: a tile leaves as 4 dwordx4 stores per lane instead of 16 dword stores.
  auto flush = [&](int tile, bool whole, int slot) {
    const int ml = wm * 32 + (l & 31), m = tile * BM + ml;
    float *pt = g.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + slot) * (BM * BN) + ml * BN;
    float *cr = g.C + (size_t)m * g.ldc + n0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = (wn * NT + nt) * 32 + 8 * q + 4 * (l >> 5);
        f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
        if (whole) {
          if (g.bias) v += ldg4(g.bias + n0 + nl);
          if (m < g.M) *reinterpret_cast<f32x4 *>(cr + nl) = v;
        } else {
          *reinterpret_cast<f32x4 *>(pt + nl) = v;
        }
      }
    }
  };

  // Schedule of one unit (the compiler is held to it with sched_barrier): the unit's products run in GS groups of
  // two (k, k+2 -> one ds_read2_b32 per operand); the fragments of group i+1 are read from LDS before the MFMAs of
  // group i issue.  Half-way through, the next unit (in registers since the previous unit) is written to the other
  // LDS buffer and the loads of the unit after it are issued; the block's only barrier comes before the last
  // group, followed by the read of the next unit's first fragments, so both hide behind that group's MFMAs.
  float fa[2][2], fb[2][NT][2];
  auto fload = [&](int buf, int grp, int slot) {
    const float *as = &As[buf][(wm * 32 + (l & 31)) * LD + (l >> 5)];
    const float *bs = &Bs[buf][(wn * NT * 32 + (l & 31)) * LD + (l >> 5)];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      fa[slot][h] = as[4 * grp + 2 * h];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[slot][nt][h] = bs[nt * 32 * LD + 4 * grp + 2 * h];
    }
  };

  // load cursor (two units ahead of the MFMAs) and compute cursor
  int tile_l = u0 / KT, kt_l = u0 - tile_l * KT;
  int tile_c = tile_l, kt_c = kt_l, seg_kt0 = kt_l;
  bool seg_first = true;
  auto load_next = [&]() {
    gload(kt_l);
    if (++kt_l == KT) {
      kt_l = 0;
      ++tile_l;
      if (tile_l < g.ntiles) setrow(tile_l);
    }
  };
  setrow(tile_l);
  load_next();
  lstore(0);
  __syncthreads();
  if (nu > 1) load_next();
  fload(0, 0, 0);
  zero_acc();
  for (int i = 0; i < nu; ++i) {
    const int buf = i & 1;
#pragma unroll
    for (int grp = 0; grp < GS; ++grp) {
      if (grp == GS / 2 && i + 1 < nu) {
        lstore(buf ^ 1);
        if (i + 2 < nu) load_next();
      }
      if (grp + 1 < GS) {
        fload(buf, grp + 1, (grp + 1) & 1);
      } else if (i + 1 < nu) {
        __syncthreads();
        fload(buf ^ 1, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[grp & 1][nt][h], fa[grp & 1][h], acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt_c == KT - 1 || i == nu - 1) {
      flush(tile_c, seg_kt0 == 0 && kt_c == KT - 1, seg_first ? 0 : 1);
      zero_acc();
      seg_first = false;
      seg_kt0 = 0;
    }
    if (++kt_c == KT) kt_c = 0, ++tile_c;
  }
}

// Finishes the row tiles whose K range was cut by a run boundary: C = bias + the partial tiles in K order.  One
// block per run boundary; the boundary that is the first one inside its tile does the tile.
template <int BM>
__global__ __launch_bounds__(256) void k_gemm_fixup(Gemm g, int KT, int G) {
  const int b_lo = blockIdx.x, n0 = blockIdx.y * BN;
  const int cut = run_start(g, b_lo + 1);  // first unit of the next run
  if (cut % KT == 0) return;               // the boundary coincides with a tile boundary
  const int tile = cut / KT, ua = tile * KT;
  if (run_owner(g, ua) != b_lo) return;    // an earlier boundary inside the same tile owns it
  const int b_hi = run_owner(g, ua + KT - 1);
  const int m0 = tile * BM;
  constexpr int V = BM * BN / 4 / 256;     // float4 per thread
  f32x4 s[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b = b_lo; b <= b_hi; ++b) {
    const int slot = run_start(g, b) >= ua ? 0 : 1;  // a run's first segment is in slot 0, a later one in slot 1
    const float *pt = g.part + (((size_t)blockIdx.y * G + b) * 2 + slot) * (BM * BN);
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] += ldg4(pt + (threadIdx.x + 256 * j) * 4);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int e = (threadIdx.x + 256 * j) * 4, ml = e / BN, nl = e - ml * BN;
    if (m0 + ml < g.M) {
      f32x4 v = s[j];
      if (g.bias) v += ldg4(g.bias + n0 + nl);
      *reinterpret_cast<f32x4 *>(g.C + (size_t)(m0 + ml) * g.ldc + n0 + nl) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_gemm_b3: the large-M products (conv1, conv2) on the bf16 matrix pipe at f32-level accuracy.
//
// Every f32 operand is written as hi + mid + lo, three bf16 numbers obtained by round-to-nearest of the running
// residual (x - hi and x - hi - mid are exact in f32, so hi + mid + lo == x: 3 x 8 significand bits and two signs cover
// f32's 24).  A product a*w then expands into nine bf16 products; the six of relative size >= 2^-16 are kept
//     hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)
// and the dropped ones (mid*lo, lo*mid, lo*lo) are <= 3 * 2^-24 |a*w|: the order of one f32 rounding of the
// product.  Each kept product is exact in the f32 accumulator (8 x 8 bits), so what differs from the f32 pipe is that
// dropped tail and the summation order.  v_mfma_f32_32x32x16_bf16 retires 16 times the products per cycle of
// v_mfma_f32_32x32x2_f32: six of them instead of eight f32 instructions per 32x32x16 block = 2.67 x fewer
// matrix-pipe cycles.  Non-finite inputs come out as NaN (inf - inf in the residual).
//
// W is split once on the host into the image the LDS wants (one 96-byte record [hi 16][mid 16][lo 16] per (K tile of
// 16, output column)); activations are split as they are stored to LDS (v_cvt_pk_bf16_f32 + shifts, ~5.5 VALU per
// element).  Block tile 256 rows x 160 columns x 16, one block per CU: eight compute waves of 32 rows x 160 columns
// (two per SIMD) and four loader waves, two LDS stages of 39 KB.
// Work split, pipeline and tile hand-over (stream-K runs, k_gemm_fixup) as in k_gemm.
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 8 compute waves (32 rows x 160 columns each, two per SIMD) + 4 loader waves (one per SIMD)
constexpr int B3_CW = 8, B3_LW = 4, B3_BM = 32 * B3_CW, B3_BK = 16, B3_T = 64 * (B3_CW + B3_LW), B3_LT = 64 * B3_LW, B3_RS = 96;
constexpr int B3_A_BYTES = B3_BM * B3_RS, B3_W_BYTES = BN * B3_RS, B3_STAGE = B3_A_BYTES + B3_W_BYTES;
constexpr int B3_LDS = 2 * B3_STAGE;                 // 79 872 bytes
constexpr int B3_W_PIECES = BN * 96 / 16;            // 16-byte pieces of one K tile of the W image: 960
constexpr int B3_SETS = 4;                           // register sets of a loader thread = units in flight from HBM

__device__ inline uint32_t pk_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
// four f32 -> their hi / mid / lo bf16 parts, packed in k order
__device__ inline void split3(const f32x4 x, u32x2 &hi, u32x2 &mid, u32x2 &lo) {
  float r[4] = {x.x, x.y, x.z, x.w};
  uint32_t o[3][2];
#pragma unroll
  for (int lvl = 0; lvl < 3; ++lvl)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t pk = pk_bf16(r[2 * h], r[2 * h + 1]);
      o[lvl][h] = pk;
      if (lvl < 2) {
        // (asm: the compiler would pair these into v_pk_add_f32, which costs an MFMA-paced wave ~6x a plain one)
        asm("v_sub_f32 %0, %0, %1" : "+v"(r[2 * h]) : "v"(pk << 16));
        asm("v_sub_f32 %0, %0, %1" : "+v"(r[2 * h + 1]) : "v"(pk & 0xffff0000u));
      }
    }
  hi = u32x2{o[0][0], o[0][1]}, mid = u32x2{o[1][0], o[1][1]}, lo = u32x2{o[2][0], o[2][1]};
}

// W [N][K] f32 -> k_gemm_b3's image [N / 160][K / 16][160][hi 16 | mid 16 | lo 16] (sf_policy_gemm_split; the
// network's own weights are split on the host by split_weights, same arithmetic)
__global__ void k_split_weights(const float *W, uint16_t *img, int N, int K) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)N * K) return;
  const int n = (int)(e / K), k = (int)(e - (size_t)n * K);
  float r = W[e];
  uint16_t part[3];
#pragma unroll
  for (int lvl = 0; lvl < 3; ++lvl) {
    const uint32_t pk = pk_bf16(r, 0.f);
    part[lvl] = (uint16_t)pk;
    r -= __builtin_bit_cast(float, pk << 16);
  }
  const size_t rec = (((size_t)(n / BN) * (K / B3_BK) + k / B3_BK) * BN + n % BN) * 48 + k % B3_BK;
  img[rec] = part[0], img[rec + 16] = part[1], img[rec + 32] = part[2];
}

template <int MODE>
__global__ __launch_bounds__(B3_T, 1) void k_gemm_b3(Gemm g) {
  constexpr int BM = B3_BM, NT = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char b3_lds[];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int n0 = blockIdx.y * BN;
  const int KT = g.K / B3_BK;
  const int u0 = run_start(g, blockIdx.x);
  const int nu = g.unit_base + ((int)blockIdx.x < g.unit_rem ? 1 : 0);
  if (nu == 0) return;
  // LDS image of a tile: row r = 96 bytes [hi 32][mid 32][lo 32], the two 16-byte k halves of each part swapped on rows
  // with bit 3 set.  Conflict-free for every access: the sixteen rows of a ds_read_b128 lane group land on sixteen
  // different 16-byte slots (6 r mod 16 alone would only reach the eight even ones), four rows of a ds_write_b64 group
  // tile the 128-byte bank window (96 r mod 128 = 0, 96, 64, 32), and the W pieces stay contiguous.
  if (w >= B3_CW) {
    // ------------------------------------------------------------------------------------------------------
    // Loader waves.  Unit u's operands go HBM / L2 -> registers (B3_SETS units in flight per thread) -> split -> LDS
    // stage u & 1, one unit ahead of the compute waves.  What a block moves per unit (16 KB of A + 15 KB of W) is what
    // bounds this kernel: a CU's vector-memory path delivered ~25 B/clk here (in-kernel stamps: 8 load instructions took
    // a loader wave ~1300 cycles to issue), about one unit's bytes per unit's MFMA time.  With the same loads, splits and
    // stores done as fillers between the compute waves' MFMAs, in-order issue put every stall of that path in front of
    // matrix instructions (57-66 % pipe use); here they stay in these four waves.  Every load is issued whatever the run
    // length (a unit past the run's end reads clamped, valid addresses and lands in the idle stage): with one static
    // instruction stream the compiler's vmcnt leaves the younger sets in flight.
    // ------------------------------------------------------------------------------------------------------
    const int lt = t - B3_CW * 64;
    constexpr int AJ = BM * 4 / B3_LT;                          // float4 of A per thread: 4 (rows lt/4 + 64 j)
    constexpr int NP = (B3_W_PIECES + B3_LT - 1) / B3_LT;       // W pieces per thread: 4 (the last one for lt < 192)
    const bool wlast = lt + (NP - 1) * B3_LT < B3_W_PIECES;
    const float *arow[AJ];
    auto setrow = [&](int tile) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        int m = tile * BM + (lt >> 2) + (BM / AJ) * j;
        if (m >= g.M) m = g.M - 1;
        if (MODE == MODE_DENSE) {
          arow[j] = g.A + (size_t)m * g.lda + (lt & 3) * 4;
        } else {
          const int so2 = g.So * g.So;
          const int b = m / so2, r = m - b * so2, oy = r / g.So, ox = r - oy * g.So;
          arow[j] = g.A + ((size_t)(b * g.S + 2 * oy) * g.S + 2 * ox) * g.Cin + (lt & 3) * 4;
        }
      }
    };
    const u32x4 *wimg = reinterpret_cast<const u32x4 *>(g.W3) + (size_t)blockIdx.y * KT * B3_W_PIECES;
    auto wdst = [](int piece) { const int n = piece / 6, c = piece % 6; return B3_A_BYTES + n * B3_RS + ((c ^ ((n >> 3) & 1)) * 16); };
    int wdstp[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) wdstp[q] = wdst(lt + q * B3_LT);
    const int adst = (lt >> 2) * B3_RS + ((((lt & 3) >> 1) ^ ((lt >> 5) & 1)) * 16) + (lt & 1) * 8;

    f32x4 ra[B3_SETS][AJ];
    u32x4 rb[B3_SETS][NP];
    int tile_l = u0 / KT, kt_l = u0 - tile_l * KT;
    auto load_next = [&](auto set) {
      constexpr int R = decltype(set)::value;
      int off = kt_l * B3_BK;
      if (MODE == MODE_NHWC) {
        const int tap = off / g.Cin, c0 = off - tap * g.Cin, ky = tap / 3, kx = tap - ky * 3;
        off = (ky * g.S + kx) * g.Cin + c0;
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) ra[R][j] = ldg4(arow[j] + off);
      const u32x4 *ws = wimg + (size_t)kt_l * B3_W_PIECES;
#pragma unroll
      for (int q = 0; q < NP; ++q) rb[R][q] = ws[q + 1 < NP || wlast ? lt + q * B3_LT : lt];
      if (++kt_l == KT) {
        kt_l = 0;
        ++tile_l;
        if (tile_l < g.ntiles) setrow(tile_l);
      }
    };
    auto lstore = [&](int stage, auto set) {
      constexpr int R = decltype(set)::value;
      unsigned char *base = b3_lds + stage * B3_STAGE;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        u32x2 hi, mid, lo;
        split3(ra[R][j], hi, mid, lo);
        unsigned char *d = base + adst + j * ((BM / AJ) * B3_RS);
        *reinterpret_cast<u32x2 *>(d) = hi;
        *reinterpret_cast<u32x2 *>(d + 32) = mid;
        *reinterpret_cast<u32x2 *>(d + 64) = lo;
      }
#pragma unroll
      for (int q = 0; q < NP; ++q)
        if (q + 1 < NP || wlast) *reinterpret_cast<u32x4 *>(base + wdstp[q]) = rb[R][q];
    };
    // during unit i (between the barriers of units i - 1 and i): unit i + 1 -> stage (i + 1) & 1, then the loads of
    // unit i + 1 + B3_SETS into the freed set
    auto step = [&](int i, auto set) {
      lstore((i + 1) & 1, set);
      load_next(set);
      __syncthreads();
    };
    setrow(tile_l);
    load_next(std::integral_constant<int, 0>());
    lstore(0, std::integral_constant<int, 0>());
    load_next(std::integral_constant<int, 1>());
    load_next(std::integral_constant<int, 2>());
    load_next(std::integral_constant<int, 3>());
    load_next(std::integral_constant<int, 0>());
    __syncthreads();
    int i = 0;
    for (; i + 3 < nu; i += 4) {
      step(i, std::integral_constant<int, 1>());
      step(i + 1, std::integral_constant<int, 2>());
      step(i + 2, std::integral_constant<int, 3>());
      step(i + 3, std::integral_constant<int, 0>());
    }
    if (i < nu) step(i, std::integral_constant<int, 1>());
    if (i + 1 < nu) step(i + 1, std::integral_constant<int, 2>());
    if (i + 2 < nu) step(i + 2, std::integral_constant<int, 3>());
    return;
  }

  // ----------------------------------------------------------------------------------------------------------
  // Compute waves: fragments from LDS and MFMAs, nothing else.  One unit = one K tile of 16 = five groups (one per
  // 32-column tile) of six MFMAs.  The W fragments of group n + 1 are read behind the first MFMA of group n; behind
  // the first MFMA of group 4 the wave releases the stage it has now read completely, checks that the loaders have
  // filled the other one and reads the next unit's first fragments.  No barrier: with one per unit, the two waves
  // of a SIMD met at it with nothing queued and the matrix pipe drained once per unit (232 -> 205 TFLOP/s f32-equivalent
  // with the loaders switched off, against 306 for the same loop without the barrier).  sched_barrier pins the
  // order; ten groups make one period of the fragment slots, so the loop body is two units.
  // ----------------------------------------------------------------------------------------------------------
  f32x16 acc[NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  };
  // as in k_gemm: the W fragment is the MFMA's first operand, so a lane ends up with row m = l & 31 and four
  // consecutive columns per register quad
  auto flush = [&](int tile, bool whole, int slot) {
    const int ml = w * 32 + (l & 31), m = tile * BM + ml;
    float *pt = g.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + slot) * (BM * BN) + ml * BN;
    float *cr = g.C + (size_t)m * g.ldc + n0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = nt * 32 + 8 * q + 4 * (l >> 5);
        f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
        if (whole) {
          if (g.bias) v += ldg4(g.bias + n0 + nl);
          if (m < g.M) *reinterpret_cast<f32x4 *>(cr + nl) = v;
        } else {
          *reinterpret_cast<f32x4 *>(pt + nl) = v;
        }
      }
    }
  };
  // fragments: lane l holds k = 8 (l >> 5) .. +7 of row (l & 31), one ds_read_b128 per part
  bf16x8 fa[2][3], fw[2][3];
  const int frag = (l & 31) * B3_RS + (((l >> 5) ^ ((l >> 3) & 1)) * 16);
  auto read_a = [&](int stage, int slot) {
    const unsigned char *s = b3_lds + stage * B3_STAGE + w * 32 * B3_RS + frag;
#pragma unroll
    for (int part = 0; part < 3; ++part) fa[slot][part] = *reinterpret_cast<const bf16x8 *>(s + part * 32);
  };
  auto read_w = [&](int stage, int nt, int slot) {
    const unsigned char *s = b3_lds + stage * B3_STAGE + B3_A_BYTES + nt * 32 * B3_RS + frag;
#pragma unroll
    for (int part = 0; part < 3; ++part) fw[slot][part] = *reinterpret_cast<const bf16x8 *>(s + part * 32);
  };
  int tile_c = u0 / KT, kt_c = u0 - tile_c * KT, seg_kt0 = kt_c;
  bool seg_first = true;
  auto unit = [&](int i, auto parity) {
    constexpr int P = decltype(parity)::value;
    const int st = P;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int gslot = (P * NT + nt) & 1;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        // smallest terms first
        const int wp = k == 0 ? 2 : (k == 2 || k == 3) ? 1 : 0, ap = k == 1 ? 2 : (k == 2 || k == 4) ? 1 : 0;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[gslot][wp], fa[P][ap], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k == 0) {
          if (nt + 1 < NT) read_w(st, nt + 1, gslot ^ 1);
          else read_w(st ^ 1, 0, gslot ^ 1);
        } else if (nt == 3 && k == 1) {
          __syncthreads();
          read_a(st ^ 1, P ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kt_c == KT - 1 || i == nu - 1) {
      flush(tile_c, seg_kt0 == 0 && kt_c == KT - 1, seg_first ? 0 : 1);
      zero_acc();
      seg_first = false;
      seg_kt0 = 0;
    }
    if (++kt_c == KT) kt_c = 0, ++tile_c;
  };
  __syncthreads();
  read_a(0, 0);
  read_w(0, 0, 0);
  zero_acc();
  int i = 0;
  for (; i + 1 < nu; i += 2) {
    unit(i, std::integral_constant<int, 0>());
    unit(i + 1, std::integral_constant<int, 1>());
  }
  if (i < nu) unit(i, std::integral_constant<int, 0>());
}

// ---------------------------------------------------------------------------------------------------------
// conv0 on the observation as it really is: 0.8 % non-zero (a 31x31 window of mostly empty cells, 32 features each).
// One 16-wave workgroup per agent keeps the agent's whole conv0 output (15 x 15 x 160 f32 = 144 KB) in LDS, streams
// the 123 KB observation once, appends its non-zeros to an LDS list in scan order (block-wide prefix sum: the list
// order, hence the order of every f32 sum, is deterministic), and applies each (channel, y, x, value) to the <= 4
// output pixels whose 3x3/stride-2 window contains it: out[oy][ox][n] += value * W[n][c][ky][kx], n on the lanes.
// An output pixel belongs to the wavefront (oy & 3, ox & 3), so there are no atomics; a wavefront looks at 64 list
// entries at a time (one per lane: is one of its targets mine?), then walks its own ones in list order, weight rows
// fetched four targets ahead.  Work is proportional to the non-zeros (~250 per agent, against 65 000 products per
// output channel in the dense form); any density is handled (the list is flushed when full).
// ---------------------------------------------------------------------------------------------------------
constexpr int C0_OUT = 15, C0_ACC = C0_OUT * C0_OUT * HID;  // 36000 floats
constexpr int C0_T = 1024, C0_WAVES = C0_T / 64;
constexpr int C0_LCAP = 2048;
static_assert(C0_LCAP == SF_POLICY_LIST_MAX, "the entry points bound cap by the kernels' list limit");
constexpr size_t C0_LDS = (size_t)C0_ACC * 4 + (size_t)C0_LCAP * 8 + 4 * (2 * 16 * C0_WAVES + 4);

// LIST: the non-zeros arrive as a list (sf_observe_sparse_device: same keys, same order as the scan below builds), so the
// 123 KB scan of the dense observation is gone; `obs` is unused.
struct C0List {
  const uint32_t *keys;
  const float *vals;
  const uint32_t *counts;
  int cap;
  uint32_t *overflows;  // bumped once per agent whose list did not fit (it is then evaluated on an empty list)
};
template <bool LIST>
__global__ __launch_bounds__(C0_T) void k_conv0_sparse(const float *obs, const float *wt, float *act0, int agents, C0List li) {
  extern __shared__ __attribute__((aligned(16))) float c0_lds[];
  float *acc = c0_lds;
  float *lval = acc + C0_ACC;
  uint32_t *lkey = reinterpret_cast<uint32_t *>(lval + C0_LCAP);
  uint32_t *cnt = lkey + C0_LCAP;          // [16 pieces][16 waves] non-zero counts
  uint32_t *offs = cnt + 16 * C0_WAVES;    // their exclusive prefix sums, then the total
  const int t = threadIdx.x, w = t >> 6, l = t & 63;

  const int cy = w >> 2, cx = w & 3;  // this wavefront's output pixels: oy % 4 == cy, ox % 4 == cx
  // the candidate along one axis: input coordinate v lies in the windows of outputs (v - k) / 2 for k == v (mod 2);
  // at most one of them is congruent to `cls` modulo 4.  Returns the output coordinate or -1, and k.
  auto axis = [](int v, int cls, int &k) -> int {
    if (v & 1) {
      k = 1;
      const int o = (v - 1) >> 1;
      return ((o & 3) == cls && o < C0_OUT) ? o : -1;
    }
    const int o0 = v >> 1;  // k = 0
    if ((o0 & 3) == cls) {
      k = 0;
      return o0 < C0_OUT ? o0 : -1;
    }
    k = 2;
    const int o2 = o0 - 1;
    return (o2 >= 0 && (o2 & 3) == cls) ? o2 : -1;
  };
  auto process = [&](uint32_t n) {
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
      const uint32_t e = e0 + (uint32_t)l;
      uint32_t rowo = 0, wro = 0;
      float val = 0.f;
      bool mine = false;
      if (e < n) {
        const uint32_t key = lkey[e];
        int ky, kx;
        const int oy = axis((int)((key >> 9) & 31u), cy, ky), ox = axis((int)((key >> 14) & 31u), cx, kx);
        mine = oy >= 0 && ox >= 0;
        rowo = (uint32_t)((oy * C0_OUT + ox) * HID);
        wro = (uint32_t)(((int)(key & 511u) + ky * 3 + kx) * HID);
        val = lval[e];
      }
      uint64_t m = __builtin_amdgcn_ballot_w64(mine);
      while (m) {  // four of this wavefront's targets at a time: all weight loads first, then the LDS updates in order
        uint32_t ro[4], nt = 0;
        float vv[4], wv[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ro[q] = 0, vv[q] = 0.f;
          wv[q][0] = wv[q][1] = wv[q][2] = 0.f;
          if (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1ull;
            ro[q] = (uint32_t)__builtin_amdgcn_readlane((int)rowo, src);
            const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)wro, src);
            vv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), src));
            const float *wr = wt + wq;
            wv[q][0] = wr[l], wv[q][1] = wr[l + 64];
            if (l < HID - 128) wv[q][2] = wr[l + 128];
            nt = (uint32_t)q + 1u;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((uint32_t)q < nt) {
            float *row = acc + ro[q];
            row[l] = fmaf(vv[q], wv[q][0], row[l]);
            row[l + 64] = fmaf(vv[q], wv[q][1], row[l + 64]);
            if (l < HID - 128) row[l + 128] = fmaf(vv[q], wv[q][2], row[l + 128]);
          }
      }
    }
  };

  if (LIST) {
    // entries of the next agent wait in registers (two per thread cover C0_LCAP) while this one's are applied
    constexpr int EPT = C0_LCAP / C0_T;
    uint32_t pk[EPT], pn = 0;
    float pvv[EPT];
    auto fetch = [&](int b) {
      pn = li.counts[b];
      if (pn > (uint32_t)C0_LCAP || pn > (uint32_t)li.cap) {  // (the 0xffffffff marker of a crowded window too)
        pn = 0u;
        if (t == 0 && li.overflows) atomicAdd(li.overflows, 1u);  // null: a dense fallback launch redoes this agent
      }
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const uint32_t e = (uint32_t)(k * C0_T + t);
        pk[k] = 0u, pvv[k] = 0.f;
        if (e < pn) pk[k] = li.keys[(size_t)b * li.cap + e], pvv[k] = li.vals[(size_t)b * li.cap + e];
      }
    };
    if ((int)blockIdx.x < agents) fetch((int)blockIdx.x);
    for (int b = (int)blockIdx.x; b < agents; b += (int)gridDim.x) {
      for (int i = t; i < C0_ACC / 4; i += C0_T) reinterpret_cast<f32x4 *>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const uint32_t n = pn;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const uint32_t e = (uint32_t)(k * C0_T + t);
        if (e < n) lkey[e] = pk[k], lval[e] = pvv[k];
      }
      __syncthreads();
      if (b + (int)gridDim.x < agents) fetch(b + (int)gridDim.x);
      process(n);
      __syncthreads();
      f32x4 *dst = reinterpret_cast<f32x4 *>(act0 + (size_t)b * C0_ACC);
      for (int i = t; i < C0_ACC / 4; i += C0_T) dst[i] = reinterpret_cast<const f32x4 *>(acc)[i];
      __syncthreads();  // the tile has been read out before the next agent zeroes it
    }
    return;
  }
  // The workgroup is persistent (one per CU: the output tile fills its LDS) and walks agents b, b + gridDim.x, ...:
  // the next agent's observation is requested as soon as this one's has been scanned, so its latency passes under
  // this agent's list processing and write-back, and the write-back's stores drain under the next agent's scan.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NV = OBS_F / 2;  // 15376 8-byte pieces: a chunk adds at most 2 * 1024 = C0_LCAP entries
  constexpr int NIT = (NV + C0_T - 1) / C0_T;  // 16 pieces per thread, all requested before the first is looked at
  f32x2 pre[NIT];
  auto request = [&](int b) {
    const f32x2 *src = reinterpret_cast<const f32x2 *>(obs + (size_t)b * OBS_F);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = k * C0_T + t;
      pre[k] = f32x2{0.f, 0.f};
      if (i < NV) pre[k] = __builtin_nontemporal_load(src + i);
    }
  };
  // li.counts given (the fallback launch behind a list-form forward): only the agents whose list did not fit — count
  // beyond cap or the kernel's list, or the crowded-window marker — are redone from the dense buffer; the others keep
  // what the list launch wrote.  nxt(b): the first such agent among b, b + gridDim.x, ... (uniform over the workgroup)
  auto nxt = [&](int b) {
    if (li.counts)
      while (b < agents && !(li.counts[b] > (uint32_t)C0_LCAP || li.counts[b] > (uint32_t)li.cap)) b += (int)gridDim.x;
    return b;
  };
  int bnext = nxt((int)blockIdx.x);
  if (bnext < agents) request(bnext);
  for (int b = bnext; b < agents; b = bnext) {
  bnext = nxt(b + (int)gridDim.x);
  for (int i = t; i < C0_ACC / 4; i += C0_T) reinterpret_cast<f32x4 *>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  uint32_t count = 0;  // list length, the same value in every thread
  auto append = [&](int k, uint32_t m, uint32_t pos) {
    if (m) {
      const int i = k * C0_T + t;
      const float vv[2] = {pre[k].x, pre[k].y};
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if ((m >> j) & 1u) {
          const uint32_t idx = 2u * (uint32_t)i + (uint32_t)j;  // = ch * 961 + y * 31 + x
          const uint32_t ch = idx / (uint32_t)(OBS_W * OBS_W), r = idx - ch * (uint32_t)(OBS_W * OBS_W);
          const uint32_t y = r / (uint32_t)OBS_W, x = r - y * (uint32_t)OBS_W;
          lkey[pos] = (ch * 9u) | (y << 9) | (x << 14);
          lval[pos] = vv[j];
          ++pos;
        }
    }
  };
  // Where does every non-zero go?  Wave-level: two ballots per piece (x non-zero, y non-zero) give a lane's offset
  // (mbcnt) and the wave's count (popcount) without any shuffle; the 16 pieces x 16 waves counts go to LDS, wave 0
  // turns them into offsets in (piece, wave) order = scan order, and if everything fits the list (it does, unless the
  // input is not an observation) all entries are written in one go.  Three barriers instead of two per piece.
  uint32_t lanepre[NIT], mpack = 0;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const uint32_t m = (pre[k].x != 0.f ? 1u : 0u) | (pre[k].y != 0.f ? 2u : 0u);
    const uint64_t b0 = __builtin_amdgcn_ballot_w64((m & 1u) != 0u), b1 = __builtin_amdgcn_ballot_w64((m & 2u) != 0u);
    lanepre[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                 __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
    if (l == k) cnt[k * C0_WAVES + w] = (uint32_t)(__builtin_popcountll(b0) + __builtin_popcountll(b1));
    mpack |= m << (2 * k);
  }
  __syncthreads();
  if (w == 0) {  // exclusive prefix of the 256 counts, four per lane
    uint32_t c4[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c4[j] = cnt[4 * l + j], sum += c4[j];
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
      if (l >= o) incl += up;
    }
    uint32_t run = incl - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) offs[4 * l + j] = run, run += c4[j];
    if (l == 63) offs[NIT * C0_WAVES] = incl;
  }
  __syncthreads();
  // write and apply the list; an observation fits in one batch, anything denser goes in as many as it takes
  // (pieces [ks, ke) at a time, ke the furthest piece boundary that still fits: a single piece always does)
  int ks = 0;
  do {
    const uint32_t base = offs[ks * C0_WAVES];
    int ke = ks + 1;
#pragma unroll
    for (int kk = 2; kk <= NIT; ++kk)
      if (kk > ks + 1 && offs[kk * C0_WAVES] - base <= (uint32_t)C0_LCAP) ke = kk;
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (k >= ks && k < ke) append(k, (mpack >> (2 * k)) & 3u, offs[k * C0_WAVES + w] - base + lanepre[k]);
    count = offs[ke * C0_WAVES] - base;
    __syncthreads();
    if (ke == NIT && bnext < agents) request(bnext);  // `pre` is free: fetch the next agent
    process(count);
    __syncthreads();
    ks = ke;
  } while (ks < NIT);
  f32x4 *dst = reinterpret_cast<f32x4 *>(act0 + (size_t)b * C0_ACC);
  for (int i = t; i < C0_ACC / 4; i += C0_T) dst[i] = reinterpret_cast<const f32x4 *>(acc)[i];
  __syncthreads();  // the tile has been read out before the next agent zeroes it
  }
}

// ---------------------------------------------------------------------------------------------------------
// The convolution stack folded into one matrix.
//
// GameCNN::forward (Modules.hpp:66-71) is conv3(conv2(conv1(conv0(x)))): four bias-free convolutions with nothing in
// between, i.e. one LINEAR map of the 32 x 31 x 31 observation onto the 160 features `feat`.  sf_policy_create composes
// the four weight tensors once (k_fold: the transposed convolutions applied to conv3's 160 output rows, in f64 on the
// device) into that map's matrix F [row = x << 10 | y << 5 | channel][160] f32 — 21 MB, which stays in the L2s / the
// infinity cache — and a forward pass is
//     feat = sum over the observation's non-zero floats of  value * F[row]          (~300 of 30 752 are non-zero)
// 0.1 MFLOP per agent instead of 48.6: the 15x15, 7x7 and 3x3 activations never exist, and neither do the three
// largest kernels of the layered path (k_conv0_sparse, two k_gemm_b3 launches), which stays in the library behind
// SF_POLICY_LAYERED=1 as the cross-check that evaluates the layers in the reference's order.
// Arithmetic: one partial sum per pair of channels — an fmaf chain in f32 over that pair's non-zeros in the observation's
// scan order — and the partial sums of the non-empty pairs added in f64 in channel order, one rounding to f32 at the
// end.  List form and dense form follow the same order, so their results are the same bits; against the reference's
// layer-by-layer f32 evaluation the difference is of the size of the reference's own rounding (F's entries are within
// half an ulp of the exact composition, a partial sum has ~20 terms where a convolution output has 288 to 1440).
// ---------------------------------------------------------------------------------------------------------
constexpr int FD_ROWS = 32 * 32 * 32;  // rows of F: x (5 bits), y (5 bits), channel (5 bits); x, y = 31 unused
constexpr int FD_G = OBS_C / 2;        // partial sums: one per pair of channels
constexpr int FD_SEG = 2 * OBS_W * OBS_W;  // floats of the dense observation behind one partial sum

// one transposed 3x3 / stride-2 convolution: Tout[o][y][x][cin] = sum over taps (ky, kx) with (y - ky, x - kx) even and
// inside, and over c:  Tin[o][(y - ky) / 2][(x - kx) / 2][c] * W[c][cin][ky][kx].  W is torch's [c][cin][3][3] (perm 0)
// or this file's [c][tap][cin] (perm 1).
__global__ __launch_bounds__(256) void k_fold(const double *Tin, const float *W, double *Tout, int So, int S, int C, int Cin, int perm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)HID * S * S * Cin) return;
  const int cin = (int)(i % Cin);
  size_t r = i / Cin;
  const int x = (int)(r % S);
  r /= S;
  const int y = (int)(r % S), o = (int)(r / S);
  double acc = 0.0;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = y - ky;
    if (ty < 0 || (ty & 1) || (ty >> 1) >= So) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = x - kx;
      if (tx < 0 || (tx & 1) || (tx >> 1) >= So) continue;
      const double *tin = Tin + (((size_t)o * So + (ty >> 1)) * So + (tx >> 1)) * C;
      const int tap = ky * 3 + kx;
      for (int c = 0; c < C; ++c)
        acc += tin[c] * (double)(perm ? W[((size_t)c * 9 + tap) * Cin + cin] : W[((size_t)c * Cin + cin) * 9 + tap]);
    }
  }
  Tout[i] = acc;
}
__global__ __launch_bounds__(256) void k_to_f64(const float *src, double *dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}
// T0 [o][y][x][channel] f64 -> F [x << 10 | y << 5 | channel][o] f32
__global__ __launch_bounds__(256) void k_fold_out(const double *T0, float *F) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)OBS_W * OBS_W * OBS_C * HID) return;
  const int o = (int)(i % HID);
  size_t r = i / HID;
  const int ch = (int)(r % OBS_C);
  r /= OBS_C;
  const int x = (int)(r % OBS_W), y = (int)(r / OBS_W);
  F[((size_t)((x << 10) | (y << 5) | ch)) * HID + o] = (float)T0[(((size_t)o * OBS_W + y) * OBS_W + x) * OBS_C + ch];
}

// list form: one wavefront per agent walks the agent's list, 64 entries per fetch (one per lane, the next 64 requested
// before these are used); lane l < 40 owns features 4 l .. 4 l + 3 (one 16-byte piece of a row of F), an entry's row and
// value reach all lanes by v_readlane, eight rows are requested before the first is used.  Lanes past the list's end
// carry value 0 and row 0 (fmaf(0, w, c) == c: c starts at +0 and so is never -0), so a batch needs no end test.
__global__ __launch_bounds__(64) void k_feat_list(const float *__restrict__ F, float *__restrict__ feat, int agents, C0List li) {
  const int b = (int)blockIdx.x, l = (int)threadIdx.x;
  if (b >= agents) return;
  uint32_t n = li.counts[b];
  if (n > (uint32_t)C0_LCAP || n > (uint32_t)li.cap) {  // (the 0xffffffff marker of a crowded window too)
    n = 0u;
    if (l == 0 && li.overflows) atomicAdd(li.overflows, 1u);  // null: k_feat_dense redoes this agent
  }
  const uint32_t *__restrict__ keys = li.keys + (size_t)b * li.cap;
  const float *__restrict__ vals = li.vals + (size_t)b * li.cap;
  const float *Fl = F + 4 * (l < HID / 4 ? l : 0);
  double tot[4] = {0.0, 0.0, 0.0, 0.0};
  f32x2 c01 = {0.f, 0.f}, c23 = {0.f, 0.f};  // the open partial sum (v_pk_fma_f32: two features per instruction)
  uint32_t g = 0u;  // the pair of channels of the last entry seen
  uint32_t nkey = (uint32_t)l < n ? keys[l] : 0u;
  float nval = (uint32_t)l < n ? vals[l] : 0.f;
  auto close = [&]() {
    asm volatile("" ::: "memory");  // (a real branch, taken <= 16 times per agent: not selects on every entry)
    tot[0] += (double)c01.x, tot[1] += (double)c01.y, tot[2] += (double)c23.x, tot[3] += (double)c23.y;
    c01 = f32x2{0.f, 0.f}, c23 = f32x2{0.f, 0.f};
  };
  for (uint32_t e0 = 0u; e0 < n; e0 += 64u) {
    const uint32_t key = nkey;
    const float val = nval;
    {
      const uint32_t e = e0 + 64u + (uint32_t)l;
      nkey = e < n ? keys[e] : 0u, nval = e < n ? vals[e] : 0.f;
    }
    const uint32_t ch = ((key & 511u) * 57u) >> 9;  // the key's low field is 9 * channel
    const uint32_t rowo = ((((key >> 9) & 1023u) << 5) | ch) * (uint32_t)HID;
    const uint32_t cnt = n - e0 < 64u ? n - e0 : 64u;
    // which entries open a new pair of channels (the partial sum is closed in front of them)
    const uint32_t gl = ch >> 1;
    uint32_t gprev = (uint32_t)__shfl_up((int)gl, 1, 64);
    if (l == 0) gprev = g;
    const uint64_t opens = __builtin_amdgcn_ballot_w64((uint32_t)l < cnt && gl != gprev);
    g = (uint32_t)__builtin_amdgcn_readlane((int)gl, (int)(cnt - 1u));
    constexpr uint32_t U = 8u;
    for (uint32_t u0 = 0u; u0 < cnt; u0 += U) {
      f32x4 w4[U];
#pragma unroll
      for (uint32_t u = 0u; u < U; ++u) w4[u] = ldg4(Fl + (uint32_t)__builtin_amdgcn_readlane((int)rowo, (int)(u0 + u)));
      const uint32_t ob = (uint32_t)(opens >> u0) & 0xffu;
      if (ob == 0u) {  // the usual batch: all eight entries go on with the open partial sum
#pragma unroll
        for (uint32_t u = 0u; u < U; ++u) {
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), (int)(u0 + u)));
          const f32x2 vv = {v, v};
          c01 = __builtin_elementwise_fma(vv, f32x2{w4[u].x, w4[u].y}, c01);
          c23 = __builtin_elementwise_fma(vv, f32x2{w4[u].z, w4[u].w}, c23);
        }
      } else {
#pragma unroll
        for (uint32_t u = 0u; u < U; ++u) {
          if ((ob >> u) & 1u) close();
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), (int)(u0 + u)));
          const f32x2 vv = {v, v};
          c01 = __builtin_elementwise_fma(vv, f32x2{w4[u].x, w4[u].y}, c01);
          c23 = __builtin_elementwise_fma(vv, f32x2{w4[u].z, w4[u].w}, c23);
        }
      }
    }
  }
  if (l < HID / 4) {
    f32x4 o;
    o.x = (float)(tot[0] + (double)c01.x), o.y = (float)(tot[1] + (double)c01.y);
    o.z = (float)(tot[2] + (double)c23.x), o.w = (float)(tot[3] + (double)c23.y);
    *reinterpret_cast<f32x4 *>(feat + (size_t)b * HID + 4 * l) = o;
  }
}

// dense form: one 16-wave workgroup per agent, wavefront w scans channels 2 w and 2 w + 1 of the observation (its partial
// sum), the sixteen partial sums meet in LDS.  With li.counts given (the launch behind k_feat_list) only the agents whose
// list did not fit are done.
constexpr int FD_T = 64 * FD_G;
__global__ __launch_bounds__(FD_T) void k_feat_dense(const float *__restrict__ obs, const float *__restrict__ F, float *__restrict__ feat,
                                                     int agents, C0List li) {
  __shared__ float part[FD_G][HID];
  __shared__ uint32_t some[FD_G];
  const int t = (int)threadIdx.x, w = t >> 6, l = t & 63;
  auto nxt = [&](int b) {
    if (li.counts)
      while (b < agents && !(li.counts[b] > (uint32_t)C0_LCAP || li.counts[b] > (uint32_t)li.cap)) b += (int)gridDim.x;
    return b;
  };
  if (li.counts) {
    // the launch behind k_feat_list is idle when every list fitted: each thread looks at one of this workgroup's agents
    // (one load each, all in flight together — walking them with nxt() is a chain of dependent loads: 7 us of a 230 us loop)
    bool mine = false;
    for (int b = (int)blockIdx.x + t * (int)gridDim.x; b < agents; b += FD_T * (int)gridDim.x)
      mine = mine || li.counts[b] > (uint32_t)C0_LCAP || li.counts[b] > (uint32_t)li.cap;
    if (!__syncthreads_or(mine)) return;
  }
  constexpr int NIT = (FD_SEG + 63) / 64;  // 31 floats per lane
  const float *Fl = F + 4 * (l < HID / 4 ? l : 0);
  for (int b = nxt((int)blockIdx.x); b < agents; b = nxt(b + (int)gridDim.x)) {
    const float *src = obs + (size_t)b * OBS_F + (size_t)w * FD_SEG;
    float pre[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = k * 64 + l;
      pre[k] = i < FD_SEG ? __builtin_nontemporal_load(src + i) : 0.f;
    }
    float cur[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t any = 0u;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      uint64_t m = __builtin_amdgcn_ballot_w64(pre[k] != 0.f);
      if (!m) continue;
      any = 1u;
      const uint32_t idx = (uint32_t)(w * FD_SEG + k * 64 + l);  // = channel * 961 + y * 31 + x
      const uint32_t chn = idx / (uint32_t)(OBS_W * OBS_W), r = idx - chn * (uint32_t)(OBS_W * OBS_W);
      const uint32_t y = r / (uint32_t)OBS_W, x = r - y * (uint32_t)OBS_W;
      const uint32_t rowo = ((x << 10) | (y << 5) | chn) * (uint32_t)HID;
      while (m) {  // four non-zeros at a time: their rows of F first, then the sums in scan order
        f32x4 w4[4];
        float vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w4[q] = f32x4{0.f, 0.f, 0.f, 0.f}, vv[q] = 0.f;
          if (m) {
            const int s = __builtin_ctzll(m);
            m &= m - 1ull;
            vv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pre[k]), s));
            w4[q] = ldg4(Fl + (uint32_t)__builtin_amdgcn_readlane((int)rowo, s));
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // (an unused slot: value 0 on a zero row, which leaves the sums as they are)
          cur[0] = fmaf(vv[q], w4[q].x, cur[0]);
          cur[1] = fmaf(vv[q], w4[q].y, cur[1]);
          cur[2] = fmaf(vv[q], w4[q].z, cur[2]);
          cur[3] = fmaf(vv[q], w4[q].w, cur[3]);
        }
      }
    }
    if (l < HID / 4) *reinterpret_cast<f32x4 *>(&part[w][4 * l]) = f32x4{cur[0], cur[1], cur[2], cur[3]};
    if (l == 0) some[w] = any;
    __syncthreads();
    if (t < HID) {
      double tot = 0.0;
      for (int g = 0; g < FD_G; ++g)
        if (some[g]) tot += (double)part[g][t];
      feat[(size_t)b * HID + t] = (float)tot;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Row kernels: one wavefront per agent, lane l owns elements l, l+64, l+128 (< 160) of a 160-vector.
// ---------------------------------------------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
struct Row3 {
  float v[3];
};
__device__ inline Row3 row_load(const float *p, int l) {
  Row3 r;
  r.v[0] = p[l], r.v[1] = p[l + 64], r.v[2] = (l < HID - 128) ? p[l + 128] : 0.f;
  return r;
}
__device__ inline void row_store(float *p, int l, const Row3 &r) {
  p[l] = r.v[0], p[l + 64] = r.v[1];
  if (l < HID - 128) p[l + 128] = r.v[2];
}
// x * 160 / (sum|x| + 1e-8)   Modules.hpp:43,46,108,112,126,130
__device__ inline Row3 row_norm(const Row3 &x) {
  const float s = wave_sum(fabsf(x.v[0]) + fabsf(x.v[1]) + fabsf(x.v[2])) + 1e-8f;
  Row3 y;
#pragma unroll
  for (int i = 0; i < 3; ++i) y.v[i] = x.v[i] * (float)HID / s;
  return y;
}
__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// torch GRU cell, gate order r,z,n; gi = W_ih x + b_ih, gh = W_hh h + b_hh (both from k_gemm):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
__device__ inline Row3 gru_cell(const float *gi, const float *gh, const Row3 &h, int l) {
  Row3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = l + 64 * i;
    if (e < HID) {
      const float r = sigmoidf_(gi[e] + gh[e]);
      const float z = sigmoidf_(gi[HID + e] + gh[HID + e]);
      const float n = tanhf(gi[2 * HID + e] + r * gh[2 * HID + e]);
      o.v[i] = (1.f - z) * n + z * h.v[i];
    } else {
      o.v[i] = 0.f;
    }
  }
  return o;
}

#define SFP_ROW_PROLOGUE            \
  const int l = threadIdx.x & 63;   \
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6); \
  if (a >= agents) return;

__global__ __launch_bounds__(256) void k_norm(const float *x, float *y, float *y2, int agents) {  // y2: optional copy
  SFP_ROW_PROLOGUE
  const Row3 r = row_norm(row_load(x + (size_t)a * HID, l));
  row_store(y + (size_t)a * HID, l, r);
  if (y2) row_store(y2 + (size_t)a * HID, l, r);
}

// gru0 + the assembly of `combined` (Modules.hpp:110-123): comb[0:160] = norm(h0') + feat_n,
// comb[160:329] = norm(pov), comb[329:352] = 0 (K padding)
__global__ __launch_bounds__(256) void k_gru0(const float *gi, const float *gh, float *h, const float *feat_n,
                                              const float *obs, const float *action_input, float *comb, int agents) {
  SFP_ROW_PROLOGUE
  float *hp = h + (size_t)a * HID;
  const Row3 hn = gru_cell(gi + (size_t)a * G3, gh + (size_t)a * G3, row_load(hp, l), l);
  row_store(hp, l, hn);
  const Row3 on = row_norm(hn), f = row_load(feat_n + (size_t)a * HID, l);
  Row3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i) c.v[i] = on.v[i] + f.v[i];
  float *cp = comb + (size_t)a * COMB_PAD;
  row_store(cp, l, c);
  // pov: cells (-1,0) (0,-1) (0,0) (0,1) (1,0) around the centre, 32 channels each, then the action one-hot
  const float *op = obs + (size_t)a * OBS_F;
  float pv[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = l + 64 * i;
    float v = 0.f;
    if (e < 5 * OBS_C) {
      const int cell = e >> 5, ch = e & 31;
      const int dy = (cell == 0) ? -1 : (cell == 4) ? 1 : 0;
      const int dx = (cell == 1) ? -1 : (cell == 3) ? 1 : 0;
      v = op[(size_t)ch * OBS_W * OBS_W + (OBS_W / 2 + dy) * OBS_W + (OBS_W / 2 + dx)];
    } else if (e < POV) {
      v = action_input[(size_t)a * ACT + (e - 5 * OBS_C)];
    }
    pv[i] = v;
    s += fabsf(v);
  }
  s = wave_sum(s) + 1e-8f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = l + 64 * i;
    if (e < COMB_PAD - HID) cp[HID + e] = (e < POV) ? pv[i] * (float)HID / s : 0.f;
  }
}

// gru1 + residual (Modules.hpp:128-131): out = norm(h1') + gated_n
__global__ __launch_bounds__(256) void k_gru1(const float *gi, const float *gh, float *h, const float *gated_n,
                                              float *out, int agents) {
  SFP_ROW_PROLOGUE
  float *hp = h + (size_t)a * HID;
  const Row3 hn = gru_cell(gi + (size_t)a * G3, gh + (size_t)a * G3, row_load(hp, l), l);
  row_store(hp, l, hn);
  const Row3 on = row_norm(hn), gn = row_load(gated_n + (size_t)a * HID, l);
  Row3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i) o.v[i] = on.v[i] + gn.v[i];
  row_store(out + (size_t)a * HID, l, o);
}

// one ResB layer after its Linear (Modules.hpp:45-46): x <- norm(relu(lin) + x); blockIdx.y picks the head
__global__ __launch_bounds__(256) void k_res(const float *lin0, float *x0, const float *lin1, float *x1, int agents) {
  SFP_ROW_PROLOGUE
  const float *lin = blockIdx.y ? lin1 : lin0;
  float *x = blockIdx.y ? x1 : x0;
  float *xp = x + (size_t)a * HID;
  const Row3 y = row_load(lin + (size_t)a * HID, l), xv = row_load(xp, l);
  Row3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.v[i] = fmaxf(y.v[i], 0.f) + xv.v[i];
  row_store(xp, l, row_norm(r));
}

// the two output layers (Modules.hpp:172-175): p = softmax(W_p x_p + b_p) + 1e-8, v = sigmoid(W_v x_v + b_v)
__global__ __launch_bounds__(256) void k_heads(const float *xp, const float *xv, const float *wp, const float *bp,
                                               const float *wv, const float *bv, float *probs, float *value,
                                               int agents) {
  SFP_ROW_PROLOGUE
  const Row3 p = row_load(xp + (size_t)a * HID, l), v = row_load(xv + (size_t)a * HID, l);
  float logit[ACT];
#pragma unroll
  for (int k = 0; k < ACT; ++k) {
    const Row3 wr = row_load(wp + k * HID, l);
    logit[k] = wave_sum(p.v[0] * wr.v[0] + p.v[1] * wr.v[1] + p.v[2] * wr.v[2]) + bp[k];
  }
  const Row3 wr = row_load(wv, l);
  const float val = wave_sum(v.v[0] * wr.v[0] + v.v[1] * wr.v[1] + v.v[2] * wr.v[2]) + bv[0];
  float mx = logit[0];
#pragma unroll
  for (int k = 1; k < ACT; ++k) mx = fmaxf(mx, logit[k]);
  float e[ACT], s = 0.f;
#pragma unroll
  for (int k = 0; k < ACT; ++k) e[k] = expf(logit[k] - mx), s += e[k];
  if (l < ACT) {
    float mine = e[0];
#pragma unroll
    for (int k = 1; k < ACT; ++k) mine = (l == k) ? e[k] : mine;
    probs[(size_t)a * ACT + l] = mine / s + 1e-8f;
  }
  if (l == 0) value[a] = sigmoidf_(val);
}

// ---------------------------------------------------------------------------------------------------------
// k_tail: everything behind conv2 in one launch (conv3, both GRU cells, combined_processor, the two heads: 7 matrix
// and 9 row launches before).  One 16-wave workgroup per 16 agents; the activations of those agents stay in LDS from
// layer to layer, the weights (3 MB in all) stream from L2 once per workgroup straight into MFMA operands.
//   matrix steps  v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: the same fmaf-chain arithmetic as k_gemm): a wave
//                 owns 16-column tiles of the [16 agents][N] output; per 16 k it reads one float4 of its agent row
//                 from LDS and one float4 of its weight row from global memory (lane l: row / column l & 15,
//                 k = 4 (l >> 4) + j in MFMA j — any fixed permutation of k works as long as both operands use it),
//                 the weights of the wave's NEXT tile on their way while this one is multiplied (ts_tile160)
//   row steps     one wave per agent, the row kernels' own functions (row_norm, gru_cell) on LDS rows
// LDS rows that feed a matrix step are K + 8 floats apart: conflict-free for the ds_read_b128 lane groups.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int TL_R = 16, TL_T = 1024;
constexpr int TL_LD = HID + 8, TL_LDC = COMB_PAD + 8, TL_LDX = 9 * HID + 8;  // 168, 360, 1448
constexpr int TL_A = 0;                              // region A: conv3's input rows, later gi / gh / comb, later lin0 / lin1 / x0 / x1
constexpr int TL_GI = TL_A, TL_GH = TL_A + TL_R * G3, TL_COMB = TL_A + 2 * TL_R * G3;
constexpr int TL_LIN0 = TL_A, TL_LIN1 = TL_A + TL_R * TL_LD, TL_X0 = TL_COMB, TL_X1 = TL_COMB + TL_R * TL_LD;
constexpr int TL_B0 = TL_A + TL_R * TL_LDX;          // feat_n
constexpr int TL_B1 = TL_B0 + TL_R * TL_LD;          // h0, then h1
constexpr int TL_B2 = TL_B1 + TL_R * TL_LD;          // gated_n
constexpr int TL_Y0 = TL_B2 + TL_R * TL_LD;          // feat, then gated
constexpr int TL_PV = TL_Y0 + TL_R * TL_LD;          // raw pov values and h1, fetched at the start
constexpr int TL_LDP = 192;                          // a pov row: 169 values, three per lane
constexpr int TL_H1 = TL_PV + TL_R * TL_LDP;
constexpr int TL_FLOATS = TL_H1 + TL_R * TL_LD;
constexpr int TL_LDS = TL_FLOATS * 4;                // 158 720 bytes
static_assert(TL_COMB + TL_R * TL_LDC <= TL_B0 && TL_X1 + TL_R * TL_LD <= TL_B0, "region A holds its tenants");

__device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct ActStr {
  char c[ACT];
};

// The tail of Agent::predict() (Agent.hpp:200-216) for agent a: v[0] = 0.5, the rest scaled to 0.5 in all, one draw from
// discrete_distribution(v) — or the arg-max.  One body for k_act and for k_tail's last lines (sf_policy_predict_sparse).
__device__ inline int act_pick(float (&v)[ACT], uint64_t seed, uint64_t draw, int greedy, int a) {
  const float sc = 0.5f / (1.f - v[0] + 1e-5f);
#pragma unroll
  for (int k = 1; k < ACT; ++k) v[k] *= sc;
  v[0] = 0.5f;
  int pick = 0;
  if (greedy) {
#pragma unroll
    for (int k = 1; k < ACT; ++k)
      if (v[k] > v[pick]) pick = k;
  } else {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < ACT; ++k) tot += v[k];
    const uint64_t r = mix64(mix64(seed ^ mix64((uint64_t)a)) + draw);
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f) * tot;  // [0, tot)
    float c = 0.f;
    bool found = false;
    pick = ACT - 1;
#pragma unroll
    for (int k = 0; k < ACT - 1; ++k) {
      c += v[k];
      if (!found && u < c) pick = k, found = true;
    }
  }
  return pick;
}

struct TailArgs {
  const float *act2, *obs, *pov, *conv3_w;  // pov: the 160 centre values as a dense row per agent, or null (gather them from obs)
  const float *feat;                        // the folded convolution stack's output (k_feat_*): conv3 is then not run here
  const float *gru_w_ih[2], *gru_w_hh[2], *gru_b_ih[2], *gru_b_hh[2];
  const float *comb_w, *comb_b;
  const float *res_w[2][3], *res_b[2][3], *head_w[2], *head_b[2];
  float *h[2];
  float *action_input;  // (read at the start; written by the folded sf_policy_act)
  float *probs, *value;
  int agents;
  // sf_policy_predict_sparse: the calls around the forward folded into it
  //   before: sf_policy_reset_memory — an agent whose mask byte, or whose arena's word, is non-zero starts from h = 0 and
  //           the "no action" one-hot instead of what is stored
  //   after:  sf_policy_act — the draw, the one-hot for the next call, the command char (act != 0)
  const uint8_t *reset_mask;   // [agents] or null
  const int32_t *reset_words;  // word (a / reset_group) * reset_stride, or null (sf_done_view_device)
  int reset_stride, reset_group;
  int act, greedy;
  ActStr as;
  uint64_t seed, draw;
  uint8_t *cmd;
  int32_t *action;
};

// out[r][n0 + c] = bias[n0 + c] + sum_k in[r][k] * W[n0 + c][k] for the 16 agents r and 16 columns c of one tile
template <int K>
__device__ inline void tail_tile(const float *in, int ldi, const float *W, const float *bias, float *out, int ldo, int n0, int l) {
  constexpr int STEPS = K / 16, U = 8, NBATCH = (STEPS + U - 1) / U;  // weight loads in batches of 8 float4, two batches in flight
  const int row = l & 15, g = l >> 4;
  const float *ap = in + row * ldi + 4 * g;
  const float *wp = W + (size_t)(n0 + row) * K + 4 * g;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 wq[2][U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < STEPS) wq[0][u] = ldg4(wp + 16 * u);
#pragma unroll
  for (int b = 0; b < NBATCH; ++b) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((b + 1) * U + u < STEPS) wq[(b + 1) & 1][u] = ldg4(wp + 16 * ((b + 1) * U + u));
    __builtin_amdgcn_sched_barrier(0);  // (the compiler would sink every load to just in front of its MFMAs: 2 in flight)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (b * U + u >= STEPS) continue;
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ap + 16 * (b * U + u));
      const f32x4 w4 = wq[b & 1][u];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w4.w, acc, 0, 0, 0);
    }
  }
  const float bv = bias ? bias[n0 + row] : 0.f;  // lane l holds column n0 + (l & 15) of agents 4 g .. 4 g + 3
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * ldo + n0 + row] = acc[r] + bv;
}

// ---- k_tail's weight stream ----------------------------------------------------------------------------------
// A wave's tiles follow each other — inside a layer and from layer to layer — and the weights do not depend on anything
// the kernel computes.  They are fetched in batches of five k-steps (of 16: a K = 160 tile is two batches) into two
// register buffers, always one batch ahead of the MFMAs: the second batch of a tile while its first is multiplied, the
// first batch of the wave's NEXT tile — of this layer or, across the barriers and the row steps in between, of the next
// one — while its second is.  What stays exposed is the very first batch of the kernel.  Same products in the same
// order as tail_tile: the results are the same bits.
constexpr int TS_U = 5;
struct TsBuf {  // (passed and returned by value: every element stays a register)
  f32x4 v[TS_U];
};
// The streamed weights are stored in the order the loads take them (upload_tiles(), sf_policy_create): tile (16 output
// columns) by tile, k-step (16 inputs) by k-step, lane by lane — one k-step of a tile is 1 KB that a wave's
// global_load_dwordx4 reads as eight whole 128-byte lines.  (From the row-major matrix the same load touched sixteen
// lines, half of each, and the vector L1's tag pipe — not the L2, not the matrix pipe — set the pace of the tile phases:
// in-kernel stamps, round 4.)
__device__ inline const float *ts_wp(const float *W, int K, int n0, int l) { return W + (size_t)(n0 >> 4) * ((size_t)K * 16) + 4 * l; }
template <int N>
__device__ inline TsBuf ts_issue(TsBuf b, const float *wp, int s0) {
#pragma unroll
  for (int u = 0; u < N; ++u) b.v[u] = ldg4(wp + 256 * (s0 + u));
  return b;
}
template <int N>
__device__ inline void ts_mma(f32x4v &acc, const TsBuf &b, const float *ap, int s0) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ap + 16 * (s0 + u));
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b.v[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b.v[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b.v[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b.v[u].w, acc, 0, 0, 0);
  }
}
__device__ inline void ts_store(const f32x4v &acc, const float *bias, float *out, int ldo, int n0, int l) {
  const int row = l & 15, g = l >> 4;
  const float bv = bias ? bias[n0 + row] : 0.f;  // lane l holds column n0 + (l & 15) of agents 4 g .. 4 g + 3
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * ldo + n0 + row] = acc[r] + bv;
}
// one K = 160 tile: its first batch is already on its way in b0; `next(b0)` requests the wave's next tile's first batch
template <class Next>
__device__ inline void ts_tile160(const float *in, int ldi, const float *wp, const float *bias, float *out, int ldo, int n0, int l,
                                  TsBuf &b0, TsBuf &b1, Next next) {
  const float *ap = in + (l & 15) * ldi + 4 * (l >> 4);
  b1 = ts_issue<5>(b1, wp, 5);
  __builtin_amdgcn_sched_barrier(0);  // (the compiler would sink the loads to just in front of their MFMAs)
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  ts_mma<5>(acc, b0, ap, 0);
  b0 = next(b0);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<5>(acc, b1, ap, 5);
  ts_store(acc, bias, out, ldo, n0, l);
}
// the K = 352 tile of combined_processor: 22 k-steps as batches of 4 4 4 4 3 3 (an even number of batches: the next tile
// starts in b0 again); its first batch (four k-steps) is already on its way in b0
template <class Next>
__device__ inline void ts_tile352(const float *in, int ldi, const float *wp, const float *bias, float *out, int ldo, int n0, int l,
                                  TsBuf &b0, TsBuf &b1, Next next) {
  static_assert(COMB_PAD == 16 * 22, "combined_processor's padded K");
  const float *ap = in + (l & 15) * ldi + 4 * (l >> 4);
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  b1 = ts_issue<4>(b1, wp, 4);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<4>(acc, b0, ap, 0);
  b0 = ts_issue<4>(b0, wp, 8);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<4>(acc, b1, ap, 4);
  b1 = ts_issue<4>(b1, wp, 12);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<4>(acc, b0, ap, 8);
  b0 = ts_issue<3>(b0, wp, 16);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<4>(acc, b1, ap, 12);
  b1 = ts_issue<3>(b1, wp, 19);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<3>(acc, b0, ap, 16);
  b0 = next(b0);
  __builtin_amdgcn_sched_barrier(0);
  ts_mma<3>(acc, b1, ap, 19);
  ts_store(acc, bias, out, ldo, n0, l);
}

// k_tail's waves hand data to each other through LDS only: its barriers order LDS traffic and leave global loads and
// stores (weights on their way, recurrent state on its way out) in flight
__device__ __forceinline__ void tail_barrier() {
#ifdef SF_TAIL_FULL_BARRIER
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

#ifdef SF_DIAG_TAIL  // diagnostic build only (tools/r04_tail_stamps.py): cycles per phase of k_tail, per wave of the last launch
__device__ uint32_t sf_diag_tail[4096 * 16 * 24];  // [workgroup][wave][phase]
#define TL_STAMP(ph)                                                                                         \
  do {                                                                                                       \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096u)                                                       \
      sf_diag_tail[(blockIdx.x * 16u + (threadIdx.x >> 6)) * 24u + (ph)] = (uint32_t)(t_ - tl_last_);        \
    tl_last_ = t_;                                                                                           \
  } while (0)
#else
#define TL_STAMP(ph)
#endif
__global__ __launch_bounds__(TL_T) void k_tail(TailArgs t) {
  extern __shared__ __attribute__((aligned(16))) float tl[];
#ifdef SF_DIAG_TAIL
  unsigned long long tl_last_ = __builtin_amdgcn_s_memtime();
#endif
  // (readfirstlane: the wave index is uniform, and the compiler should know — tile choices become scalar branches)
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = threadIdx.x & 63;
  const int a_raw = blockIdx.x * TL_R + w;
  const bool valid = a_raw < t.agents;
  const int a = valid ? a_raw : t.agents - 1;  // a ragged last workgroup computes its missing rows on the last agent, stores nothing
  // the weight stream (see ts_tile160): where each of this wave's tiles lives
  TsBuf b0 = {}, b1 = {};
  auto gru_wp = [&](int g, int tt) {
    const int hh = tt >= G3 / 16, n0 = 16 * (tt - hh * (G3 / 16));
    return ts_wp(hh ? t.gru_w_hh[g] : t.gru_w_ih[g], HID, n0, l);
  };
  auto res_wp = [&](int i, int tt) {
    const int hd = tt >= HID / 16, n0 = 16 * (tt - hd * (HID / 16));
    return ts_wp(t.res_w[hd][i], HID, n0, l);
  };
  // ---- the prologue's global loads: the restart flags, h0, h1, the agent's pov (5 cells x 32 channels around the centre of
  // the observation, then the action one-hot), its feature row, the first weights.  All of them are issued before the
  // first is waited for: every load is unconditional, from an address that is valid whatever the options (written behind
  // uniform branches the compiler kept them in program order, a wait after each — eight round trips, 7 k cycles of the
  // kernel's start).
  const uint8_t *fmp = t.reset_mask ? t.reset_mask + a : reinterpret_cast<const uint8_t *>(t.h[0]);
  const int32_t *fwp = t.reset_words ? t.reset_words + (size_t)(a / t.reset_group) * (size_t)t.reset_stride : reinterpret_cast<const int32_t *>(t.h[0]);
  const uint8_t fm = *fmp;
  const int32_t fw = *fwp;
  Row3 h0 = row_load(t.h[0] + (size_t)a * HID, l), h1 = row_load(t.h[1] + (size_t)a * HID, l);
  float pvv[3];
  {
    const float *op = t.obs + (size_t)a * OBS_F;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = l + 64 * i;
      const float *src = t.action_input + (size_t)a * ACT;  // (lanes past the row: any valid address, the value is dropped)
      if (e < 5 * OBS_C) {
        const int cell = e >> 5, ch = e & 31;
        const int dy = (cell == 0) ? -1 : (cell == 4) ? 1 : 0;
        const int dx = (cell == 1) ? -1 : (cell == 3) ? 1 : 0;
        src = t.pov ? t.pov + (size_t)a * (5 * OBS_C) + e : op + (size_t)ch * OBS_W * OBS_W + (OBS_W / 2 + dy) * OBS_W + (OBS_W / 2 + dx);
      } else if (e < POV) {
        src += e - 5 * OBS_C;
      }
      pvv[i] = *src;
    }
  }
  const Row3 fr = row_load((t.feat ? t.feat : t.h[0]) + (size_t)a * HID, l);
  if (t.feat) b0 = ts_issue<5>(b0, gru_wp(0, w), 0);  // folded form: gru0's first tile is this wave's first
  // a restarted game's agent is a new Agent (gameplay.hpp:481): zero memory, "no action" as its last action
  const bool fresh = __builtin_amdgcn_readfirstlane((int)((t.reset_mask && fm != 0) || (t.reset_words && fw != 0))) != 0;  // (uniform over the wave)
  if (fresh) h0 = Row3{}, h1 = Row3{};
  if (!t.feat) {  // conv3's input (act2 row = 9 pixels x 160 channels, the K order of the permuted weight)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(t.act2 + (size_t)a * (9 * HID));
    f32x4 *dst = reinterpret_cast<f32x4 *>(tl + TL_A + w * TL_LDX);
    for (int i = l; i < 9 * HID / 4; i += 64) dst[i] = src[i];
  }
  row_store(tl + TL_B1 + w * TL_LD, l, h0);
  row_store(tl + TL_H1 + w * TL_LD, l, h1);  // (used after gru0)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = l + 64 * i;
    float v = 0.f;
    if (e < 5 * OBS_C) v = pvv[i];
    else if (e < POV) v = fresh ? (e == 5 * OBS_C ? 1.f : 0.f) : pvv[i];
    tl[TL_PV + w * TL_LDP + e] = v;
  }
  if (t.feat) {  // feat_n is a row of the wave's own agent                                                     :108
    row_store(tl + TL_B0 + w * TL_LD, l, row_norm(fr));
    TL_STAMP(0);
    TL_STAMP(1);
    TL_STAMP(2);
  } else {
    TL_STAMP(0);
    tail_barrier();
    TL_STAMP(1);
    if (w < HID / 16) tail_tile<9 * HID>(tl + TL_A, TL_LDX, t.conv3_w, nullptr, tl + TL_Y0, TL_LD, 16 * w, l);  // Modules.hpp:66-71
    b0 = ts_issue<5>(b0, gru_wp(0, w), 0);
    tail_barrier();
    row_store(tl + TL_B0 + w * TL_LD, l, row_norm(row_load(tl + TL_Y0 + w * TL_LD, l)));  // feat_n :108
    TL_STAMP(2);
  }
  tail_barrier();
  TL_STAMP(3);
  // ---- gru0: gi = W_ih feat_n + b_ih, gh = W_hh h0 + b_hh (30 + 30 tiles)                          :110-113
  for (int tt = w; tt < 2 * (G3 / 16); tt += TL_R) {
    const int hh = tt >= G3 / 16, n0 = 16 * (tt - hh * (G3 / 16));
    ts_tile160(tl + (hh ? TL_B1 : TL_B0), TL_LD, gru_wp(0, tt), hh ? t.gru_b_hh[0] : t.gru_b_ih[0], tl + (hh ? TL_GH : TL_GI), G3, n0, l,
               b0, b1, [&](TsBuf b) {
                 if (tt + TL_R < 2 * (G3 / 16)) return ts_issue<5>(b, gru_wp(0, tt + TL_R), 0);
                 if (w < HID / 16) return ts_issue<4>(b, ts_wp(t.comb_w, COMB_PAD, 16 * w, l), 0);  // next: combined_processor
                 return ts_issue<5>(b, gru_wp(1, w), 0);                                            // (no tile there: gru1)
               });
  }
  TL_STAMP(4);
  tail_barrier();
  TL_STAMP(5);
  {  // gru cell, combined = [norm(h0') + feat_n | norm(pov) | 0]                                        :110-123
    const Row3 hn = gru_cell(tl + TL_GI + w * G3, tl + TL_GH + w * G3, row_load(tl + TL_B1 + w * TL_LD, l), l);
    if (valid) row_store(t.h[0] + (size_t)a * HID, l, hn);
    const Row3 on = row_norm(hn), f = row_load(tl + TL_B0 + w * TL_LD, l);
    Row3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i) c.v[i] = on.v[i] + f.v[i];
    float *cp = tl + TL_COMB + w * TL_LDC;
    row_store(cp, l, c);
    float pv[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = l + 64 * i;
      pv[i] = tl[TL_PV + w * TL_LDP + e];
      s += fabsf(pv[i]);
    }
    s = wave_sum(s) + 1e-8f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = l + 64 * i;
      if (e < COMB_PAD - HID) cp[HID + e] = (e < POV) ? pv[i] * (float)HID / s : 0.f;
    }
    row_store(tl + TL_B1 + w * TL_LD, l, row_load(tl + TL_H1 + w * TL_LD, l));  // h1 takes h0's place
  }
  TL_STAMP(6);
  tail_barrier();
  TL_STAMP(7);
  if (w < HID / 16)                                                                                         // :125
    ts_tile352(tl + TL_COMB, TL_LDC, ts_wp(t.comb_w, COMB_PAD, 16 * w, l), t.comb_b, tl + TL_Y0, TL_LD, 16 * w, l, b0, b1,
               [&](TsBuf b) { return ts_issue<5>(b, gru_wp(1, w), 0); });
  TL_STAMP(8);
  tail_barrier();
  TL_STAMP(9);
  row_store(tl + TL_B2 + w * TL_LD, l, row_norm(row_load(tl + TL_Y0 + w * TL_LD, l)));  // gated_n :126
  TL_STAMP(10);
  tail_barrier();
  TL_STAMP(11);
  for (int tt = w; tt < 2 * (G3 / 16); tt += TL_R) {  // gru1                                             :128-131
    const int hh = tt >= G3 / 16, n0 = 16 * (tt - hh * (G3 / 16));
    ts_tile160(tl + (hh ? TL_B1 : TL_B2), TL_LD, gru_wp(1, tt), hh ? t.gru_b_hh[1] : t.gru_b_ih[1], tl + (hh ? TL_GH : TL_GI), G3, n0, l,
               b0, b1, [&](TsBuf b) {
                 if (tt + TL_R < 2 * (G3 / 16)) return ts_issue<5>(b, gru_wp(1, tt + TL_R), 0);
                 return ts_issue<5>(b, res_wp(0, w), 0);  // next: the first ResB layer
               });
  }
  TL_STAMP(12);
  tail_barrier();
  TL_STAMP(13);
  {  // out = norm(h1') + gated_n; both heads start from norm(out)
    const Row3 hn = gru_cell(tl + TL_GI + w * G3, tl + TL_GH + w * G3, row_load(tl + TL_B1 + w * TL_LD, l), l);
    if (valid) row_store(t.h[1] + (size_t)a * HID, l, hn);
    const Row3 on = row_norm(hn), gn = row_load(tl + TL_B2 + w * TL_LD, l);
    Row3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i) o.v[i] = on.v[i] + gn.v[i];
    const Row3 xn = row_norm(o);
    row_store(tl + TL_X0 + w * TL_LD, l, xn);
    row_store(tl + TL_X1 + w * TL_LD, l, xn);
  }
  TL_STAMP(14);
  tail_barrier();
  TL_STAMP(15);
  for (int i = 0; i < 3; ++i) {  // ResB layers of the two heads                                           :41-48
    for (int tt = w; tt < 2 * (HID / 16); tt += TL_R) {
      const int hd = tt >= HID / 16, n0 = 16 * (tt - hd * (HID / 16));
      ts_tile160(tl + (hd ? TL_X1 : TL_X0), TL_LD, res_wp(i, tt), t.res_b[hd][i], tl + (hd ? TL_LIN1 : TL_LIN0), TL_LD, n0, l, b0, b1,
                 [&](TsBuf b) {
                   if (tt + TL_R < 2 * (HID / 16)) return ts_issue<5>(b, res_wp(i, tt + TL_R), 0);
                   if (i < 2) return ts_issue<5>(b, res_wp(i + 1, w), 0);
                   if (w < 2) return ts_issue<5>(b, ts_wp(t.head_w[w], HID, 0, l), 0);  // next: the output layers
                   return b;
                 });
    }
    TL_STAMP(16);
    tail_barrier();
    TL_STAMP(17);
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      float *xp = tl + (hd ? TL_X1 : TL_X0) + w * TL_LD;
      const Row3 y = row_load(tl + (hd ? TL_LIN1 : TL_LIN0) + w * TL_LD, l), xv = row_load(xp, l);
      Row3 r;
#pragma unroll
      for (int q = 0; q < 3; ++q) r.v[q] = fmaxf(y.v[q], 0.f) + xv.v[q];
      row_store(xp, l, row_norm(r));
    }
    TL_STAMP(18);
    tail_barrier();
    TL_STAMP(19);
  }
  // p = softmax(W_p x_p + b_p) + 1e-8, v = sigmoid(W_v x_v + b_v)                                           :172-175
  // (the two output layers as one MFMA tile each: weights padded with zero rows to 16 columns)
  if (w < 2)
    ts_tile160(tl + (w ? TL_X1 : TL_X0), TL_LD, ts_wp(t.head_w[w], HID, 0, l), t.head_b[w], tl + (w ? TL_LIN1 : TL_LIN0), TL_LD, 0, l, b0, b1,
               [&](TsBuf b) { return b; });
  TL_STAMP(20);
  tail_barrier();
  TL_STAMP(21);
  if (valid) {
    const float *lg = tl + TL_LIN0 + w * TL_LD;
    float mx = lg[0];
#pragma unroll
    for (int k = 1; k < ACT; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f, mine = 0.f;
#pragma unroll
    for (int k = 0; k < ACT; ++k) {
      const float e = expf(lg[k] - mx);
      s += e;
      mine = (l == k) ? e : mine;
    }
    const float pl = mine / s + 1e-8f;
    if (l < ACT) t.probs[(size_t)a * ACT + l] = pl;
    if (l == 0) t.value[a] = sigmoidf_(tl[TL_LIN1 + w * TL_LD]);
    if (t.act) {  // sf_policy_act on the probabilities just stored (lanes 0..8 hold them)
      float v[ACT];
#pragma unroll
      for (int k = 0; k < ACT; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), k));
      const int pick = act_pick(v, t.seed, t.draw, t.greedy, a);
      if (l < ACT) t.action_input[(size_t)a * ACT + l] = (l == pick) ? 1.f : 0.f;
      if (l == 0) {
        char c = t.as.c[0];
#pragma unroll
        for (int k = 1; k < ACT; ++k) c = (pick == k) ? t.as.c[k] : c;  // (a chain of selects: no indexed copy of the argument)
        t.cmd[a] = (uint8_t)c;
        if (t.action) t.action[a] = pick;
      }
    }
  }
  TL_STAMP(22);
#ifdef SF_DIAG_TAIL
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096u)  // HW_ID: which SIMD / CU this wave ran on
    sf_diag_tail[(blockIdx.x * 16u + (threadIdx.x >> 6)) * 24u + 23u] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
}

__global__ void k_reset_memory(float *h0, float *h1, float *action_input, const uint8_t *mask, int agents) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = i / HID, e = i - a * HID;
  if (a >= agents || (mask && !mask[a])) return;
  h0[i] = 0.f, h1[i] = 0.f;
  if (e < ACT) action_input[(size_t)a * ACT + e] = (e == 0) ? 1.f : 0.f;
}

// Agent::predict tail + Agent::update (Agent.hpp:200-222)
__global__ void k_act(const float *probs, float *action_input, ActStr as, uint64_t seed, uint64_t draw, int greedy,
                      uint8_t *cmd, int32_t *action, int agents) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= agents) return;
  float v[ACT];
#pragma unroll
  for (int k = 0; k < ACT; ++k) v[k] = probs[(size_t)a * ACT + k];
  const int pick = act_pick(v, seed, draw, greedy, a);
#pragma unroll
  for (int k = 0; k < ACT; ++k) action_input[(size_t)a * ACT + k] = (k == pick) ? 1.f : 0.f;
  cmd[a] = (uint8_t)as.c[pick];
  if (action) action[a] = pick;
}

// ---------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------
#define SFP_HIP(call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return sf::fail(SF_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// f32 -> bf16, round to nearest even (finite inputs: weights)
static uint16_t bf16_rn(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float x;
  std::memcpy(&x, &u, 4);
  return x;
}
// W [N][K] (N % 160 == 0, K % 16 == 0) -> k_gemm_b3's image: [N / 160][K / 16][160][3 parts][16] bf16
static std::vector<uint16_t> split_weights(const float *W, int N, int K) {
  std::vector<uint16_t> img((size_t)N * K * 3);
  const int KT = K / B3_BK;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float x = W[(size_t)n * K + k];
      const uint16_t hi = bf16_rn(x);
      const float r1 = x - bf16_f32(hi);
      const uint16_t mid = bf16_rn(r1);
      const float r2 = r1 - bf16_f32(mid);
      const uint16_t lo = bf16_rn(r2);
      const size_t rec = (((size_t)(n / BN) * KT + k / B3_BK) * BN + n % BN) * 48 + k % B3_BK;
      img[rec] = hi, img[rec + 16] = mid, img[rec + 32] = lo;
    }
  return img;
}

struct Policy {
  int device = 0, max_agents = 0;
  hipStream_t stream = nullptr;
  uint64_t draws = 0;
  // parameters
  float *conv0_wt = nullptr;  // conv0 weights as [c][ky][kx][n] for k_conv0_sparse
  bool dense_conv0 = false;   // SF_POLICY_DENSE_CONV0=1: the implicit-GEMM conv0 instead (A/B, tests)
  void *conv_w3[4] = {};      // conv1, conv2: the weights split into bf16 hi / mid / lo parts for k_gemm_b3
  bool f32_conv = false;      // SF_POLICY_F32_CONV=1: conv1, conv2 on the f32 matrix pipe instead (A/B, tests)
  uint32_t *d_overflows = nullptr;  // sf_policy_forward_sparse: agents whose list did not fit since the last query
  bool fused_tail = true;     // SF_POLICY_FUSED_TAIL=0: the layers behind conv2 as 16 separate launches instead (A/B, tests)
  float *fold = nullptr;      // F: the four convolutions composed into one matrix (k_fold), [FD_ROWS][160]
  bool folded = true;         // SF_POLICY_LAYERED=1: the four convolutions one after the other instead (cross-check, tests)
  float *conv_w[4] = {}, *gru_w_ih[2] = {}, *gru_w_hh[2] = {}, *gru_b_ih[2] = {}, *gru_b_hh[2] = {};
  float *comb_w = nullptr, *comb_b = nullptr;
  float *res_w[2][3] = {}, *res_b[2][3] = {}, *head_w[2] = {}, *head_b[2] = {};  // [0] policy, [1] value
  float *head_w16[2] = {}, *head_b16[2] = {};  // the same padded with zero rows to one 16-column MFMA tile (k_tail)
  // k_tail's copies of its matrices in the order its weight stream reads them (stage_tiles), ONE block in the order of use
  // (2.09 MB)
  float *tail_w = nullptr;
  std::vector<float> tail_stage;  // (host side, until sf_policy_create has uploaded it)
  size_t gru_w_ih_t[2] = {}, gru_w_hh_t[2] = {}, comb_w_t = 0, res_w_t[2][3] = {}, head_w16_t[2] = {};  // offsets in floats
  // per-agent state and scratch
  float *h[2] = {}, *action_input = nullptr;
  float *act[3] = {};  // NHWC conv outputs 15x15, 7x7, 3x3
  float *feat = nullptr, *feat_n = nullptr, *gi = nullptr, *gh = nullptr, *comb = nullptr, *gated = nullptr,
        *gated_n = nullptr, *out = nullptr, *x[2] = {}, *lin[2] = {};
  std::vector<void *> owned;
  // timing of the GEMM launches
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<char> event_split;  // per event: 0 k_gemm (f32 MFMA), 1 k_gemm_b3 (bf16 split), 2 conv0 on the non-zeros, 3 k_tail
  size_t used_events = 0;
  double flop = 0, flop_split = 0, flop_kind[4] = {0, 0, 0, 0};

  ~Policy() {
    for (void *p : owned) (void)hipFree(p);
    for (auto &ev : events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  }
  int dalloc(float **p, size_t floats) {
    void *d = nullptr;
    if (hipMalloc(&d, floats * sizeof(float)) != hipSuccess) return fail(SF_ERR_MEMORY, "hipMalloc failed (policy)");
    owned.push_back(d);
    *p = (float *)d;
    return SF_OK;
  }
  // k_tail's weight stream order (ts_wp / ts_issue): Wt[tile][k-step][lane][j] = W[16 tile + (lane & 15)][16 k-step + 4 (lane >> 4) + j]
  int stage_tiles(size_t *off, const float *src, int N, int K) {
    if (!src) return fail(SF_ERR_ARG, "sf_policy_weights has a null pointer");
    *off = tail_stage.size();
    tail_stage.resize(*off + (size_t)N * K);
    float *t = tail_stage.data() + *off;
    for (int tile = 0; tile < N / 16; ++tile)
      for (int st = 0; st < K / 16; ++st)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j)
            t[(((size_t)tile * (K / 16) + st) * 64 + l) * 4 + j] = src[(size_t)(16 * tile + (l & 15)) * K + 16 * st + 4 * (l >> 4) + j];
    return SF_OK;
  }
  int upload(float **p, const float *src, size_t floats) {
    if (!src) return fail(SF_ERR_ARG, "sf_policy_weights has a null pointer");
    int rc = dalloc(p, floats);
    if (rc) return rc;
    SFP_HIP(hipMemcpy(*p, src, floats * sizeof(float), hipMemcpyHostToDevice));
    return SF_OK;
  }

  // Resident blocks of the large-M shape: its 76 KB of LDS pins two 4-wave blocks on each CU (set in create())
  int sk_blocks = 512;
  float *part = nullptr;  // [sk_blocks][2][128*160]

  template <int WM, int WN, int BKT, int MODE>
  void launch_t(Gemm g) {
    const int BM = WM * 32, KT = g.K / BKT;
    g.ntiles = (g.M + BM - 1) / BM;
    g.part = part;
    const long units = (long)g.ntiles * KT;
    int G = g.ntiles;  // one run per row tile: the classic decomposition
    if (g.N == BN && !g.A2 && units >= 4L * sk_blocks) G = sk_blocks;  // (both block shapes: conv3 is the small-M case)
    g.unit_base = (int)(units / G), g.unit_rem = (int)(units % G);
    hipLaunchKernelGGL((k_gemm<WM, WN, BKT, MODE>), dim3((unsigned)G, (unsigned)(g.N / BN), g.A2 ? 2u : 1u), dim3(WM * WN * 64), 0, stream, g);
    if (G != g.ntiles)
      hipLaunchKernelGGL((k_gemm_fixup<WM * 32>), dim3((unsigned)(G - 1), (unsigned)(g.N / BN)), dim3(256), 0, stream, g, KT, G);
  }
  // the bf16-split form: one 8-wave block per CU
  template <int MODE>
  void launch_b3(Gemm g) {
    const int KT = g.K / B3_BK;
    g.ntiles = (g.M + B3_BM - 1) / B3_BM;
    g.part = part;
    const long units = (long)g.ntiles * KT;
    const int resident = sk_blocks / 2;  // one block per CU
    int G = g.ntiles;
    if (g.N == BN && units >= 4L * resident) G = resident;
    g.unit_base = (int)(units / G), g.unit_rem = (int)(units % G);
    hipLaunchKernelGGL((k_gemm_b3<MODE>), dim3((unsigned)G, (unsigned)(g.N / BN)), dim3(B3_T), B3_LDS, stream, g);
    if (G != g.ntiles)
      hipLaunchKernelGGL((k_gemm_fixup<B3_BM>), dim3((unsigned)(G - 1), (unsigned)(g.N / BN)), dim3(256), 0, stream, g, KT, G);
  }
  template <int MODE>
  void launch_m(const Gemm &g) {
    if (g.W3 && MODE != MODE_NCHW) launch_b3<MODE == MODE_NCHW ? MODE_NHWC : MODE>(g);
    else if (g.M >= 16384) launch_t<4, 1, 32, MODE>(g);
    else launch_t<1, 5, 32, MODE>(g);
  }
  // timing of one matrix launch: records the start event, returns the end event to record behind the launch
  int time_begin(double fl, bool split, hipEvent_t *e1, int kind = -1) {
    *e1 = nullptr;
    if (!timing) return SF_OK;
    if (used_events == events.size()) {
      hipEvent_t a, b;
      SFP_HIP(hipEventCreate(&a));
      SFP_HIP(hipEventCreate(&b));
      events.emplace_back(a, b);
      event_split.push_back(0);
    }
    const hipEvent_t e0 = events[used_events].first;
    *e1 = events[used_events].second;
    if (kind < 0) kind = split ? 1 : 0;
    event_split[used_events] = (char)kind;
    ++used_events;
    if (kind != 2) (split ? flop_split : flop) += fl;  // (conv0 on the non-zeros is not a matrix launch: sf_policy_kernel_time leaves it out)
    flop_kind[kind] += fl;
    SFP_HIP(hipEventRecord(e0, stream));
    return SF_OK;
  }
  int gemm(const Gemm &g, int mode) {
    if (g.K % 32 || g.N % BN || g.M < 1) return fail(SF_ERR_ARG, "policy gemm: unsupported shape");
    hipEvent_t e1 = nullptr;
    {
      const int rc = time_begin(2.0 * g.M * g.N * g.K * (g.A2 ? 2 : 1), g.W3 && mode != MODE_NCHW, &e1);
      if (rc) return rc;
    }
    switch (mode) {
      case MODE_DENSE: launch_m<MODE_DENSE>(g); break;
      case MODE_NHWC: launch_m<MODE_NHWC>(g); break;
      default: launch_m<MODE_NCHW>(g); break;
    }
    SFP_HIP(hipGetLastError());
    if (e1) SFP_HIP(hipEventRecord(e1, stream));
    return SF_OK;
  }
  int dense(const float *A, int lda, const float *W, const float *bias, float *C, int ldc, int M, int N, int K) {
    Gemm g{A, W, bias, C, M, N, K, lda, ldc, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    return gemm(g, MODE_DENSE);
  }
  // two independent products of one shape in one launch
  int dense2(const float *A, const float *W, const float *bias, float *C, const float *A2, const float *W2,
             const float *bias2, float *C2, int lda, int ldc, int M, int N, int K) {
    Gemm g{A, W, bias, C, M, N, K, lda, ldc, 0, 0, 0, 0, 0, 0, nullptr, A2, W2, bias2, C2};
    return gemm(g, MODE_DENSE);
  }
  int conv(const float *in, const float *W, float *outp, int agents, int S, int Cin, int nchw, const void *W3 = nullptr) {
    const int So = (S - 3) / 2 + 1;
    Gemm g{in, W, nullptr, outp, agents * So * So, HID, Cin * 9, 0, HID, S, Cin, So, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (W3 && g.M >= 16384 && !f32_conv) g.W3 = W3;
    return gemm(g, nchw ? MODE_NCHW : MODE_NHWC);
  }
};

static int check_agents(Policy *p, int agents) {
  if (!p) return fail(SF_ERR_ARG, "null policy");
  if (agents < 1 || agents > p->max_agents) return fail(SF_ERR_ARG, "agents out of range for this policy");
  return SF_OK;
}

static int create(const sf_policy_weights *w, int max_agents, int device, sf_policy **out) {
  if (!w || !out) return fail(SF_ERR_ARG, "null argument");
  if (w->abi_version != SF_POLICY_ABI_VERSION) return fail(SF_ERR_ARG, "sf_policy_weights.abi_version mismatch");
  if (max_agents < 1 || max_agents > (1 << 20)) return fail(SF_ERR_ARG, "max_agents must be 1..1048576");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
    return fail(SF_ERR_DEVICE, "no HIP device: the policy network has no CPU path");
  if (device < 0 || device >= n) return fail(SF_ERR_DEVICE, "device ordinal out of range");
  SFP_HIP(hipSetDevice(device));
  Policy *p = new Policy();
  p->device = device, p->max_agents = max_agents;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      p->sk_blocks = 2 * prop.multiProcessorCount;
  }
  int rc = SF_OK;
#define SFP_TRY(x)   \
  if ((rc = (x))) {  \
    delete p;        \
    return rc;       \
  }
  for (int i = 0; i < 4; ++i)
    if (!w->conv_w[i]) SFP_TRY(fail(SF_ERR_ARG, "conv weight is null"));
  SFP_TRY(p->upload(&p->conv_w[0], w->conv_w[0], (size_t)HID * OBS_C * 9));
  {
    std::vector<float> tr((size_t)OBS_C * 9 * HID);
    for (int nn = 0; nn < HID; ++nn)
      for (int ck = 0; ck < OBS_C * 9; ++ck) tr[(size_t)ck * HID + nn] = w->conv_w[0][(size_t)nn * OBS_C * 9 + ck];
    SFP_TRY(p->upload(&p->conv0_wt, tr.data(), tr.size()));
    const char *e = getenv("SF_POLICY_DENSE_CONV0");
    p->dense_conv0 = e && e[0] == '1';
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv0_sparse<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C0_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv0_sparse<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C0_LDS) != hipSuccess)
      SFP_TRY(fail(SF_ERR_DEVICE, "k_conv0_sparse needs 157 KB of LDS per workgroup"));
  }
  for (int i = 1; i < 4; ++i) {
    // [n][cin][ky][kx] -> [n][ky][kx][cin]: the K-order of an NHWC im2col row
    std::vector<float> perm((size_t)HID * HID * 9);
    for (int nn = 0; nn < HID; ++nn)
      for (int c = 0; c < HID; ++c)
        for (int tap = 0; tap < 9; ++tap)
          perm[((size_t)nn * 9 + tap) * HID + c] = w->conv_w[i][((size_t)nn * HID + c) * 9 + tap];
    SFP_TRY(p->upload(&p->conv_w[i], perm.data(), perm.size()));
    if (i < 3) {
      const std::vector<uint16_t> img = split_weights(perm.data(), HID, HID * 9);
      float *d = nullptr;
      SFP_TRY(p->dalloc(&d, img.size() / 2));
      SFP_HIP(hipMemcpy(d, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      p->conv_w3[i] = d;
    }
  }
  {
    // F = conv3 o conv2 o conv1 o conv0 as one matrix (see k_fold): T3 = conv3's rows, then three transposed convolutions
    const char *e = getenv("SF_POLICY_LAYERED");
    p->folded = !(e && e[0] == '1');
    if (p->folded) {
      double *T[4] = {nullptr, nullptr, nullptr, nullptr};
      const int side[4] = {31, 15, 7, 3};
      const size_t elems[4] = {(size_t)HID * 31 * 31 * OBS_C, (size_t)HID * 15 * 15 * HID, (size_t)HID * 7 * 7 * HID, (size_t)HID * 9 * HID};
      bool ok = true;
      for (int i = 0; i < 4 && ok; ++i) ok = hipMalloc(reinterpret_cast<void **>(&T[i]), elems[i] * sizeof(double)) == hipSuccess;
      if (ok) ok = p->dalloc(&p->fold, (size_t)FD_ROWS * HID) == SF_OK;
      if (ok) {
        (void)hipMemset(p->fold, 0, (size_t)FD_ROWS * HID * sizeof(float));
        hipLaunchKernelGGL(k_to_f64, dim3((unsigned)((elems[3] + 255) / 256)), dim3(256), 0, nullptr, p->conv_w[3], T[3], elems[3]);
        for (int i = 2; i >= 0; --i)  // T[i]: the map from conv i's input to feat
          hipLaunchKernelGGL(k_fold, dim3((unsigned)((elems[i] + 255) / 256)), dim3(256), 0, nullptr, T[i + 1], p->conv_w[i], T[i],
                             side[i + 1], side[i], HID, i ? HID : OBS_C, i ? 1 : 0);
        hipLaunchKernelGGL(k_fold_out, dim3((unsigned)(((size_t)OBS_F * HID + 255) / 256)), dim3(256), 0, nullptr, T[0], p->fold);
        // (a launch that never started — a bad configuration — reports here, not at the synchronise: F would stay all zero)
        ok = hipGetLastError() == hipSuccess;
        ok = hipDeviceSynchronize() == hipSuccess && ok;
      }
      for (int i = 0; i < 4; ++i)
        if (T[i]) (void)hipFree(T[i]);
      if (!ok) SFP_TRY(fail(SF_ERR_DEVICE, "composing the convolution stack failed"));
    }
  }
  {
    const char *e = getenv("SF_POLICY_F32_CONV");
    p->f32_conv = e && e[0] == '1';
    const char *ft = getenv("SF_POLICY_FUSED_TAIL");
    p->fused_tail = !(ft && ft[0] == '0');
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TL_LDS) != hipSuccess)
      SFP_TRY(fail(SF_ERR_DEVICE, "k_tail needs 155 KB of LDS per workgroup"));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_b3<MODE_NHWC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)B3_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_b3<MODE_DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)B3_LDS) != hipSuccess)
      SFP_TRY(fail(SF_ERR_DEVICE, "k_gemm_b3 needs 78 KB of LDS per workgroup"));
  }
  for (int g = 0; g < 2; ++g) {
    SFP_TRY(p->upload(&p->gru_w_ih[g], w->gru_w_ih[g], (size_t)G3 * HID));
    SFP_TRY(p->upload(&p->gru_w_hh[g], w->gru_w_hh[g], (size_t)G3 * HID));
    SFP_TRY(p->upload(&p->gru_b_ih[g], w->gru_b_ih[g], G3));
    SFP_TRY(p->upload(&p->gru_b_hh[g], w->gru_b_hh[g], G3));
  }
  {
    if (!w->comb_w) SFP_TRY(fail(SF_ERR_ARG, "combined_processor weight is null"));
    std::vector<float> pad((size_t)HID * COMB_PAD, 0.f);
    for (int nn = 0; nn < HID; ++nn) std::memcpy(&pad[(size_t)nn * COMB_PAD], w->comb_w + (size_t)nn * COMB, COMB * sizeof(float));
    SFP_TRY(p->upload(&p->comb_w, pad.data(), pad.size()));
    // k_tail's block, in the order the kernel goes through it
    SFP_TRY(p->stage_tiles(&p->gru_w_ih_t[0], w->gru_w_ih[0], G3, HID));
    SFP_TRY(p->stage_tiles(&p->gru_w_hh_t[0], w->gru_w_hh[0], G3, HID));
    SFP_TRY(p->stage_tiles(&p->comb_w_t, pad.data(), HID, COMB_PAD));
    SFP_TRY(p->stage_tiles(&p->gru_w_ih_t[1], w->gru_w_ih[1], G3, HID));
    SFP_TRY(p->stage_tiles(&p->gru_w_hh_t[1], w->gru_w_hh[1], G3, HID));
    SFP_TRY(p->upload(&p->comb_b, w->comb_b, HID));
  }
  for (int i = 0; i < 3; ++i) {
    SFP_TRY(p->upload(&p->res_w[0][i], w->policy_res_w[i], (size_t)HID * HID));
    SFP_TRY(p->stage_tiles(&p->res_w_t[0][i], w->policy_res_w[i], HID, HID));
    SFP_TRY(p->stage_tiles(&p->res_w_t[1][i], w->value_res_w[i], HID, HID));
    SFP_TRY(p->upload(&p->res_b[0][i], w->policy_res_b[i], HID));
    SFP_TRY(p->upload(&p->res_w[1][i], w->value_res_w[i], (size_t)HID * HID));
    SFP_TRY(p->upload(&p->res_b[1][i], w->value_res_b[i], HID));
  }
  SFP_TRY(p->upload(&p->head_w[0], w->policy_w, (size_t)ACT * HID));
  SFP_TRY(p->upload(&p->head_b[0], w->policy_b, ACT));
  SFP_TRY(p->upload(&p->head_w[1], w->value_w, HID));
  SFP_TRY(p->upload(&p->head_b[1], w->value_b, 1));
  for (int g = 0; g < 2; ++g) {
    const int rows = g ? 1 : ACT;
    std::vector<float> wpad((size_t)16 * HID, 0.f), bpad(16, 0.f);
    std::memcpy(wpad.data(), g ? w->value_w : w->policy_w, (size_t)rows * HID * sizeof(float));
    std::memcpy(bpad.data(), g ? w->value_b : w->policy_b, (size_t)rows * sizeof(float));
    SFP_TRY(p->upload(&p->head_w16[g], wpad.data(), wpad.size()));
    SFP_TRY(p->stage_tiles(&p->head_w16_t[g], wpad.data(), 16, HID));
    SFP_TRY(p->upload(&p->head_b16[g], bpad.data(), bpad.size()));
  }
  SFP_TRY(p->upload(&p->tail_w, p->tail_stage.data(), p->tail_stage.size()));
  std::vector<float>().swap(p->tail_stage);
  const size_t B = (size_t)max_agents;
  SFP_TRY(p->dalloc(&p->h[0], B * HID));
  SFP_TRY(p->dalloc(&p->h[1], B * HID));
  SFP_TRY(p->dalloc(&p->action_input, B * ACT));
  if (!p->folded) {  // the layered path's activations (181 KB per agent)
    SFP_TRY(p->dalloc(&p->act[0], B * 225 * HID));
    SFP_TRY(p->dalloc(&p->act[1], B * 49 * HID));
    SFP_TRY(p->dalloc(&p->act[2], B * 9 * HID));
  }
  SFP_TRY(p->dalloc(&p->feat, B * HID));
  SFP_TRY(p->dalloc(&p->feat_n, B * HID));
  SFP_TRY(p->dalloc(&p->gi, B * G3));
  SFP_TRY(p->dalloc(&p->gh, B * G3));
  SFP_TRY(p->dalloc(&p->comb, B * COMB_PAD));
  SFP_TRY(p->dalloc(&p->gated, B * HID));
  SFP_TRY(p->dalloc(&p->gated_n, B * HID));
  SFP_TRY(p->dalloc(&p->out, B * HID));
  SFP_TRY(p->dalloc(&p->x[0], B * HID));
  SFP_TRY(p->dalloc(&p->x[1], B * HID));
  SFP_TRY(p->dalloc(&p->lin[0], B * HID));
  SFP_TRY(p->dalloc(&p->lin[1], B * HID));
  SFP_TRY(p->dalloc(&p->part, (size_t)p->sk_blocks * 2 * 128 * BN));
  {
    float *f = nullptr;
    SFP_TRY(p->dalloc(&f, 1));
    p->d_overflows = reinterpret_cast<uint32_t *>(f);
    SFP_HIP(hipMemset(p->d_overflows, 0, sizeof(uint32_t)));
  }
#undef SFP_TRY
  *out = reinterpret_cast<sf_policy *>(p);
  return sf_policy_reset_memory(*out, nullptr);
}

// li + d_obs together: list form with a dense fallback for the agents whose list did not fit (d_obs need only be valid
// for those: sf_observe_overflow_device)
struct PredictExtra {  // what sf_policy_predict_sparse folds into k_tail (see TailArgs)
  const uint8_t *reset_mask;
  const int32_t *reset_words;
  int reset_stride, reset_group;
  ActStr as;
  uint64_t seed;
  int greedy;
  uint8_t *cmd;
  int32_t *action;
};
static int forward(Policy *p, const float *d_obs, int agents, float *d_probs, float *d_value, const C0List *li = nullptr,
                   const float *d_pov = nullptr, const PredictExtra *px = nullptr) {
  int rc = check_agents(p, agents);
  if (rc) return rc;
  if ((!d_obs && !li) || !d_probs || !d_value) return fail(SF_ERR_ARG, "null buffer");
  if (li && !p->fused_tail) return fail(SF_ERR_STATE, "sf_policy_forward_sparse needs the fused tail (SF_POLICY_FUSED_TAIL=0 is set)");
  if (px && !p->fused_tail) return fail(SF_ERR_STATE, "sf_policy_predict_sparse needs the fused tail (SF_POLICY_FUSED_TAIL=0 is set)");
  SFP_HIP(hipSetDevice(p->device));
  const dim3 rg((unsigned)((agents + 3) / 4)), rb(256);
  hipStream_t st = p->stream;
  // GameCNN                                                                    Modules.hpp:66-71
  const dim3 c0_grid((unsigned)(agents < p->sk_blocks / 2 ? agents : p->sk_blocks / 2));
  if (p->folded) {
    // the four convolutions as one matrix applied to the non-zeros (k_feat_*), timed as kind 2
    hipEvent_t e0 = nullptr;
    if ((rc = p->time_begin(0.0, false, &e0, 2))) return rc;  // useful flop depends on the lists: the caller counts the non-zeros
    const dim3 fd_grid((unsigned)(agents < 4 * p->sk_blocks ? agents : 4 * p->sk_blocks));
    if (li) {
      hipLaunchKernelGGL(k_feat_list, dim3((unsigned)agents), dim3(64), 0, st, p->fold, p->feat, agents, *li);
      if (e0) SFP_HIP(hipEventRecord(e0, st));
      if (d_obs)  // redo the agents whose list did not fit from their dense observation (none, normally: the launch is idle)
        hipLaunchKernelGGL(k_feat_dense, dim3((unsigned)(agents < p->sk_blocks ? agents : p->sk_blocks)), dim3(FD_T), 0, st, d_obs, p->fold,
                           p->feat, agents, *li);
    } else {
      hipLaunchKernelGGL(k_feat_dense, fd_grid, dim3(FD_T), 0, st, d_obs, p->fold, p->feat, agents, C0List{});
      if (e0) SFP_HIP(hipEventRecord(e0, st));
    }
  } else if (li) {
    hipEvent_t e0 = nullptr;
    if ((rc = p->time_begin(0.0, false, &e0, 2))) return rc;  // useful flop depends on the lists: the caller counts the non-zeros
    hipLaunchKernelGGL(k_conv0_sparse<true>, c0_grid, dim3(C0_T), C0_LDS, st, (const float *)nullptr, p->conv0_wt, p->act[0], agents, *li);
    if (e0) SFP_HIP(hipEventRecord(e0, st));
    if (d_obs)  // redo the agents whose list did not fit from their dense observation (none, normally: the launch is idle)
      hipLaunchKernelGGL(k_conv0_sparse<false>, c0_grid, dim3(C0_T), C0_LDS, st, d_obs, p->conv0_wt, p->act[0], agents, *li);
  } else if (p->dense_conv0) {
    if ((rc = p->conv(d_obs, p->conv_w[0], p->act[0], agents, 31, OBS_C, 1))) return rc;
  } else {
    hipLaunchKernelGGL(k_conv0_sparse<false>, c0_grid, dim3(C0_T), C0_LDS, st, d_obs, p->conv0_wt, p->act[0], agents, C0List{});
  }
  if (!p->folded) {
    if ((rc = p->conv(p->act[0], p->conv_w[1], p->act[1], agents, 15, HID, 0, p->conv_w3[1]))) return rc;
    if ((rc = p->conv(p->act[1], p->conv_w[2], p->act[2], agents, 7, HID, 0, p->conv_w3[2]))) return rc;
  }
  if (p->fused_tail) {
    // (timed as one launch on the f32 pipe: conv3 unless folded + 4 GRU gate products + combined_processor + 6 ResB layers)
    hipEvent_t e1 = nullptr;
    if ((rc = p->time_begin(2.0 * agents * ((p->folded ? 0.0 : (double)HID * 9 * HID) + 4.0 * G3 * HID + (double)HID * COMB_PAD + 6.0 * HID * HID), false, &e1, 3)))
      return rc;
    TailArgs t{};
    t.act2 = p->act[2], t.obs = d_obs, t.pov = d_pov, t.conv3_w = p->conv_w[3];
    t.feat = p->folded ? p->feat : nullptr;
    for (int g = 0; g < 2; ++g) {
      // (the matrices in k_tail's stream order, stage_tiles)
      t.gru_w_ih[g] = p->tail_w + p->gru_w_ih_t[g], t.gru_w_hh[g] = p->tail_w + p->gru_w_hh_t[g], t.gru_b_ih[g] = p->gru_b_ih[g], t.gru_b_hh[g] = p->gru_b_hh[g];
      t.h[g] = p->h[g], t.head_w[g] = p->tail_w + p->head_w16_t[g], t.head_b[g] = p->head_b16[g];
      for (int i = 0; i < 3; ++i) t.res_w[g][i] = p->tail_w + p->res_w_t[g][i], t.res_b[g][i] = p->res_b[g][i];
    }
    t.comb_w = p->tail_w + p->comb_w_t, t.comb_b = p->comb_b, t.action_input = p->action_input;

    t.probs = d_probs, t.value = d_value, t.agents = agents;
    t.reset_group = 1;
    if (px) {
      t.reset_mask = px->reset_mask, t.reset_words = px->reset_words, t.reset_stride = px->reset_stride;
      t.reset_group = px->reset_group > 0 ? px->reset_group : 1;
      t.act = 1, t.greedy = px->greedy, t.as = px->as, t.seed = px->seed, t.draw = p->draws++, t.cmd = px->cmd, t.action = px->action;
    }
    hipLaunchKernelGGL(k_tail, dim3((unsigned)((agents + TL_R - 1) / TL_R)), dim3(TL_T), TL_LDS, st, t);
#ifdef SF_DIAG_TAIL  // (diagnostic build: the stamps of a second launch, whose weights the first one left in the L2s)
    if (std::getenv("SF_DIAG_TAIL_TWICE")) hipLaunchKernelGGL(k_tail, dim3((unsigned)((agents + TL_R - 1) / TL_R)), dim3(TL_T), TL_LDS, st, t);
#endif
    SFP_HIP(hipGetLastError());
    if (e1) SFP_HIP(hipEventRecord(e1, st));
    return SF_OK;
  }
  if (!p->folded && (rc = p->conv(p->act[2], p->conv_w[3], p->feat, agents, 3, HID, 0))) return rc;
  float *const none = nullptr;
  hipLaunchKernelGGL(k_norm, rg, rb, 0, st, p->feat, p->feat_n, none, agents);               // :108
  // gru0 (both gate products in one launch)                                                    :110-113
  if ((rc = p->dense2(p->feat_n, p->gru_w_ih[0], p->gru_b_ih[0], p->gi, p->h[0], p->gru_w_hh[0], p->gru_b_hh[0], p->gh,
                      HID, G3, agents, G3, HID)))
    return rc;
  hipLaunchKernelGGL(k_gru0, rg, rb, 0, st, p->gi, p->gh, p->h[0], p->feat_n, d_obs, p->action_input, p->comb, agents);
  // combined_processor                                                                         :125-126
  if ((rc = p->dense(p->comb, COMB_PAD, p->comb_w, p->comb_b, p->gated, HID, agents, HID, COMB_PAD))) return rc;
  hipLaunchKernelGGL(k_norm, rg, rb, 0, st, p->gated, p->gated_n, none, agents);
  // gru1                                                                                       :128-131
  if ((rc = p->dense2(p->gated_n, p->gru_w_ih[1], p->gru_b_ih[1], p->gi, p->h[1], p->gru_w_hh[1], p->gru_b_hh[1], p->gh,
                      HID, G3, agents, G3, HID)))
    return rc;
  hipLaunchKernelGGL(k_gru1, rg, rb, 0, st, p->gi, p->gh, p->h[1], p->gated_n, p->out, agents);
  // heads: ResB then Linear; layer i of both heads shares a launch                             :41-48,172-175
  hipLaunchKernelGGL(k_norm, rg, rb, 0, st, p->out, p->x[0], p->x[1], agents);
  for (int i = 0; i < 3; ++i) {
    if ((rc = p->dense2(p->x[0], p->res_w[0][i], p->res_b[0][i], p->lin[0], p->x[1], p->res_w[1][i], p->res_b[1][i],
                        p->lin[1], HID, HID, agents, HID, HID)))
      return rc;
    hipLaunchKernelGGL(k_res, dim3(rg.x, 2), rb, 0, st, p->lin[0], p->x[0], p->lin[1], p->x[1], agents);
  }
  hipLaunchKernelGGL(k_heads, rg, rb, 0, st, p->x[0], p->x[1], p->head_w[0], p->head_b[0], p->head_w[1], p->head_b[1],
                     d_probs, d_value, agents);
  SFP_HIP(hipGetLastError());
  return SF_OK;
}

}  // namespace sfp

using sfp::Policy;

extern "C" {

int sf_policy_abi_version(void) { return SF_POLICY_ABI_VERSION; }

int sf_policy_create(const sf_policy_weights *w, int32_t max_agents, int32_t device, sf_policy **out) {
  return sfp::create(w, max_agents, device, out);
}

void sf_policy_destroy(sf_policy *p) { delete reinterpret_cast<Policy *>(p); }

int sf_policy_reset_memory_n(sf_policy *pp, const uint8_t *d_mask, int32_t agents) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  int rc = sfp::check_agents(p, agents);
  if (rc) return rc;
  SFP_HIP(hipSetDevice(p->device));
  const int n = agents * sfp::HID;
  hipLaunchKernelGGL(sfp::k_reset_memory, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, p->h[0], p->h[1],
                     p->action_input, d_mask, agents);
  SFP_HIP(hipGetLastError());
  return SF_OK;
}
int sf_policy_reset_memory(sf_policy *pp, const uint8_t *d_mask) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  return sf_policy_reset_memory_n(pp, d_mask, p->max_agents);
}

int sf_policy_forward(sf_policy *pp, const float *d_obs, int32_t agents, float *d_probs, float *d_value) {
  return sfp::forward(reinterpret_cast<Policy *>(pp), d_obs, agents, d_probs, d_value);
}

int sf_policy_features(sf_policy *pp, const float *d_obs, int32_t agents, float *d_feat) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  int rc = sfp::check_agents(p, agents);
  if (rc) return rc;
  if (!d_obs || !d_feat) return sfp::fail(SF_ERR_ARG, "null buffer");
  SFP_HIP(hipSetDevice(p->device));
  if (p->folded) {
    const dim3 grid((unsigned)(agents < 4 * p->sk_blocks ? agents : 4 * p->sk_blocks));
    hipLaunchKernelGGL(sfp::k_feat_dense, grid, dim3(sfp::FD_T), 0, p->stream, d_obs, p->fold, d_feat, agents, sfp::C0List{});
  } else {
    const dim3 c0_grid((unsigned)(agents < p->sk_blocks / 2 ? agents : p->sk_blocks / 2));
    hipLaunchKernelGGL(sfp::k_conv0_sparse<false>, c0_grid, dim3(sfp::C0_T), sfp::C0_LDS, p->stream, d_obs, p->conv0_wt, p->act[0], agents,
                       sfp::C0List{});
    if ((rc = p->conv(p->act[0], p->conv_w[1], p->act[1], agents, 15, sfp::HID, 0, p->conv_w3[1]))) return rc;
    if ((rc = p->conv(p->act[1], p->conv_w[2], p->act[2], agents, 7, sfp::HID, 0, p->conv_w3[2]))) return rc;
    if ((rc = p->conv(p->act[2], p->conv_w[3], d_feat, agents, 3, sfp::HID, 0))) return rc;
  }
  SFP_HIP(hipGetLastError());
  return SF_OK;
}

int sf_policy_forward_sparse(sf_policy *pp, const uint32_t *d_keys, const float *d_vals, const uint32_t *d_counts,
                             const float *d_pov, int32_t cap, int32_t agents, float *d_probs, float *d_value) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  if (!d_keys || !d_vals || !d_counts || !d_pov || cap < 1) return sfp::fail(SF_ERR_ARG, "null buffer or cap < 1");
  if (cap > SF_POLICY_LIST_MAX) return sfp::fail(SF_ERR_ARG, "cap above SF_POLICY_LIST_MAX (2048)");
  const sfp::C0List li{d_keys, d_vals, d_counts, cap, p->d_overflows};
  return sfp::forward(p, nullptr, agents, d_probs, d_value, &li, d_pov);
}

int sf_policy_forward_sparse_or_dense(sf_policy *pp, const uint32_t *d_keys, const float *d_vals, const uint32_t *d_counts,
                                      const float *d_pov, int32_t cap, int32_t agents, const float *d_dense, float *d_probs,
                                      float *d_value) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  if (!d_keys || !d_vals || !d_counts || !d_pov || !d_dense || cap < 1) return sfp::fail(SF_ERR_ARG, "null buffer or cap < 1");
  // (above it the kernels' own list limit would call an agent "overflowed" whose dense row sf_observe_overflow_device,
  // which only knows cap, never wrote)
  if (cap > SF_POLICY_LIST_MAX) return sfp::fail(SF_ERR_ARG, "cap above SF_POLICY_LIST_MAX (2048)");
  const sfp::C0List li{d_keys, d_vals, d_counts, cap, nullptr};
  return sfp::forward(p, d_dense, agents, d_probs, d_value, &li, d_pov);
}

int sf_policy_predict_sparse(sf_policy *pp, const sf_policy_predict_io *io, int32_t agents) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || !io) return sfp::fail(SF_ERR_ARG, "null policy or io");
  if (!io->d_keys || !io->d_vals || !io->d_counts || !io->d_pov || io->cap < 1) return sfp::fail(SF_ERR_ARG, "null buffer or cap < 1");
  if (io->cap > SF_POLICY_LIST_MAX) return sfp::fail(SF_ERR_ARG, "cap above SF_POLICY_LIST_MAX (2048)");
  if (!io->d_cmd || !io->action_string) return sfp::fail(SF_ERR_ARG, "null buffer");
  if (std::strlen(io->action_string) != (size_t)sfp::ACT) return sfp::fail(SF_ERR_ARG, "action_string must have 9 chars");
  if (io->d_reset_words && (io->reset_stride < 1 || io->reset_group < 1)) return sfp::fail(SF_ERR_ARG, "reset_stride and reset_group must be positive");
  sfp::PredictExtra px{};
  px.reset_mask = io->d_reset_mask, px.reset_words = io->d_reset_words, px.reset_stride = io->reset_stride, px.reset_group = io->reset_group;
  std::memcpy(px.as.c, io->action_string, sfp::ACT);
  px.seed = io->seed, px.greedy = io->greedy, px.cmd = io->d_cmd, px.action = io->d_action;
  // (without d_dense: lists that do not fit are counted, as in sf_policy_forward_sparse)
  const sfp::C0List li{io->d_keys, io->d_vals, io->d_counts, io->cap, io->d_dense ? nullptr : p->d_overflows};
  return sfp::forward(p, io->d_dense, agents, io->d_probs, io->d_value, &li, io->d_pov, &px);
}

#ifdef SF_DIAG_TAIL
int sf_policy_diag_tail_read(uint32_t *out, int32_t workgroups) {  // diagnostic build only: [workgroups][16 waves][24 phases] cycles
  if (hipDeviceSynchronize() != hipSuccess) return SF_ERR_DEVICE;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sfp::sf_diag_tail), (size_t)workgroups * 16 * 24 * sizeof(uint32_t)) == hipSuccess ? SF_OK : SF_ERR_DEVICE;
}
#endif
int sf_policy_sparse_overflows(sf_policy *pp, int32_t *count) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || !count) return sfp::fail(SF_ERR_ARG, "null argument");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  uint32_t n = 0;
  SFP_HIP(hipMemcpy(&n, p->d_overflows, sizeof(n), hipMemcpyDeviceToHost));
  SFP_HIP(hipMemset(p->d_overflows, 0, sizeof(n)));
  *count = (int32_t)n;
  return SF_OK;
}

int sf_policy_act(sf_policy *pp, const float *d_probs, int32_t agents, const char *action_string, uint64_t seed,
                  int32_t greedy, uint8_t *d_cmd, int32_t *d_action) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  int rc = sfp::check_agents(p, agents);
  if (rc) return rc;
  if (!d_probs || !d_cmd || !action_string) return sfp::fail(SF_ERR_ARG, "null buffer");
  if (std::strlen(action_string) != (size_t)sfp::ACT) return sfp::fail(SF_ERR_ARG, "action_string must have 9 chars");
  sfp::ActStr as;
  std::memcpy(as.c, action_string, sfp::ACT);
  SFP_HIP(hipSetDevice(p->device));
  hipLaunchKernelGGL(sfp::k_act, dim3((unsigned)((agents + 255) / 256)), dim3(256), 0, p->stream, d_probs,
                     p->action_input, as, seed, p->draws++, greedy, d_cmd, d_action, agents);
  SFP_HIP(hipGetLastError());
  return SF_OK;
}

int sf_policy_gemm(sf_policy *pp, const float *d_a, int32_t lda, const float *d_w, const float *d_bias, float *d_c,
                   int32_t ldc, int32_t m, int32_t n, int32_t k) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || !d_a || !d_w || !d_c) return sfp::fail(SF_ERR_ARG, "null argument");
  if (lda % 4 || ldc % 4 || lda < k || ldc < n) return sfp::fail(SF_ERR_ARG, "policy gemm: bad leading dimension");
  SFP_HIP(hipSetDevice(p->device));
  return p->dense(d_a, lda, d_w, d_bias, d_c, ldc, m, n, k);
}

int sf_policy_gemm_split(sf_policy *pp, const float *d_a, int32_t lda, const float *d_w, const float *d_bias, float *d_c,
                         int32_t ldc, int32_t m, int32_t n, int32_t k) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || !d_a || !d_w || !d_c) return sfp::fail(SF_ERR_ARG, "null argument");
  if (lda % 4 || ldc % 4 || lda < k || ldc < n) return sfp::fail(SF_ERR_ARG, "policy gemm: bad leading dimension");
  if (k % 32 || n % sfp::BN || m < 1) return sfp::fail(SF_ERR_ARG, "policy gemm: unsupported shape");
  SFP_HIP(hipSetDevice(p->device));
  void *img = nullptr;
  if (hipMalloc(&img, (size_t)n * k * 3 * sizeof(uint16_t)) != hipSuccess) return sfp::fail(SF_ERR_MEMORY, "hipMalloc failed (split W)");
  const size_t elems = (size_t)n * k;
  hipLaunchKernelGGL(sfp::k_split_weights, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, p->stream, d_w,
                     (uint16_t *)img, n, k);
  sfp::Gemm g{d_a, d_w, d_bias, d_c, m, n, k, lda, ldc, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, img};
  int rc = p->gemm(g, sfp::MODE_DENSE);
  if (hipStreamSynchronize(p->stream) != hipSuccess && !rc) rc = sfp::fail(SF_ERR_DEVICE, "policy gemm (split) failed");
  (void)hipFree(img);
  return rc;
}

int sf_policy_get_memory(sf_policy *pp, int32_t agent, float *h, float *action_input) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || agent < 0 || agent >= p->max_agents || !h || !action_input) return sfp::fail(SF_ERR_ARG, "bad argument");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  for (int g = 0; g < 2; ++g)
    SFP_HIP(hipMemcpy(h + g * sfp::HID, p->h[g] + (size_t)agent * sfp::HID, sfp::HID * sizeof(float), hipMemcpyDeviceToHost));
  SFP_HIP(hipMemcpy(action_input, p->action_input + (size_t)agent * sfp::ACT, sfp::ACT * sizeof(float), hipMemcpyDeviceToHost));
  return SF_OK;
}

int sf_policy_set_memory(sf_policy *pp, int32_t agent, const float *h, const float *action_input) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || agent < 0 || agent >= p->max_agents || !h || !action_input) return sfp::fail(SF_ERR_ARG, "bad argument");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  for (int g = 0; g < 2; ++g)
    SFP_HIP(hipMemcpy(p->h[g] + (size_t)agent * sfp::HID, h + g * sfp::HID, sfp::HID * sizeof(float), hipMemcpyHostToDevice));
  SFP_HIP(hipMemcpy(p->action_input + (size_t)agent * sfp::ACT, action_input, sfp::ACT * sizeof(float), hipMemcpyHostToDevice));
  return SF_OK;
}

int sf_policy_set_stream(sf_policy *pp, void *hip_stream) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  p->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return SF_OK;
}

int sf_policy_synchronize(sf_policy *pp) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  return SF_OK;
}

int sf_policy_kernel_time_ex(sf_policy *pp, int32_t enable, float ms[2], double flop[2], int32_t launches[2]) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p) return sfp::fail(SF_ERR_ARG, "null policy");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  float total[2] = {0.f, 0.f};
  int32_t n[2] = {0, 0};
  for (size_t i = 0; i < p->used_events; ++i) {
    float t = 0.f;
    SFP_HIP(hipEventElapsedTime(&t, p->events[i].first, p->events[i].second));
    if (p->event_split[i] == 2) continue;  // conv0 on the non-zeros: sf_policy_kernel_time_by_kernel
    total[p->event_split[i] == 1 ? 1 : 0] += t;
    ++n[p->event_split[i] == 1 ? 1 : 0];
  }
  for (int k = 0; k < 2; ++k) {
    if (ms) ms[k] = total[k];
    if (flop) flop[k] = k ? p->flop_split : p->flop;
    if (launches) launches[k] = n[k];
  }
  p->used_events = 0;
  p->flop = p->flop_split = 0;
  for (double &f : p->flop_kind) f = 0;
  p->timing = enable != 0;
  return SF_OK;
}

int sf_policy_kernel_time_by_kernel(sf_policy *pp, int32_t enable, float ms[4], double flop[4], int32_t launches[4]) {
  Policy *p = reinterpret_cast<Policy *>(pp);
  if (!p || !ms || !flop || !launches) return sfp::fail(SF_ERR_ARG, "null argument");
  SFP_HIP(hipSetDevice(p->device));
  SFP_HIP(hipStreamSynchronize(p->stream));
  for (int k = 0; k < 4; ++k) ms[k] = 0.f, flop[k] = p->flop_kind[k], launches[k] = 0;
  for (size_t i = 0; i < p->used_events; ++i) {
    float t = 0.f;
    SFP_HIP(hipEventElapsedTime(&t, p->events[i].first, p->events[i].second));
    ms[(int)p->event_split[i]] += t;
    ++launches[(int)p->event_split[i]];
  }
  p->used_events = 0;
  p->flop = p->flop_split = 0;
  for (double &f : p->flop_kind) f = 0;
  p->timing = enable != 0;
  return SF_OK;
}

int sf_policy_kernel_time(sf_policy *pp, int32_t enable, float *ms, double *flop, int32_t *launches) {
  float m[2];
  double f[2];
  int32_t n[2];
  const int rc = sf_policy_kernel_time_ex(pp, enable, m, f, n);
  if (rc) return rc;
  if (ms) *ms = m[0] + m[1];
  if (flop) *flop = f[0] + f[1];
  if (launches) *launches = n[0] + n[1];
  return SF_OK;
}

}  // extern "C"
