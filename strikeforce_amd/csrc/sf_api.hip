// sf_api.hip — libstrikeforce_amd.so: gfx950 kernels + the C-ABI of include/strikeforce.h.
//
// Kernels
//   k_reset<NB>   one wavefront per arena: setup()/load_data()/_srand + first loop top
//   k_step<NB>    one wavefront per arena: K iterations of the gameplay loop per launch, state in
//                 registers, the arena's cell-flag plane in LDS
//   k_observe     one 256-thread workgroup per (arena, agent): 32 x 31 x 31 float observation
// There is no CPU path: without a HIP device every entry point fails with SF_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstddef>

#include <new>
#include <utility>
#include <vector>

#include "wave_gfx950.hpp"
// clang-format off
#include "sf_core.hpp"
#include "sf_obs.hpp"
#include "sf_host.hpp"
// clang-format on

namespace sf {

#ifdef SF_DIAG_STAMPS
__device__ uint32_t sf_diag_buffer[16 * 65536];  // diagnostic build only: [arena][phase] wave cycles of the last launch
__device__ unsigned long long sf_diag_times[4 * 65536];  // [workgroup]: s_memrealtime (100 MHz) at the wave's start, after load(), before store(), at its end
#endif

// HP (HBM_PLANE): maps whose flag plane is too large for LDS keep it in HBM (sf_core.hpp).  BM (BITMAPS): the cell
// bitmaps fit in LDS (every LDS-plane map, and HBM-plane maps up to 128 x 128).  Variants built: (HP 0, BM 1),
// (HP 1, BM 1), (HP 1, BM 0).  ZL (large pools): zombie / exit tables of more than 64 slots in LDS; built for NB = 4 only.
template <int NB, bool HP, bool BM, bool ZL>
__global__ __launch_bounds__(64) void k_reset(Params p, const uint64_t *tb, const uint64_t *serial) {
  extern __shared__ __attribute__((aligned(2048))) uint8_t lds[];  // the RNG power table comes first (W::pow_pair)
  Core<WaveGfx950, NB, HP, BM, ZL>::reset_body(lds, p, (int)blockIdx.x, tb, serial);
}

template <int NB, bool HP, bool BM, bool ZL>
__global__ __launch_bounds__(64) void k_step(Params p, const uint8_t *cmds, int k) {
  extern __shared__ __attribute__((aligned(2048))) uint8_t lds[];  // the RNG power table comes first (W::pow_pair)
  const int a = p.perm ? (int)gptr(p.perm)[blockIdx.x] : (int)blockIdx.x;
#ifdef SF_DIAG_STAMPS
  if (threadIdx.x == 0) sf_diag_times[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  Core<WaveGfx950, NB, HP, BM, ZL>::step_body(lds, p, a, cmds, k);
#ifdef SF_DIAG_STAMPS
  __builtin_amdgcn_s_waitcnt(0);  // (the stores have left)
  if (threadIdx.x == 0) sf_diag_times[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Launch order for k_step: arenas by population (live zombies + live humans as the last store() recorded them, SC_LOAD),
// in SNAKE order — blocks of 1024 alternately descending and ascending.  A step's cost grows with the arena's population
// and a launch ends with its slowest wavefront; the chip has 1024 SIMDs and the dispatcher hands consecutive workgroups
// to different ones, so with this order the arenas that share a SIMD are one of each load class with about equal sums
// per SIMD, the busiest arenas start first, and a busy arena's neighbours finish early and leave it the SIMD.  Measured
// on configs[2] (same-call A/B, tools/r03_balance_ab.sh): 186.7 -> 204.8 M env-steps/s, 20-step launches 172 -> 189 M;
// configs[1] (all arenas at the zombie cap: nothing to order) unchanged; keyed by the arena's measured cycles per step
// instead: +1.6 % only (an arena's time depends on its neighbours, so that key chases itself).  Other arena counts and
// the HBM-plane maps: profiles/r04_rank_sweep.txt.
// One workgroup, any number of arenas (each thread walks arenas t, t + 1024, ...); a counting sort over 65 classes, ties
// in any order: arenas are independent, the order never shows in any result.  A last block of fewer than 1024 arenas
// keeps the direction of a full one (its lightest arenas meet the SIMDs that got the busiest of the block before).
__global__ __launch_bounds__(1024) void k_rank(Params p, uint32_t *perm) {
  __shared__ uint32_t hist[72];
  const int t = (int)threadIdx.x;
  if (t < 72) hist[t] = 0u;
  __syncthreads();
  auto cls_of = [&](int a) {
    const uint32_t n = (uint32_t)gptr(p.scal)[(size_t)a * SC_WORDS + SC_LOAD];
    return 64u - (n > 64u ? 64u : n);  // class 0 = the busiest
  };
  for (int a = t; a < p.A; a += 1024) atomicAdd(&hist[cls_of(a)], 1u);
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int c = 0; c < 65; ++c) {
      const uint32_t n = hist[c];
      hist[c] = run;
      run += n;
    }
  }
  __syncthreads();
  for (int a = t; a < p.A; a += 1024) {
    uint32_t pos = atomicAdd(&hist[cls_of(a)], 1u);  // rank, busiest first
    if ((pos >> 10) & 1u) {  // every second block of 1024 backwards
      const uint32_t base = pos & ~1023u, len = (uint32_t)p.A - base < 1024u ? (uint32_t)p.A - base : 1024u;
      pos = base + (len - 1u - (pos - base));
    }
    gptr(perm)[pos] = (uint32_t)a;
  }
}

// sf_step_begin / sf_step_end: one half of one iteration (sf_core.hpp step<1> / step<2>)
template <int NB, bool HP, bool BM, bool ZL>
__global__ __launch_bounds__(64) void k_step_half(Params p, const uint8_t *cmds, int phase) {
  extern __shared__ __attribute__((aligned(2048))) uint8_t lds[];
  Core<WaveGfx950, NB, HP, BM, ZL>::step_half_body(lds, p, (int)blockIdx.x, cmds, phase);
}

// Human::active_agent of every commanded human (sf_agent_alive): alive and still driven through sf_step
__global__ void k_agent_alive(Params p, uint8_t *out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= p.A * p.n_agents) return;
  const int a = i / p.n_agents, g = i % p.n_agents;
  const uint32_t fl = gptr(p.hum)[((size_t)HW_FLAGS * (size_t)p.A + (size_t)a) * (size_t)p.H + (size_t)g];
  gptr(out)[i] = (uint8_t)((fl & (HF_ALIVE | HF_CTRL)) == (HF_ALIVE | HF_CTRL));
}

// check_end()'s verdict per (arena, agent) on the device (sf_done_device)
__global__ void k_done(Params p, uint8_t *out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= p.A * p.n_agents) return;
  const SF_GLOBAL int32_t *sc = gptr(p.scal) + (size_t)(i / p.n_agents) * SC_WORDS;
  gptr(out)[i] = (uint8_t)(p.auto_reset ? sc[SC_ENDED] : sc[SC_DONE]);
}

constexpr int OBS_W2 = SF_OBS_WINDOW * SF_OBS_WINDOW;  // 961
constexpr int OBS_THREADS = 256;

// workgroup barrier that orders LDS traffic only: global stores issued before it stay in flight
static __device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int OBS_Z_STAGE = 256;    // zombie tables up to this many slots are staged in LDS with the other entities
static inline size_t obs_lds_bytes(const Params &p) {
  return (size_t)(HW_WORDS * p.H + (p.Z <= OBS_Z_STAGE ? ZW_WORDS * p.Z : 0) + BW_WORDS * p.B) * sizeof(uint32_t) + sizeof(int32_t) * 12 +
         sizeof(Derived) * (size_t)(p.npc_block + 1);
}
constexpr int OBS_CLASS_RECS = 8;   // shared records of plain static cells: '#', '^', 'v', 'O', chest types 0-3
constexpr int OBS_REC_MAX = 72;     // + cells with an entity or a player-built object on them (own record each)
constexpr int OBS_LIST_MAX = 256;  // (a) values that need a real pow, (b) overflow cells' outputs
constexpr uint32_t OBS_NOREC = 255u;

// One workgroup per (arena, agent).  The 123 KB observation is written exactly once, with 16-B-per-lane stores
// that cover whole 128-B lines (scattered 4-byte stores of the few non-zero values cost more HBM time than the
// whole zero stream), so everything is first assembled in LDS:
//   prologue  all HBM reads: the arena's entity tables -> LDS (coalesced), the window's flag bytes / damage
//   pass 2    every entity scatters itself into the window's occupant words (LDS atomics)
//   pass 3    one thread per non-empty window cell builds that cell's 32-float record in LDS; values come from
//             the host-built constant table, the rest (a few per entity) are queued for a real x^(1/5)
//   pass 3b   the queued double-precision pows run densely, one per lane, and land in the records
//   pass 4    stream the output: each lane produces 4 consecutive floats by looking up cell -> record
// record shared by every window cell with this flag byte and nothing on it, or -1
static __device__ __forceinline__ int obs_class_of(uint32_t fl) {
  // (a chain of selects: as a `switch` this became a tree of divergent branches, ~200 scalar mask instructions per use —
  // most of k_observe_list's classify phase)
  int c = -1;
  c = fl == SF_CELL_WALL ? 0 : c;
  c = fl == SF_CELL_PIN_UP ? 1 : c;
  c = fl == SF_CELL_PIN_DN ? 2 : c;
  c = fl == SF_CELL_POUT ? 3 : c;
  c = fl == (SF_CELL_CHEST | (0u << SF_CELL_CONS_SHIFT)) ? 4 : c;
  c = fl == (SF_CELL_CHEST | (1u << SF_CELL_CONS_SHIFT)) ? 5 : c;
  c = fl == (SF_CELL_CHEST | (2u << SF_CELL_CONS_SHIFT)) ? 6 : c;
  c = fl == (SF_CELL_CHEST | (3u << SF_CELL_CONS_SHIFT)) ? 7 : c;
  return c;
}
// mode 0: plain.  mode 1: plain + record which floats are non-zero in nzprev.  mode 2 (sf_observe_device_delta): the
// buffer still holds what the previous call left, nzprev says which floats of it are non-zero: only 16-byte pieces
// with an old or a new non-zero are written.
// mode 3 (sf_observe_sparse_device): no dense buffer at all — the non-zero floats leave as a list in the dense buffer's
// scan order (key = channel * 9 | y << 9 | x << 14: the form k_conv0_sparse's list has, value), `cap` entries per agent at
// most (counts[] says how many there were: more than cap, or 0xffffffff for a window too crowded for the records, tells the
// caller to take the dense path), plus the 160 values around the window's centre that the network reads directly.
struct ObsSparse {
  uint32_t *keys;
  float *vals;
  uint32_t *counts;
  float *pov;
  int cap;
};
// one agent's window (workgroup `bid` of the plain launch)
static __device__ __forceinline__ void observe_agent(const Params &p, float *out, uint32_t *nzprev, int mode, const ObsSparse &sp, const unsigned bid) {
  extern __shared__ __attribute__((aligned(16))) uint32_t ent[];  // [13][H] humans, [3][Z] zombies, [4][B] bullets
  __shared__ float rec[OBS_REC_MAX][SF_OBS_CHANNELS];
  __shared__ uint32_t occ[OBS_W2];
  __shared__ int32_t wdmg[OBS_W2];
  __shared__ uint8_t wfl[OBS_W2 + 3];
  __shared__ uint8_t slot[OBS_W2 + 3];
  __shared__ uint32_t list_idx[OBS_LIST_MAX];
  __shared__ float list_val[OBS_LIST_MAX];
  __shared__ uint32_t nzmap[OBS_W2];  // one bit per output float: non-zero (30752 bits)
  uint32_t *ormap = reinterpret_cast<uint32_t *>(wdmg);  // delta mode, pass 4: the previous call's map (wdmg is dead
                                                          // by then unless a spill follows, and a spill disables delta)
  __shared__ uint32_t cmask[OBS_CLASS_RECS];  // non-zero channels of each class record
  __shared__ uint16_t work[OBS_REC_MAX];      // window cell of record r (r >= OBS_CLASS_RECS)
  __shared__ float t_in[16], t_out[16];       // Tables::obs_in / obs_out (obs_in[0] == 1.0)
  // the leading part of Tables that obs_cell_emit reads through ObsView::tab (cons_items, then the used blocks of
  // der[]): an LDS copy (behind the entity tables in the dynamic allocation), so that the one lane describing a human
  // cell does not walk four dependent L2 loads in get_damage_effect
  const int TAB_WORDS = (int)((sizeof(int32_t) * 12 + sizeof(Derived) * (size_t)(p.npc_block + 1)) / 4);
  static_assert(offsetof(Tables, cons_items) == 0 && offsetof(Tables, der) == sizeof(int32_t) * 12,
                "obs_cell_emit's tables must lead Tables");
  __shared__ uint32_t list_n, rec_n, spill_n;
  const int a = (int)bid / p.n_agents, g = (int)bid % p.n_agents;
  const int tid = (int)threadIdx.x;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  SF_GLOBAL float *o = gptr(out) + (size_t)bid * SF_OBS_FLOATS;
  SF_GLOBAL f32x4 *o4 = reinterpret_cast<SF_GLOBAL f32x4 *>(o);  // 30752 floats = 7688 x 16 B, 16-B aligned
  // ---- prologue --------------------------------------------------------------------------------------------
  const uint32_t hf = gptr(p.hum)[((size_t)HW_FLAGS * p.A + a) * p.H + g];
  const uint32_t center = gptr(p.hum)[((size_t)HW_POS * p.A + a) * p.H + g];
  SF_GLOBAL uint32_t *old = nzprev ? gptr(nzprev) + (size_t)bid * OBS_W2 : nullptr;
  const bool redo = mode == 4;
  if (mode == 4) {  // sf_observe_overflow_device: the plain dense write, but only for the agents whose list did not fit
    const uint32_t c = gptr(sp.counts)[bid];
    if (!(c == 0xffffffffu || c > (uint32_t)sp.cap)) return;  // (uniform over the workgroup)
    mode = 0;
  }
  if (mode == 3 && (hf & (HF_ALIVE | HF_CTRL)) != (HF_ALIVE | HF_CTRL)) {  // no observer: an empty list
    if (tid == 0) gptr(sp.counts)[bid] = 0u;
    if (tid < 5 * SF_OBS_CHANNELS) gptr(sp.pov)[(size_t)bid * (5 * SF_OBS_CHANNELS) + tid] = 0.f;
    return;
  }
  if ((hf & (HF_ALIVE | HF_CTRL)) != (HF_ALIVE | HF_CTRL)) {  // no observer: all zero (uniform over the workgroup)
    for (int i = tid; i < SF_OBS_FLOATS / 4; i += OBS_THREADS)
      if (mode != 2 || ((old[(4u * (uint32_t)i) >> 5] >> ((4u * (uint32_t)i) & 31u)) & 15u))
        __builtin_nontemporal_store((f32x4)(0.f), &o4[i]);
    if (mode) {
      __syncthreads();  // every old word has been read
      for (int w = tid; w < OBS_W2; w += OBS_THREADS) old[w] = 0u;
    }
    return;
  }
  const bool zstage = p.Z <= OBS_Z_STAGE;  // a larger zombie table is read where it lies (flat loads through ObsView)
  const int nh = HW_WORDS * p.H, nz = zstage ? ZW_WORDS * p.Z : 0, nb = BW_WORDS * p.B;
  uint32_t *tab_lds = ent + nh + nz + nb;
  // Every global load of the prologue is issued before the first LDS store waits for one: the first 256 words of each
  // table and the window's four flag bytes per thread go to registers first.  (Loop by loop, each with its store behind
  // the load, this was a dozen round trips in a row — most of what a workgroup does before it starts to write.)
  auto hum_w = [&](int i) { return gptr(p.hum)[((size_t)(i / p.H) * p.A + a) * p.H + i % p.H]; };
  auto zom_w = [&](int i) { return gptr(p.zom)[((size_t)(i / p.Z) * p.A + a) * p.Z + i % p.Z]; };
  auto bul_w = [&](int i) { return gptr(p.bul)[((size_t)(i / p.B) * p.A + a) * p.B + i % p.B]; };
  auto tab_w = [&](int i) { return reinterpret_cast<const SF_GLOBAL uint32_t *>(gptr(p.tab))[i]; };
  const uint32_t rh = tid < nh ? hum_w(tid) : 0u, rz = tid < nz ? zom_w(tid) : 0u, rb = tid < nb ? bul_w(tid) : 0u;
  const uint32_t rt = tid < TAB_WORDS ? tab_w(tid) : 0u;
  const float r_in = tid < 16 ? gptr(p.tab)->obs_in[tid] : 0.f, r_out = tid < 16 ? gptr(p.tab)->obs_out[tid] : 0.f;
  const int t_n = gptr(p.tab)->obs_n;
  const int pteam = (int)((hf >> HF_TEAM_SH) & 255u);
  const int r0 = pos_r(center) - SF_OBS_WINDOW / 2, c0 = pos_c(center) - SF_OBS_WINDOW / 2, f0 = pos_f(center);
  constexpr int CELL_IT = (OBS_W2 + OBS_THREADS - 1) / OBS_THREADS;  // 4
  uint32_t rfl[CELL_IT];
#pragma unroll
  for (int q = 0; q < CELL_IT; ++q) {
    const int w = tid + q * OBS_THREADS;
    const int i = r0 + w / SF_OBS_WINDOW, j = c0 + w % SF_OBS_WINDOW;
    rfl[q] = (w < OBS_W2 && i >= 0 && j >= 0 && i < p.N && j < p.M) ? (uint32_t)gptr(p.flags)[(size_t)a * p.cells_pad + (size_t)(f0 * p.N + i) * p.M + j] : 0u;
  }
  if (tid < nh) ent[tid] = rh;
  if (tid < nz) ent[nh + tid] = rz;
  if (tid < nb) ent[nh + nz + tid] = rb;
  if (tid < TAB_WORDS) tab_lds[tid] = rt;
  if (tid < 16) t_in[tid] = r_in, t_out[tid] = r_out;
  if (tid == 0) list_n = 0u, rec_n = (uint32_t)OBS_CLASS_RECS, spill_n = 0u;
  for (int i = tid + OBS_THREADS; i < nh; i += OBS_THREADS) ent[i] = hum_w(i);  // (tables of more than 256 words)
  for (int i = tid + OBS_THREADS; i < nz; i += OBS_THREADS) ent[nh + i] = zom_w(i);
  for (int i = tid + OBS_THREADS; i < nb; i += OBS_THREADS) ent[nh + nz + i] = bul_w(i);
  for (int i = tid + OBS_THREADS; i < TAB_WORDS; i += OBS_THREADS) tab_lds[i] = tab_w(i);
#pragma unroll
  for (int q = 0; q < CELL_IT; ++q) {
    const int w = tid + q * OBS_THREADS;
    if (w >= OBS_W2) continue;
    const uint32_t fl = rfl[q];
    int32_t cdmg = 0;
    if (fl & SF_CELL_TEMP) {  // (a player-built object: on the map by construction)
      const int i = r0 + w / SF_OBS_WINDOW, j = c0 + w % SF_OBS_WINDOW;
      cdmg = gptr(p.aux_dmg)[(size_t)a * p.cells + (size_t)(f0 * p.N + i) * p.M + j];
    }
    occ[w] = 0u, wfl[w] = (uint8_t)fl, wdmg[w] = cdmg, slot[w] = (uint8_t)OBS_NOREC, nzmap[w] = 0u;
  }
  lds_barrier();
  // ---- pass 2 ----------------------------------------------------------------------------------------------
  ObsView v(p, 0);  // the LDS copy: one arena, [field][slot]
  v.hum_ = ent, v.bul_ = ent + nh + nz, v.A = 1;
  if (zstage)
    v.zom_ = ent + nh, v.zA = 1, v.za = 0;
  else
    v.zom_ = p.zom, v.zA = p.A, v.za = a;
  v.tab = reinterpret_cast<const Tables *>(tab_lds);
  for (int e = tid; e < p.H + p.Z + p.B; e += OBS_THREADS) {
    int s = -1;
    uint32_t bits = 0;
    if (e < p.H) {
      if (v.hum(HW_FLAGS, e) & HF_OCC) s = obs_window_slot(v.hum(HW_POS, e), center), bits = (uint32_t)(e + 1);
    } else if (e < p.H + p.Z) {
      const int z = e - p.H;
      const uint32_t zp = v.zom(ZW_POS, z);
      if (zp & ZF_ALIVE) s = obs_window_slot(zp & POS_MASK, center), bits = (uint32_t)(z + 1) << OCC_Z_SH;
    } else {
      const int b = e - p.H - p.Z;
      const uint32_t ba = v.bul(BW_A, b);
      if (ba & BA_REF) s = obs_window_slot(ba & POS_MASK, center), bits = (uint32_t)(b + 1) << OCC_B_SH;
    }
    if (s >= 0) atomicOr(&occ[s], bits);
  }
  lds_barrier();
  // ---- pass 3 ----------------------------------------------------------------------------------------------
  // items 0..7 are the shared class records (a pseudo-cell with that class's flag byte and nothing on it), items
  // 8.. are the window cells; one instantiation of obs_cell_emit serves both
  const Tables &tab = *p.tab;
  // 3a: every window cell is classified (empty / plain static cell -> shared class record / needs its own record);
  // the few cells that need a record are queued, so that the heavy feature code below runs once, densely, on
  // consecutive threads instead of once per wavefront per sweep of the window
  for (int w = tid; w < OBS_W2; w += OBS_THREADS) {
    const uint32_t fl = (uint32_t)wfl[w], oc = occ[w];
    if (fl == 0u && oc == 0u) continue;  // '.' with nothing on it
    const int cls = oc == 0u ? obs_class_of(fl) : -1;
    if (cls >= 0) {  // plain static cell: shared record, its non-zero bits are set after the barrier
      slot[w] = (uint8_t)cls;
      continue;
    }
    const uint32_t r = atomicAdd(&rec_n, 1u);
    if (r >= (uint32_t)OBS_REC_MAX) {
      atomicAdd(&spill_n, 1u);  // more non-empty cells than records: written after the stream, see below
      continue;
    }
    slot[w] = (uint8_t)r;
    work[r] = (uint16_t)w;
  }
  lds_barrier();
  // 3b: records 0..7 are the shared class records, finished on the host (Tables::class_rec); records 8.. belong to
  // the queued window cells and are built here
  rec[tid >> 5][tid & 31] = gptr(p.tab)->class_rec[tid >> 5][tid & 31];  // 8 x 32 = OBS_THREADS values
  if (tid < OBS_CLASS_RECS) cmask[tid] = gptr(p.tab)->class_mask[tid];
  const int n_items = (int)(rec_n < (uint32_t)OBS_REC_MAX ? rec_n : (uint32_t)OBS_REC_MAX);
  for (int item = OBS_CLASS_RECS + tid; item < n_items; item += OBS_THREADS) {
    const int w = (int)work[item];
    const uint32_t fl = (uint32_t)wfl[w], oc = occ[w];
    const uint32_t r = (uint32_t)item;
#pragma unroll
    for (int k = 0; k < SF_OBS_CHANNELS; ++k) rec[r][k] = 0.f;
    float ti[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ti[i] = i < t_n ? t_in[i] : __builtin_nanf("");
    uint32_t mask = 0u;
    obs_cell_emit(v, fl, wdmg[w], oc, pteam, [&](int k, float x) {
      // obs_map_fast() on the LDS copy of the constant table; 1.0 (a set flag) is most of what is emitted
      float y = 0.f;
      bool fast = x == 0.f;
      if (x == 1.f) {
        y = t_out[0], fast = true;
      } else if (!fast) {
        int hit = -1;  // the table's inputs are in registers (ti): a value that is not in it costs 15 compares, no LDS trip
#pragma unroll
        for (int i = 1; i < 16; ++i)
          if (x == ti[i]) hit = i;
        if (hit >= 0) y = t_out[hit], fast = true;
      }
      if (fast && y == 0.f) return;
      mask |= 1u << k;
      if (fast) {
        rec[r][k] = y;
      } else {
        const uint32_t q = atomicAdd(&list_n, 1u);
        if (q < (uint32_t)OBS_LIST_MAX)
          list_idx[q] = r * SF_OBS_CHANNELS + (uint32_t)k, list_val[q] = x;
        else
          rec[r][k] = obs_map(x);
      }
    });
    for (uint32_t m = mask; m; m &= m - 1u) {
      const uint32_t bit = (uint32_t)__builtin_ctz(m) * OBS_W2 + (uint32_t)w;
      atomicOr(&nzmap[bit >> 5], 1u << (bit & 31u));
    }
  }
  lds_barrier();
  for (int w = tid; w < OBS_W2; w += OBS_THREADS) {  // plain static cells: the class's non-zero channels
    const uint32_t sl = slot[w];
    if (sl >= (uint32_t)OBS_CLASS_RECS) continue;
    for (uint32_t m = cmask[sl]; m; m &= m - 1u) {
      const uint32_t bit = (uint32_t)__builtin_ctz(m) * OBS_W2 + (uint32_t)w;
      atomicOr(&nzmap[bit >> 5], 1u << (bit & 31u));
    }
  }
  // ---- pass 3b (same barrier interval: both only consume pass 3's results) -------------------------------------
  {
    const uint32_t n = list_n < (uint32_t)OBS_LIST_MAX ? list_n : (uint32_t)OBS_LIST_MAX;
    for (uint32_t i = (uint32_t)tid; i < n; i += OBS_THREADS) (&rec[0][0])[list_idx[i]] = obs_map(list_val[i]);
  }
  lds_barrier();
  // ---- pass 4, sparse form -----------------------------------------------------------------------------------
  if (mode == 3) {
    __shared__ uint32_t wave_tot[OBS_THREADS / 64];
    // thread t owns bitmap words 4 t .. 4 t + 3 (bits = dense indices in ascending order); an exclusive scan of the
    // threads' non-zero counts gives every entry its place in scan order
    uint32_t wv[4], cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = 4 * tid + j;
      wv[j] = w < OBS_W2 ? nzmap[w] : 0u;
      cnt += (uint32_t)__builtin_popcount(wv[j]);
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
      if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    lds_barrier();
    uint32_t pos = incl - cnt, total = 0;
#pragma unroll
    for (int q = 0; q < OBS_THREADS / 64; ++q) {
      if (q < (tid >> 6)) pos += wave_tot[q];
      total += wave_tot[q];
    }
    SF_GLOBAL uint32_t *kd = gptr(sp.keys) + (size_t)bid * (size_t)sp.cap;
    SF_GLOBAL float *vd = gptr(sp.vals) + (size_t)bid * (size_t)sp.cap;
    auto emit = [&](uint32_t e, uint32_t idx) {  // entry e of the list: dense index -> key, value
      const uint32_t k = idx / (uint32_t)OBS_W2, w = idx - k * (uint32_t)OBS_W2;
      const uint32_t y = w / (uint32_t)SF_OBS_WINDOW, x = w - y * (uint32_t)SF_OBS_WINDOW;
      if (e < (uint32_t)sp.cap) kd[e] = (k * 9u) | (y << 9) | (x << 14), vd[e] = rec[slot[w]][k];
    };
    // The non-zeros cluster (a channel that marks every wall cell fills whole bitmap words), so a thread that turned its
    // own bits into entries would make the others wait for the fullest words (13 k of the kernel's 42 k cycles).  The
    // threads only drop their bits' dense indices into an LDS list (occ[] and wdmg[] are dead by now), and the entries
    // are then built and stored round-robin: equal work, coalesced stores.
    constexpr uint32_t STAGED = 2u * (uint32_t)OBS_W2;
    if (total <= STAGED) {
      uint32_t *stage0 = occ, *stage1 = reinterpret_cast<uint32_t *>(wdmg);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        for (uint32_t m = wv[j]; m; m &= m - 1u) {
          const uint32_t idx = 32u * (uint32_t)(4 * tid + j) + (uint32_t)__builtin_ctz(m);
          (pos < (uint32_t)OBS_W2 ? stage0[pos] : stage1[pos - (uint32_t)OBS_W2]) = idx;
          ++pos;
        }
      lds_barrier();
      for (uint32_t e = (uint32_t)tid; e < total; e += OBS_THREADS)
        emit(e, e < (uint32_t)OBS_W2 ? stage0[e] : stage1[e - (uint32_t)OBS_W2]);
    } else {  // more non-zeros than the staging area holds (never an observation of the BASELINE configurations)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        for (uint32_t m = wv[j]; m; m &= m - 1u) {
          emit(pos, 32u * (uint32_t)(4 * tid + j) + (uint32_t)__builtin_ctz(m));
          ++pos;
        }
    }
    if (tid == 0) gptr(sp.counts)[bid] = spill_n ? 0xffffffffu : total;
    if (tid < 5 * SF_OBS_CHANNELS) {  // the network's pov: cells (-1,0) (0,-1) (0,0) (0,1) (1,0) around the centre, Modules.hpp:114-121
      const int cell = tid >> 5, ch = tid & 31;
      const int dy = (cell == 0) ? -1 : (cell == 4) ? 1 : 0, dx = (cell == 1) ? -1 : (cell == 3) ? 1 : 0;
      const uint32_t w = (uint32_t)((SF_OBS_WINDOW / 2 + dy) * SF_OBS_WINDOW + (SF_OBS_WINDOW / 2 + dx));
      const uint32_t bit = (uint32_t)ch * (uint32_t)OBS_W2 + w;
      gptr(sp.pov)[(size_t)bid * (5 * SF_OBS_CHANNELS) + tid] = ((nzmap[bit >> 5] >> (bit & 31u)) & 1u) ? rec[slot[w]][ch] : 0.f;
    }
    return;
  }
  // ---- pass 4 ----------------------------------------------------------------------------------------------
  const bool delta = mode == 2 && spill_n == 0u;  // (a window crowded beyond the records is written in full)
  if (delta) {  // the old map next to the new one in LDS (over wdmg): one coalesced read instead of one per piece
    for (int w = tid; w < OBS_W2; w += OBS_THREADS) ormap[w] = old[w];
    lds_barrier();
  }
#pragma unroll 2
  for (int i = tid; i < SF_OBS_FLOATS / 4; i += OBS_THREADS) {
    const uint32_t idx = 4u * (uint32_t)i;
    const uint32_t nib = (nzmap[idx >> 5] >> (idx & 31u)) & 15u;  // idx is a multiple of 4: a nibble never straddles
    if (delta && !(nib | ((ormap[idx >> 5] >> (idx & 31u)) & 15u))) continue;  // was zero, stays zero: not written
    f32x4 val = (f32x4)(0.f);
    if (nib) {  // ~5 % of the 16-B chunks
      uint32_t k = idx / (uint32_t)OBS_W2, w = idx - k * (uint32_t)OBS_W2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((nib >> j) & 1u) val[j] = rec[slot[w]][k];
        if (++w == (uint32_t)OBS_W2) w = 0u, ++k;
      }
    }
    __builtin_nontemporal_store(val, &o4[i]);  // streamed once, never re-read by this kernel
  }
  if (mode)  // what this buffer now holds; after a spill, "anything": the next delta call rewrites it all
    for (int w = tid; w < OBS_W2; w += OBS_THREADS) old[w] = spill_n ? 0xffffffffu : nzmap[w];
  if (spill_n) {  // a window crowded beyond OBS_REC_MAX cells (never in the BASELINE configs): direct, slower
    __syncthreads();  // the streamed zeros of those cells are complete before they are overwritten
    for (int w = tid; w < OBS_W2; w += OBS_THREADS) {
      const uint32_t fl = wfl[w], oc = occ[w];
      if ((fl == 0u && oc == 0u) || slot[w] != OBS_NOREC) continue;
      obs_cell_emit(v, fl, wdmg[w], oc, pteam, [&](int k, float x) {
        float y;
        if (!obs_map_fast(tab, x, y)) y = obs_map(x);
        if (y != 0.f) o[k * OBS_W2 + w] = y;
      });
    }
  }
  if (redo) {  // and the 160 centre values of such an agent, from the dense row just written (a crowded window's records
               // do not cover every cell): cell = t >> 5 of (-1,0) (0,-1) (0,0) (0,1) (1,0), channel = t & 31
    __syncthreads();  // (the workgroup's stores to the row are complete; this CU has not read the row before)
    if (tid < 5 * SF_OBS_CHANNELS) {
      const int cell = tid >> 5, ch = tid & 31;
      const int dy = (cell == 0) ? -1 : (cell == 4) ? 1 : 0, dx = (cell == 1) ? -1 : (cell == 3) ? 1 : 0;
      const int w = (SF_OBS_WINDOW / 2 + dy) * SF_OBS_WINDOW + (SF_OBS_WINDOW / 2 + dx);
      gptr(sp.pov)[(size_t)bid * (5 * SF_OBS_CHANNELS) + tid] = o[ch * OBS_W2 + w];
    }
  }
}

__global__ __launch_bounds__(OBS_THREADS, 6) void k_observe(Params p, float *out, uint32_t *nzprev, int mode, ObsSparse sp) {
  observe_agent(p, out, nzprev, mode, sp, blockIdx.x);
}
// sf_observe_overflow_device: a few workgroups walk the agents and redo (mode 4) those whose list did not fit — none, normally,
// and then the launch is one load per thread (as one workgroup per agent the idle launch cost 4 us of a 230 us loop).  A kernel
// of its own: two copies of the window code in one kernel spilled registers.
__global__ __launch_bounds__(OBS_THREADS) void k_observe_redo(Params p, float *out, ObsSparse sp) {
  const unsigned n = (unsigned)(p.A * p.n_agents);
  auto over = [&](unsigned b) {
    const uint32_t c = gptr(sp.counts)[b];
    return c == 0xffffffffu || c > (uint32_t)sp.cap;
  };
  bool mine = false;
  for (unsigned b = blockIdx.x + threadIdx.x * gridDim.x; b < n; b += OBS_THREADS * gridDim.x) mine = mine || over(b);
  if (!__syncthreads_or(mine)) return;
  for (unsigned b = blockIdx.x; b < n; b += gridDim.x) {
    if (!over(b)) continue;  // (uniform over the workgroup)
    observe_agent(p, out, nullptr, 4, sp, b);
    __syncthreads();  // the next agent takes over the workgroup's LDS
  }
}

// ---- sf_observe_sparse_device: the observation as the list of its non-zero floats, one WAVEFRONT per (arena, agent) ----
// The dense kernel above is built around streaming 123 KB per agent; for the list form that stream does not exist and
// what is left — 16 384 wavefronts for 4096 agents, five workgroup barriers around a few hundred useful operations per
// thread — takes 0.075 ms (round 3).  Here one 64-lane wavefront does
// the whole window, without workgroup barriers:
//   0  the window's 961 flag bytes are requested first, all sixteen loads of a lane in flight at once
//   1  every entity of the arena scatters itself into the window's occupant words (LDS atomics)
//   2a the window cells are classified, 64 per pass: empty / plain static cell (one of 8 shared records, finished on the
//      host) / needs a record of its own (an entity or a player-built object on it: ~20 cells) — those are queued
//   2b the queued cells' records are built, one lane each, by the same describe() code as everywhere else; values that
//      need a real x^(1/5) are queued once more and evaluated densely, one per lane (a human cell has ten of them)
//   3  the list leaves in the dense buffer's scan order — channel, then row, then column — which is the order
//      k_feat_list's partial sums are defined over: for every channel, the passes that hold a cell with that channel
//      set (a 16-bit set per channel, from one OR per pass) emit their entries at base + rank-below-me (ballot + mbcnt)
// Same values, same order, same counts as mode 3 of the dense kernel (tests/test_gpu_sparse_obs.py compares the list
// with the dense observation float by float).  A window with more than OL_REC own records is "crowded" (count
// 0xffffffff: the caller takes the dense call for that agent, as before; the dense kernel's own limit is 64).
// LDS: 9.7 KB per wavefront, so that the 16 wavefronts a CU gets of a 4096-agent launch are resident together.
constexpr int OL_REC = 48;                            // own records per window
constexpr int OL_STRIDE = SF_OBS_CHANNELS + 1;        // (odd stride: the lanes of pass 2b write different banks)
constexpr int OL_PASSES = (OBS_W2 + 63) / 64;         // 16
constexpr int OL_POWQ = 384;                          // queued x^(1/5) evaluations per window; more are done in place
constexpr int OL_CELLS = 640;                         // non-empty cells per window (walls included); more: "crowded"
static_assert(OL_REC + OBS_CLASS_RECS <= 64 && OBS_W2 <= 1024, "a compact cell entry is window cell | slot << 10 in 16 bits");
// LDS of one window, carved out of a caller-provided region (the stand-alone kernel's own, or k_step's dynamic region
// once the step has stored its state): 16-byte aligned, OL_LDS_BYTES long
struct ObsListLds {
  uint32_t occ_rec[OL_REC * OL_STRIDE];  // the occupant words (961) while cells are classified, then the records
  float crec[OBS_CLASS_RECS][SF_OBS_CHANNELS];
  uint32_t cmask[OBS_CLASS_RECS], recmask[OL_REC], work_oc[OL_REC], powq_n;
  uint16_t cell[OL_CELLS];               // the non-empty cells in window order: window cell | slot << 10
  uint16_t work_w[OL_REC], powq[OL_POWQ];
  uint8_t work_fl[OL_REC];
};
static_assert(OL_REC * OL_STRIDE >= OBS_W2 + 3, "the occupant words fit the record area");
constexpr size_t OL_LDS_BYTES = (sizeof(ObsListLds) + 15) & ~(size_t)15;
static_assert(OL_LDS_BYTES <= 10 * 1024, "16 wavefronts per CU");

#ifdef SF_DIAG_OBS  // diagnostic build only (tools/r04_obs_stamps.py): wave cycles per phase of the list observation
__device__ uint32_t sf_diag_obs[65536 * 8];  // [wave][phase]: the last launch's cycles (no atomics: they would be the measurement)
#define OL_STAMP(ph)                                                                                      \
  do {                                                                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
    if (threadIdx.x == 0 && blockIdx.x < 65536u) sf_diag_obs[blockIdx.x * 8u + (ph)] = (uint32_t)(t_ - ol_last_); \
    ol_last_ = t_;                                                                                        \
  } while (0)
#else
#define OL_STAMP(ph)
#endif

// bitwise OR over the wavefront's 64 lanes, as a wave-uniform value (DPP row steps, then the four row totals)
static __device__ __forceinline__ uint32_t wave_or(uint32_t x) {
  int v = (int)x;
  v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
  v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror: every lane holds its row's OR
  return (uint32_t)(__builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) |
                    __builtin_amdgcn_readlane(v, 48));
}
static __device__ __forceinline__ uint32_t rank_below(uint64_t bal) {  // set bits of `bal` below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
}

// The list observation of agent `agent` (= arena * n_agents + g) by the calling wavefront; L: OL_LDS_BYTES of LDS.
static __device__ __forceinline__ void observe_list_wave(const Params &p, const ObsSparse &sp, int agent, ObsListLds &L) {
#ifdef SF_DIAG_OBS
  unsigned long long ol_last_ = __builtin_amdgcn_s_memtime();
#endif
  uint32_t *occ = L.occ_rec;
  float *rec = reinterpret_cast<float *>(L.occ_rec);
  const int l = (int)threadIdx.x;
  const int a = agent / p.n_agents, g = agent % p.n_agents;
  const uint32_t hf = gptr(p.hum)[((size_t)HW_FLAGS * p.A + a) * p.H + g];
  const uint32_t center = gptr(p.hum)[((size_t)HW_POS * p.A + a) * p.H + g];
  SF_GLOBAL float *pov = gptr(sp.pov) + (size_t)agent * (5 * SF_OBS_CHANNELS);
  if ((hf & (HF_ALIVE | HF_CTRL)) != (HF_ALIVE | HF_CTRL)) {  // no observer: an empty list (uniform over the wave)
    if (l == 0) gptr(sp.counts)[agent] = 0u;
    for (int t = l; t < 5 * SF_OBS_CHANNELS; t += 64) pov[t] = 0.f;
    return;
  }
  auto crowded = [&]() {  // the caller takes the dense call for this agent
    if (l == 0) gptr(sp.counts)[agent] = 0xffffffffu;
    for (int t = l; t < 5 * SF_OBS_CHANNELS; t += 64) pov[t] = 0.f;  // (sf_observe_overflow_device rewrites it from the dense row)
  };
  // ---- 0: the window's flag bytes --------------------------------------------------------------------------------
  const int pteam = (int)((hf >> HF_TEAM_SH) & 255u);
  const int r0 = pos_r(center) - SF_OBS_WINDOW / 2, c0 = pos_c(center) - SF_OBS_WINDOW / 2, f0 = pos_f(center);
  const SF_GLOBAL uint8_t *plane = gptr(p.flags) + (size_t)a * p.cells_pad + (size_t)f0 * p.N * p.M;
  uint32_t flv[OL_PASSES];
#pragma unroll
  for (int it = 0; it < OL_PASSES; ++it) {
    const int w = it * 64 + l;
    const int i = r0 + w / SF_OBS_WINDOW, j = c0 + w % SF_OBS_WINDOW;
    flv[it] = (w < OBS_W2 && i >= 0 && j >= 0 && i < p.N && j < p.M) ? (uint32_t)plane[i * p.M + j] : 0u;
  }
  OL_STAMP(0);
  // ---- 1: occupant words -------------------------------------------------------------------------------------
  // (every global load of this phase first, into registers: class records, the constant table, the first 64 slots of each
  // entity table — written one by one, each with its LDS store or atomic behind it, they were ten round trips in a row)
  const ObsView v(p, a);  // entity tables where they lie: a window holds ~20 of them
  const uint32_t cm_ld = l < OBS_CLASS_RECS ? gptr(p.tab)->class_mask[l] : 0u;
  float cr_ld[OBS_CLASS_RECS * SF_OBS_CHANNELS / 64];
#pragma unroll
  for (int q = 0; q < OBS_CLASS_RECS * SF_OBS_CHANNELS / 64; ++q) {
    const int t = l + 64 * q;
    cr_ld[q] = gptr(p.tab)->class_rec[t >> 5][t & 31];
  }
  // the host-built constant table (the reference's libm): obs_map_fast.  One entry per lane, read by v_readlane below
  const float t_in = l < 16 ? gptr(p.tab)->obs_in[l] : 0.f, t_out = l < 16 ? gptr(p.tab)->obs_out[l] : 0.f;
  const int t_n = gptr(p.tab)->obs_n;
  // (large pools: only the words of the zombie table that are in use, sf_core.hpp ZL)
  int zlim = p.Z;
  if (large_pools(p.Z, p.P)) {
    const int used = 64 * (int)gptr(p.scal)[(size_t)a * SC_WORDS + SC_ZWN];
    zlim = used < p.Z ? used : p.Z;
  }
  const uint32_t h_fl = l < p.H ? v.hum(HW_FLAGS, l) : 0u, h_pos = l < p.H ? v.hum(HW_POS, l) : 0u;
  const uint32_t z_first = l < zlim ? v.zom(ZW_POS, l) : 0u, b_first = l < p.B ? v.bul(BW_A, l) : 0u;
  for (int w4 = l; w4 < (OBS_W2 + 3) / 4; w4 += 64) reinterpret_cast<u32x4 *>(occ)[w4] = (u32x4)(0u);
  if (l < OBS_CLASS_RECS) L.cmask[l] = cm_ld;
  if (l == 0) L.powq_n = 0u;
#pragma unroll
  for (int q = 0; q < OBS_CLASS_RECS * SF_OBS_CHANNELS / 64; ++q) {
    const int t = l + 64 * q;
    L.crec[t >> 5][t & 31] = cr_ld[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (h_fl & HF_OCC) {
    const int s = obs_window_slot(h_pos, center);
    if (s >= 0) atomicOr(&occ[s], (uint32_t)(l + 1));
  }
  for (int z = l; z < zlim; z += 64) {
    const uint32_t zp = z < 64 ? z_first : v.zom(ZW_POS, z);
    if (zp & ZF_ALIVE) {
      const int s = obs_window_slot(zp & POS_MASK, center);
      if (s >= 0) atomicOr(&occ[s], (uint32_t)(z + 1) << OCC_Z_SH);
    }
  }
  for (int b = l; b < p.B; b += 64) {
    const uint32_t ba = b < 64 ? b_first : v.bul(BW_A, b);
    if (ba & BA_REF) {
      const int s = obs_window_slot(ba & POS_MASK, center);
      if (s >= 0) atomicOr(&occ[s], (uint32_t)(b + 1) << OCC_B_SH);
    }
  }
  __syncthreads();
  OL_STAMP(1);
  // ---- 2a: classify; the non-empty cells are compacted, in window order ----------------------------------------
  uint32_t nrec = 0u, ncell = 0u;  // wave-uniform
#pragma unroll
  for (int it = 0; it < OL_PASSES; ++it) {
    const int w = it * 64 + l;
    const uint32_t fl = flv[it], oc = w < OBS_W2 ? occ[w] : 0u;
    const int cls = oc == 0u ? obs_class_of(fl) : -1;
    const bool some = fl != 0u || oc != 0u, own = some && cls < 0;
    const uint64_t bal = __builtin_amdgcn_ballot_w64(some), balo = __builtin_amdgcn_ballot_w64(own);
    if (some) {
      uint32_t s = (uint32_t)cls;
      if (own) {
        const uint32_t r = nrec + rank_below(balo);
        s = (uint32_t)OBS_CLASS_RECS + (r < (uint32_t)OL_REC ? r : 0u);
        if (r < (uint32_t)OL_REC) L.work_w[r] = (uint16_t)w, L.work_oc[r] = oc, L.work_fl[r] = (uint8_t)fl;
      }
      const uint32_t c = ncell + rank_below(bal);
      if (c < (uint32_t)OL_CELLS) L.cell[c] = (uint16_t)((uint32_t)w | (s << 10));
    }
    nrec += (uint32_t)__builtin_popcountll(balo), ncell += (uint32_t)__builtin_popcountll(bal);
  }
  if (nrec > (uint32_t)OL_REC || ncell > (uint32_t)OL_CELLS) return crowded();
  __syncthreads();  // every occupant word has been read: the records may overwrite them
  OL_STAMP(2);
  // ---- 2b: the queued cells' records, one lane each ------------------------------------------------------------
  const float t_out0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_out), 0));  // of 1.0: a set flag
  if ((uint32_t)l < nrec) {
    const int w = (int)L.work_w[l];
    const uint32_t fl = L.work_fl[l];
    int32_t cdmg = 0;
    if (fl & SF_CELL_TEMP) {  // (a player-built object: on the map by construction)
      const int i = r0 + w / SF_OBS_WINDOW, j = c0 + w % SF_OBS_WINDOW;
      cdmg = gptr(p.aux_dmg)[(size_t)a * p.cells + (size_t)(f0 * p.N + i) * p.M + j];
    }
    uint32_t m = 0u;
    obs_cell_emit(v, fl, cdmg, L.work_oc[l], pteam, [&](int k, float x) {
      if (x == 0.f) return;
      m |= 1u << k;
      if (x == 1.f) {  // (Tables::obs_in[0] == 1.0: most of what a cell emits)
        rec[l * OL_STRIDE + k] = t_out0;
        return;
      }
      // raw value now; the table / x^(1/5) pass below maps every queued one densely
      rec[l * OL_STRIDE + k] = x;
      const uint32_t q = atomicAdd(&L.powq_n, 1u);
      if (q < (uint32_t)OL_POWQ)
        L.powq[q] = (uint16_t)(l * OL_STRIDE + k);
      else
        rec[l * OL_STRIDE + k] = obs_map(x);  // (obs_map of a table entry is the table's value up to the host's libm: never reached in practice)
    });
    L.recmask[l] = m;
  }
  __syncthreads();
  OL_STAMP(3);
  {
    const uint32_t nq = L.powq_n < (uint32_t)OL_POWQ ? L.powq_n : (uint32_t)OL_POWQ;
    for (uint32_t q = (uint32_t)l; q < nq; q += 64u) {
      const uint32_t idx = L.powq[q];
      const float x = rec[idx];
      float y = 0.f;
      bool fast = false;
      for (int t = 1; t < t_n; ++t) {
        const float ti = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_in), t));
        const float to = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_out), t));
        if (x == ti) y = to, fast = true;
      }
      rec[idx] = fast ? y : obs_map(x);
    }
  }
  __syncthreads();
  OL_STAMP(4);
  // ---- 3: the list, in scan order: channel, then window cell ------------------------------------------------------
  auto mask_of = [&](uint32_t s) { return s < (uint32_t)OBS_CLASS_RECS ? L.cmask[s] : L.recmask[s - OBS_CLASS_RECS]; };
  auto value_of = [&](uint32_t s, uint32_t k) { return s < (uint32_t)OBS_CLASS_RECS ? L.crec[s][k] : rec[(s - OBS_CLASS_RECS) * OL_STRIDE + k]; };
  // the compact cells, 64 per pass: what a lane needs of its cell in each pass stays in registers for the whole channel
  // loop (record, channel mask, the key's position bits), and which channels a pass holds at all
  constexpr int CP_MAX = OL_CELLS / 64;  // 10
  const int npass = (int)((ncell + 63u) / 64u);
  uint32_t cpm = 0u;  // lane i < npass: the OR of pass i's channel masks
  uint32_t pm[CP_MAX], ps[CP_MAX], pkey[CP_MAX];
#pragma unroll
  for (int cp = 0; cp < CP_MAX; ++cp) {
    pm[cp] = 0u, ps[cp] = 0u, pkey[cp] = 0u;
    if (cp >= npass) continue;  // (uniform)
    const uint32_t c = (uint32_t)cp * 64u + (uint32_t)l;
    if (c < ncell) {
      const uint32_t ent = (uint32_t)L.cell[c], w = ent & 1023u;
      const uint32_t y = w / (uint32_t)SF_OBS_WINDOW, x = w - y * (uint32_t)SF_OBS_WINDOW;
      ps[cp] = ent >> 10, pm[cp] = mask_of(ps[cp]), pkey[cp] = (y << 9) | (x << 14);
    }
    const uint32_t o = wave_or(pm[cp]);
    cpm = l == cp ? o : cpm;
  }
  OL_STAMP(5);
  SF_GLOBAL uint32_t *kd = gptr(sp.keys) + (size_t)agent * (size_t)sp.cap;
  SF_GLOBAL float *vd = gptr(sp.vals) + (size_t)agent * (size_t)sp.cap;
  uint32_t base = 0u;
  for (uint32_t chans = wave_or(cpm); chans; chans &= chans - 1u) {  // the channels the window holds at all, ascending
    const uint32_t k = (uint32_t)__builtin_ctz(chans);
    const uint32_t passes = (uint32_t)__builtin_amdgcn_ballot_w64(((cpm >> k) & 1u) != 0u);  // which passes hold channel k
#pragma unroll
    for (int cp = 0; cp < CP_MAX; ++cp) {
      if (!((passes >> cp) & 1u)) continue;  // (uniform)
      const bool has = ((pm[cp] >> k) & 1u) != 0u;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(has);
      if (has) {
        const uint32_t e = base + rank_below(bal);
        if (e < (uint32_t)sp.cap) kd[e] = (k * 9u) | pkey[cp], vd[e] = value_of(ps[cp], k);
      }
      base += (uint32_t)__builtin_popcountll(bal);
    }
  }
  OL_STAMP(6);
  if (l == 0) gptr(sp.counts)[agent] = base;
  // the network's pov: cells (-1,0) (0,-1) (0,0) (0,1) (1,0) around the centre, channel fastest (Modules.hpp:114-121).
  // Their compact entries first (wave-uniform, by ballot over the passes), then 160 lanes' worth of values
  uint32_t pent[5] = {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu};
  for (int cp = 0; cp < npass; ++cp) {
    const uint32_t c = (uint32_t)cp * 64u + (uint32_t)l;
    const uint32_t ent = c < ncell ? (uint32_t)L.cell[c] : 0xffffu;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int dy = (q == 0) ? -1 : (q == 4) ? 1 : 0, dx = (q == 1) ? -1 : (q == 3) ? 1 : 0;
      const uint32_t w = (uint32_t)((SF_OBS_WINDOW / 2 + dy) * SF_OBS_WINDOW + (SF_OBS_WINDOW / 2 + dx));
      const uint64_t hit = __builtin_amdgcn_ballot_w64(c < ncell && (ent & 1023u) == w);
      if (hit) pent[q] = (uint32_t)__builtin_amdgcn_readlane((int)ent, __builtin_ctzll(hit));
    }
  }
  for (int t = l; t < 5 * SF_OBS_CHANNELS; t += 64) {
    const int q = t >> 5, ch = t & 31;
    const uint32_t ent = q == 0 ? pent[0] : q == 1 ? pent[1] : q == 2 ? pent[2] : q == 3 ? pent[3] : pent[4];
    const uint32_t s = ent >> 10;
    pov[t] = (ent != 0xffffu && ((mask_of(s) >> ch) & 1u)) ? value_of(s, (uint32_t)ch) : 0.f;
  }
  OL_STAMP(7);
}

__global__ __launch_bounds__(64) void k_observe_list(Params p, ObsSparse sp) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[OL_LDS_BYTES];
  observe_list_wave(p, sp, (int)blockIdx.x, *reinterpret_cast<ObsListLds *>(lds));
}

#define SF_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return fail(SF_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct HipRT {
  hipStream_t stream = nullptr;
  int device = 0;
  size_t lds_limit = 64 * 1024;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // one pair per timed step launch
  size_t used_events = 0;

  int init(int dev) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
      return fail(SF_ERR_DEVICE, "no HIP device: strikeforce_amd has no CPU path");
    if (dev < 0 || dev >= n) return fail(SF_ERR_DEVICE, "device ordinal out of range");
    device = dev;
    SF_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    SF_HIP(hipGetDeviceProperties(&prop, dev));
    lds_limit = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    return SF_OK;
  }
  void shutdown() {
    for (auto &ev : events) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
    events.clear();
  }
  size_t max_lds() const { return lds_limit; }
  void *alloc(size_t n) {
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr;
    return p;
  }
  void free(void *p) { (void)hipFree(p); }
  void h2d(void *d, const void *s, size_t n) { note(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream)); }
  void d2h(void *d, const void *s, size_t n) { note(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream)); }
  void d2d(void *d, const void *s, size_t n) { note(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream)); }
  void zero(void *d, size_t n) { note(hipMemsetAsync(d, 0, n, stream)); }
  // copies report through sync(): the first failure is kept and returned there
  hipError_t pending = hipSuccess;
  void note(hipError_t e) {
    if (e != hipSuccess && pending == hipSuccess) pending = e;
  }
  int sync() {
    if (pending != hipSuccess) {
      hipError_t e = pending;
      pending = hipSuccess;
      return fail(SF_ERR_DEVICE, std::string("async copy: ") + hipGetErrorString(e));
    }
    SF_HIP(hipStreamSynchronize(stream));
    return SF_OK;
  }

  template <class K>
  int lds_attr(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
      SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)bytes));
    return SF_OK;
  }

  // one launcher per kernel: picks the (HP, BM) variant by the map's size, sets the LDS attribute, launches
  template <int NB, bool ZL>
  int do_reset(const Params &p, const uint64_t *tb, const uint64_t *serial) {
    const size_t lds = lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P);
    int rc;
    if (!hbm_plane(p.cells_pad)) {
      if ((rc = lds_attr(k_reset<NB, false, true, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_reset<NB, false, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, tb, serial);
    } else if (use_bitmaps(p.cells_pad)) {
      if ((rc = lds_attr(k_reset<NB, true, true, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_reset<NB, true, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, tb, serial);
    } else {
      if ((rc = lds_attr(k_reset<NB, true, false, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_reset<NB, true, false, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, tb, serial);
    }
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_reset(const Params &p, int NB, const uint64_t *tb, const uint64_t *serial) {
    SF_HIP(hipSetDevice(device));
    if (large_pools(p.Z, p.P)) return do_reset<4, true>(p, tb, serial);
    switch (NB) {
      case 1: return do_reset<1, false>(p, tb, serial);
      case 2: return do_reset<2, false>(p, tb, serial);
      case 3: return do_reset<3, false>(p, tb, serial);
      default: return do_reset<4, false>(p, tb, serial);
    }
  }

  template <int NB, bool ZL>
  int do_step(const Params &p, const uint8_t *cmds, int k) {
    // SF_HBM_PLANE_K_MAX=k (A/B switch, default 0 = off): launches of at most k steps leave the flag plane in HBM (the
    // HBM-plane variant, whatever the map's size) instead of staging it — 8 KB less HBM traffic per arena of a one-step
    // launch of configs[2] (of 14), and measured slower: 0.0817 -> 0.0859 ms per step of the interactive loop, the step's
    // own cell reads then pay L2 latency one by one (tools/experiments/README.md)
    static const int hp_k_max = [] {
      const char *e = getenv("SF_HBM_PLANE_K_MAX");
      return e ? atoi(e) : 0;
    }();
    const bool hp = hbm_plane(p.cells_pad) || k <= hp_k_max, bm = use_bitmaps(p.cells_pad);
    const size_t lds = hbm_plane(p.cells_pad) || !hp ? lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P)
                                                     : lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P) - (size_t)p.cells_pad;
    int rc = !hp ? lds_attr(k_step<NB, false, true, ZL>, lds) : bm ? lds_attr(k_step<NB, true, true, ZL>, lds) : lds_attr(k_step<NB, true, false, ZL>, lds);
    if (rc) return rc;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (timing) {
      if (used_events == events.size()) {
        hipEvent_t a, b;
        SF_HIP(hipEventCreate(&a));
        SF_HIP(hipEventCreate(&b));
        events.emplace_back(a, b);
      }
      ev = &events[used_events++];
      SF_HIP(hipEventRecord(ev->first, stream));
    }
    if (rank_pending) {
      hipLaunchKernelGGL(k_rank, dim3(1), dim3(1024), 0, stream, p, rank_pending);
      rank_pending = nullptr;
      SF_HIP(hipGetLastError());
    }
    if (!hp)
      hipLaunchKernelGGL((k_step<NB, false, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, k);
    else if (bm)
      hipLaunchKernelGGL((k_step<NB, true, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, k);
    else
      hipLaunchKernelGGL((k_step<NB, true, false, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, k);
    SF_HIP(hipGetLastError());
    if (ev) SF_HIP(hipEventRecord(ev->second, stream));
    return SF_OK;
  }
  bool can_rank() const { return true; }
  // k_rank runs right in front of the k_step launch it orders, inside that launch's pair of timing events: what
  // sf_kernel_time reports (and bench.py's roofline prices) includes it
  uint32_t *rank_pending = nullptr;
  int launch_rank(const Params &, uint32_t *perm) {
    rank_pending = perm;
    return SF_OK;
  }
  int launch_step(const Params &p, int NB, const uint8_t *cmds, int k) {
    SF_HIP(hipSetDevice(device));
    if (large_pools(p.Z, p.P)) return do_step<4, true>(p, cmds, k);
    switch (NB) {
      case 1: return do_step<1, false>(p, cmds, k);
      case 2: return do_step<2, false>(p, cmds, k);
      case 3: return do_step<3, false>(p, cmds, k);
      default: return do_step<4, false>(p, cmds, k);
    }
  }
  template <int NB, bool ZL>
  int do_step_half(const Params &p, const uint8_t *cmds, int phase) {
    const bool hp = hbm_plane(p.cells_pad), bm = use_bitmaps(p.cells_pad);
    const size_t lds = lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P);
    int rc;
    if (!hp) {
      if ((rc = lds_attr(k_step_half<NB, false, true, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_step_half<NB, false, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, phase);
    } else if (bm) {
      if ((rc = lds_attr(k_step_half<NB, true, true, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_step_half<NB, true, true, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, phase);
    } else {
      if ((rc = lds_attr(k_step_half<NB, true, false, ZL>, lds))) return rc;
      hipLaunchKernelGGL((k_step_half<NB, true, false, ZL>), dim3((unsigned)p.A), dim3(64), lds, stream, p, cmds, phase);
    }
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_step_half(const Params &p, int NB, const uint8_t *cmds, int phase) {
    SF_HIP(hipSetDevice(device));
    if (large_pools(p.Z, p.P)) return do_step_half<4, true>(p, cmds, phase);
    switch (NB) {
      case 1: return do_step_half<1, false>(p, cmds, phase);
      case 2: return do_step_half<2, false>(p, cmds, phase);
      case 3: return do_step_half<3, false>(p, cmds, phase);
      default: return do_step_half<4, false>(p, cmds, phase);
    }
  }
  int launch_agent_alive(const Params &p, uint8_t *d_out) {
    SF_HIP(hipSetDevice(device));
    const int n = p.A * p.n_agents;
    hipLaunchKernelGGL(k_agent_alive, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, d_out);
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_done(const Params &p, uint8_t *d_out) {
    SF_HIP(hipSetDevice(device));
    const int n = p.A * p.n_agents;
    hipLaunchKernelGGL(k_done, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, d_out);
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_observe(const Params &p, int, float *out, uint32_t *nzprev, int mode) {
    SF_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_observe, dim3((unsigned)(p.A * p.n_agents)), dim3(OBS_THREADS),
                       obs_lds_bytes(p),
                       stream, p, out, nzprev, mode, ObsSparse{});
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_observe_sparse(const Params &p, uint32_t *keys, float *vals, uint32_t *counts, float *pov, int cap) {
    SF_HIP(hipSetDevice(device));
    static const bool old_form = getenv("SF_OBS_LIST_BLOCK") != nullptr;  // (A/B: the list as mode 3 of the dense kernel, round 3's form)
    if (!old_form)
      hipLaunchKernelGGL(k_observe_list, dim3((unsigned)(p.A * p.n_agents)), dim3(64), 0, stream, p, ObsSparse{keys, vals, counts, pov, cap});
    else
    hipLaunchKernelGGL(k_observe, dim3((unsigned)(p.A * p.n_agents)), dim3(OBS_THREADS),
                       obs_lds_bytes(p),
                       stream, p, (float *)nullptr, (uint32_t *)nullptr, 3, ObsSparse{keys, vals, counts, pov, cap});
    SF_HIP(hipGetLastError());
    return SF_OK;
  }

  int launch_observe_overflow(const Params &p, const uint32_t *counts, int cap, float *dense, float *pov) {
    SF_HIP(hipSetDevice(device));
    const int n = p.A * p.n_agents;
    hipLaunchKernelGGL(k_observe_redo, dim3((unsigned)(n < 512 ? n : 512)), dim3(OBS_THREADS),
                       obs_lds_bytes(p),
                       stream, p, dense,
                       ObsSparse{nullptr, nullptr, const_cast<uint32_t *>(counts), pov, cap});
    SF_HIP(hipGetLastError());
    return SF_OK;
  }

  int kernel_time(int enable, float *ms, int *launches) {
    SF_HIP(hipStreamSynchronize(stream));
    float total = 0.f;
    for (size_t i = 0; i < used_events; ++i) {
      float t = 0.f;
      SF_HIP(hipEventElapsedTime(&t, events[i].first, events[i].second));
      total += t;
    }
    if (ms) *ms = total;
    if (launches) *launches = (int)used_events;
    used_events = 0;
    timing = enable != 0;
    return SF_OK;
  }
};

// The one exchange step of the multi-GPU path: an RCCL all-gather of the end-of-episode result
// records, issued on a stream of the library's own so that it runs beside the next launch.  RCCL is loaded on first
// use (dlopen; single-GPU users never map it) and the communicator is the library's own: the caller only carries
// the 128-byte unique id from rank 0 to the other ranks (torch.distributed, MPI, a file: anything).
struct ncclUniqueIdBytes {  // ncclUniqueId of rccl.h: 128 opaque bytes, passed by value
  char internal[SF_COMM_ID_BYTES];
};
struct Comm {
  typedef int (*get_id_t)(void *);
  typedef int (*init_rank_t)(void **, int, ncclUniqueIdBytes, int);
  typedef int (*all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
  typedef int (*destroy_t)(void *);
  typedef const char *(*err_t)(int);
  typedef int (*count_t)(void *, int *);
  count_t count_fn = nullptr;
  void *dl = nullptr;
  get_id_t get_id = nullptr;
  init_rank_t init_rank = nullptr;
  all_gather_t all_gather = nullptr;
  destroy_t destroy_fn = nullptr;
  err_t err = nullptr;
  void *comm = nullptr;
  int world = 0, rank = 0;
  hipStream_t side = nullptr;
  hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
  int32_t *staging[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  unsigned issued = 0;

  int load() {
    if (dl) return SF_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!dl) return fail(SF_ERR_DEVICE, std::string("RCCL not found: ") + dlerror());
    get_id = (get_id_t)dlsym(dl, "ncclGetUniqueId");
    init_rank = (init_rank_t)dlsym(dl, "ncclCommInitRank");
    all_gather = (all_gather_t)dlsym(dl, "ncclAllGather");
    destroy_fn = (destroy_t)dlsym(dl, "ncclCommDestroy");
    err = (err_t)dlsym(dl, "ncclGetErrorString");
    count_fn = (count_t)dlsym(dl, "ncclCommCount");
    if (!get_id || !init_rank || !all_gather || !destroy_fn || !err) return fail(SF_ERR_DEVICE, "RCCL lacks a symbol");
    return SF_OK;
  }
  int nccl(int rc, const char *what) { return rc == 0 ? SF_OK : fail(SF_ERR_DEVICE, std::string(what) + ": " + err(rc)); }
  void shutdown() {
    if (comm) (void)destroy_fn(comm), comm = nullptr;
    for (int j = 0; j < 2; ++j) {
      if (ready[j]) (void)hipEventDestroy(ready[j]), ready[j] = nullptr;
      if (done[j]) (void)hipEventDestroy(done[j]), done[j] = nullptr;
      if (staging[j]) (void)hipFree(staging[j]), staging[j] = nullptr;
    }
    if (side) (void)hipStreamDestroy(side), side = nullptr;
  }
};

}  // namespace sf

struct sf_env {
  sf::Env<sf::HipRT> e;
  sf::Comm comm;
};

using sf::fail;  // SF_HIP in the entry points below

extern "C" {

int sf_abi_version(void) { return SF_ABI_VERSION; }
const char *sf_last_error(void) { return sf::last_error().c_str(); }
void sf_config_defaults(sf_config *cfg) {
  if (cfg) sf::config_defaults(cfg);
}

int sf_create(const sf_config *cfg, sf_env **out) {
  if (!out) return sf::fail(SF_ERR_ARG, "null output handle");
  *out = nullptr;
  sf_env *env = new (std::nothrow) sf_env();
  if (!env) return sf::fail(SF_ERR_MEMORY, "host allocation failed");
  int rc = env->e.create(cfg);
  if (rc != SF_OK) {
    env->e.destroy();
    delete env;
    return rc;
  }
  *out = env;
  return SF_OK;
}
int sf_destroy(sf_env *env) {
  if (!env) return SF_OK;
  env->comm.shutdown();
  env->e.destroy();
  delete env;
  return SF_OK;
}
#define SF_ENV(env) \
  if (!(env)) return sf::fail(SF_ERR_ARG, "null environment")

int sf_reset(sf_env *env, const uint64_t *tb, const uint64_t *serial) {
  SF_ENV(env);
  return env->e.reset(tb, serial);
}
int sf_step(sf_env *env, const uint8_t *cmd) {
  SF_ENV(env);
  return env->e.step_host(cmd);
}
int sf_step_device(sf_env *env, const uint8_t *d_cmd, int32_t k) {
  SF_ENV(env);
  return env->e.step_device(d_cmd, k);
}
int sf_observe(sf_env *env, float *out_host) {
  SF_ENV(env);
  return env->e.observe_host(out_host);
}
int sf_observe_device_delta(sf_env *env, float *d_out) {
  SF_ENV(env);
  return env->e.observe_device_delta(d_out);
}
int sf_observe_device(sf_env *env, float *d_out) {
  SF_ENV(env);
  return env->e.observe_device(d_out);
}
int sf_observe_overflow_device(sf_env *env, const uint32_t *d_counts, int32_t cap, float *d_dense, float *d_pov) {
  SF_ENV(env);
  return env->e.observe_overflow_device(d_counts, cap, d_dense, d_pov);
}
int sf_observe_sparse_device(sf_env *env, uint32_t *d_keys, float *d_vals, uint32_t *d_counts, float *d_pov, int32_t cap) {
  SF_ENV(env);
  return env->e.observe_sparse_device(d_keys, d_vals, d_counts, d_pov, cap);
}
int sf_results(sf_env *env, int32_t *out_host) {
  SF_ENV(env);
  return env->e.results_host(out_host);
}
int sf_results_device(sf_env *env, int32_t *d_out) {
  SF_ENV(env);
  return env->e.results_device(d_out);
}
int sf_done_device(sf_env *env, uint8_t *d_out) {
  SF_ENV(env);
  return env->e.done_device(d_out);
}
int sf_done_view_device(sf_env *env, const int32_t **d_words, int32_t *stride_words, int32_t *agents_per_arena) {
  SF_ENV(env);
  return env->e.done_view_device(d_words, stride_words, agents_per_arena);
}
int sf_done(sf_env *env, uint8_t *out_host) {
  SF_ENV(env);
  return env->e.done_host(out_host);
}
int sf_phase_draws(sf_env *env, int32_t *out_host) {
  SF_ENV(env);
  return env->e.phase_draws_host(out_host);
}
int sf_step_begin(sf_env *env) {
  SF_ENV(env);
  return env->e.step_begin();
}
int sf_step_end(sf_env *env, const uint8_t *cmd) {
  SF_ENV(env);
  return env->e.step_end_host(cmd);
}
int sf_step_end_device(sf_env *env, const uint8_t *d_cmd) {
  SF_ENV(env);
  return env->e.step_end_device(d_cmd);
}
int sf_agent_alive(sf_env *env, uint8_t *out_host) {
  SF_ENV(env);
  return env->e.agent_alive_host(out_host);
}
int sf_agent_alive_device(sf_env *env, uint8_t *d_out) {
  SF_ENV(env);
  return env->e.agent_alive_device(d_out);
}
int sf_state_digest(sf_env *env, uint64_t *out_host) {
  SF_ENV(env);
  return env->e.state_digest(out_host);
}
int sf_dump_arena(sf_env *env, int32_t arena, sf_arena_hdr *hdr, sf_human_rec *humans, sf_zombie_rec *zombies,
                  sf_bullet_rec *bullets, sf_portal_rec *portals, uint8_t *cell_flags, int32_t *cell_dmg,
                  int32_t *cell_portal) {
  SF_ENV(env);
  return env->e.dump_arena(arena, hdr, humans, zombies, bullets, portals, cell_flags, cell_dmg, cell_portal);
}
int sf_set_stream(sf_env *env, void *hip_stream) {
  SF_ENV(env);
  env->e.rt.stream = reinterpret_cast<hipStream_t>(hip_stream);
  return SF_OK;
}
int sf_synchronize(sf_env *env) {
  SF_ENV(env);
  return env->e.rt.sync();
}
int sf_comm_unique_id(uint8_t *id) {
  if (!id) return sf::fail(SF_ERR_ARG, "null id buffer");
  sf::Comm c;
  int rc = c.load();
  if (rc) return rc;
  return c.nccl(c.get_id(id), "ncclGetUniqueId");  // (the handle of a loaded library is reference-counted by dlopen)
}
int sf_comm_init(sf_env *env, const uint8_t *id, int32_t rank, int32_t world) {
  SF_ENV(env);
  sf::Comm &c = env->comm;
  if (!id || world < 1 || rank < 0 || rank >= world) return sf::fail(SF_ERR_ARG, "bad communicator arguments");
  if (c.comm) return sf::fail(SF_ERR_ARG, "communicator already initialised");
  int rc = c.load();
  if (rc) return rc;
  SF_HIP(hipSetDevice(env->e.rt.device));
  sf::ncclUniqueIdBytes uid;
  memcpy(uid.internal, id, sizeof uid.internal);
  rc = c.nccl(c.init_rank(&c.comm, world, uid, rank), "ncclCommInitRank");
  if (rc) return rc;
  c.world = world, c.rank = rank;
  // a failure from here on leaves no half-built communicator behind: a retry starts from scratch
  bool ok = hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) == hipSuccess;
  const size_t bytes = (size_t)env->e.p.A * env->e.p.n_agents * 8 * sizeof(int32_t);
  for (int j = 0; j < 2 && ok; ++j)
    ok = hipEventCreateWithFlags(&c.ready[j], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c.done[j], hipEventDisableTiming) == hipSuccess &&
         hipMalloc((void **)&c.staging[j], bytes) == hipSuccess;
  if (!ok) {
    c.shutdown();
    c.used[0] = c.used[1] = false, c.issued = 0;
    return sf::fail(SF_ERR_DEVICE, "sf_comm_init: stream / event / staging allocation failed");
  }
  return SF_OK;
}
int sf_comm_ranks(sf_env *env, int32_t *ranks) {
  SF_ENV(env);
  sf::Comm &c = env->comm;
  if (!ranks) return sf::fail(SF_ERR_ARG, "null output");
  if (!c.comm) return sf::fail(SF_ERR_ARG, "sf_comm_init has not been called");
  if (!c.count_fn) return sf::fail(SF_ERR_DEVICE, "RCCL lacks ncclCommCount");
  int n = 0;
  int rc = c.nccl(c.count_fn(c.comm, &n), "ncclCommCount");
  if (rc) return rc;
  *ranks = n;
  return SF_OK;
}
int sf_results_allgather(sf_env *env, int32_t *d_out) {
  SF_ENV(env);
  sf::Comm &c = env->comm;
  if (!c.comm) return sf::fail(SF_ERR_ARG, "sf_comm_init has not been called");
  if (!d_out) return sf::fail(SF_ERR_ARG, "null gather buffer");
  if (!env->e.was_reset) return sf::fail(SF_ERR_ARG, "sf_reset has not been called");
  if (int rc0 = env->e.not_mid_step("sf_results_allgather")) return rc0;
  SF_HIP(hipSetDevice(env->e.rt.device));
  hipStream_t st = env->e.rt.stream;
  const size_t count = (size_t)env->e.p.A * env->e.p.n_agents * 8;
  const int j = (int)(c.issued++ & 1u);
  // the records are snapshotted on the simulation stream (later launches latch new ones), the gather of the snapshot
  // runs on the side stream; a snapshot buffer is reused only after the gather that read it has finished
  if (c.used[j]) SF_HIP(hipStreamWaitEvent(st, c.done[j], 0));
  SF_HIP(hipMemcpyAsync(c.staging[j], env->e.p.results, count * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  SF_HIP(hipEventRecord(c.ready[j], st));
  SF_HIP(hipStreamWaitEvent(c.side, c.ready[j], 0));
  int rc = c.nccl(c.all_gather(c.staging[j], d_out, count, 2 /* ncclInt32 */, c.comm, c.side), "ncclAllGather");
  if (rc) return rc;
  SF_HIP(hipEventRecord(c.done[j], c.side));
  c.used[j] = true;
  return SF_OK;
}
int sf_comm_wait(sf_env *env, int32_t host_too) {
  SF_ENV(env);
  sf::Comm &c = env->comm;
  if (!c.comm) return sf::fail(SF_ERR_ARG, "sf_comm_init has not been called");
  SF_HIP(hipSetDevice(env->e.rt.device));
  for (int j = 0; j < 2; ++j)
    if (c.used[j]) SF_HIP(hipStreamWaitEvent(env->e.rt.stream, c.done[j], 0));
  if (host_too) SF_HIP(hipStreamSynchronize(c.side));
  return SF_OK;
}
#ifdef SF_DIAG_OBS
int sf_diag_obs_read(sf_env *env, uint32_t *out, int32_t waves) {  // diagnostic build only: [waves][8] cycles of the last launch
  SF_ENV(env);
  if (env->e.rt.sync() != SF_OK) return SF_ERR_DEVICE;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sf::sf_diag_obs), (size_t)waves * 8 * sizeof(uint32_t)) == hipSuccess ? SF_OK : SF_ERR_DEVICE;
}
#endif
#ifdef SF_DIAG_STAMPS
int sf_diag_times_read(sf_env *env, unsigned long long *out_host, int32_t workgroups) {  // diagnostic build only (tools/r04_k1_times.py)
  SF_ENV(env);
  if (env->e.rt.sync() != SF_OK) return SF_ERR_DEVICE;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(sf::sf_diag_times), (size_t)workgroups * 4 * sizeof(unsigned long long)) == hipSuccess
             ? SF_OK : SF_ERR_DEVICE;
}
int sf_diag_read(sf_env *env, uint32_t *out_host, int32_t arenas) {  // diagnostic build only (tools/diag_stamps.sh)
  SF_ENV(env);
  if (env->e.rt.sync() != SF_OK) return SF_ERR_DEVICE;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(sf::sf_diag_buffer), (size_t)arenas * 16 * sizeof(uint32_t)) == hipSuccess
             ? SF_OK : SF_ERR_DEVICE;
}
#endif
int sf_kernel_time(sf_env *env, int32_t enable, float *ms, int32_t *launches) {
  SF_ENV(env);
  return env->e.rt.kernel_time(enable, ms, launches);
}

}  // extern "C"
