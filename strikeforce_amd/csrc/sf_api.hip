// sf_api.hip — libstrikeforce_amd.so: gfx950 kernels + the C-ABI of include/strikeforce.h.
//
// Kernels
//   k_reset<NB>   one wavefront per arena: setup()/load_data()/_srand + first loop top
//   k_step<NB>    one wavefront per arena: K iterations of the gameplay loop per launch, state in
//                 registers, the arena's cell-flag plane in LDS
//   k_observe     one 256-thread workgroup per (arena, agent): 32 x 31 x 31 float observation
// There is no CPU path: without a HIP device every entry point fails with SF_ERR_DEVICE.
#include <hip/hip_runtime.h>

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

template <int NB>
__global__ __launch_bounds__(64) void k_reset(Params p, const uint64_t *tb, const uint64_t *serial) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  Core<WaveGfx950, NB>::reset_body(lds, p, (int)blockIdx.x, tb, serial);
}

template <int NB>
__global__ __launch_bounds__(64) void k_step(Params p, const uint8_t *cmds, int k) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  Core<WaveGfx950, NB>::step_body(lds, p, (int)blockIdx.x, cmds, k);
}

constexpr int OBS_W2 = SF_OBS_WINDOW * SF_OBS_WINDOW;  // 961
constexpr int OBS_THREADS = 256;

__global__ __launch_bounds__(OBS_THREADS) void k_observe(Params p, float *out) {
  __shared__ uint32_t occ[OBS_W2];
  __shared__ int32_t wdmg[OBS_W2];
  __shared__ uint8_t wfl[OBS_W2 + 3];
  const int a = (int)blockIdx.x / p.n_agents, g = (int)blockIdx.x % p.n_agents;
  const int tid = (int)threadIdx.x;
  float *o = out + (size_t)blockIdx.x * SF_OBS_FLOATS;
  const ObsView v(p, a);
  const uint32_t hf = v.hum(HW_FLAGS, g);
  if ((hf & (HF_ALIVE | HF_CTRL)) != (HF_ALIVE | HF_CTRL)) {  // no observer: all zero
    float4 *o4 = reinterpret_cast<float4 *>(o);                // 30752 floats = 7688 float4, 16-B aligned
    for (int i = tid; i < SF_OBS_FLOATS / 4; i += OBS_THREADS) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const uint32_t center = v.hum(HW_POS, g);
  const int pteam = (int)((hf >> HF_TEAM_SH) & 255u);
  const int r0 = pos_r(center) - SF_OBS_WINDOW / 2, c0 = pos_c(center) - SF_OBS_WINDOW / 2, f0 = pos_f(center);
  for (int w = tid; w < OBS_W2; w += OBS_THREADS) {
    const int i = r0 + w / SF_OBS_WINDOW, j = c0 + w % SF_OBS_WINDOW;
    uint32_t fl = 0;
    int32_t cdmg = 0;
    if (i >= 0 && j >= 0 && i < p.N && j < p.M) {
      const size_t ci = (size_t)(f0 * p.N + i) * p.M + j;
      fl = p.flags[(size_t)a * p.cells_pad + ci];
      if (fl & SF_CELL_TEMP) cdmg = p.aux_dmg[(size_t)a * p.cells + ci];
    }
    occ[w] = 0u, wfl[w] = (uint8_t)fl, wdmg[w] = cdmg;
  }
  __syncthreads();
  for (int e = tid; e < p.H + p.Z + p.B; e += OBS_THREADS) {
    int s = -1;
    uint32_t bits = 0;
    if (e < p.H) {
      if (v.hum(HW_FLAGS, e) & HF_OCC) s = obs_window_slot(v.hum(HW_POS, e), center), bits = (uint32_t)(e + 1);
    } else if (e < p.H + p.Z) {
      const int z = e - p.H;
      const uint32_t zp = v.zom(ZW_POS, z);
      if (zp & ZF_ALIVE) s = obs_window_slot(zp & POS_MASK, center), bits = (uint32_t)(z + 1) << 8;
    } else {
      const int b = e - p.H - p.Z;
      const uint32_t ba = v.bul(BW_A, b);
      if (ba & BA_REF) s = obs_window_slot(ba & POS_MASK, center), bits = (uint32_t)(b + 1) << 16;
    }
    if (s >= 0) atomicOr(&occ[s], bits);
  }
  __syncthreads();
  for (int idx = tid; idx < SF_OBS_FLOATS; idx += OBS_THREADS) {
    const int k = idx / OBS_W2, w = idx - k * OBS_W2;
    o[idx] = obs_map(obs_feature(v, k, wfl[w], wdmg[w], occ[w], pteam));
  }
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

  template <int NB>
  int do_reset(const Params &p, const uint64_t *tb, const uint64_t *serial) {
    int rc = lds_attr(k_reset<NB>, lds_bytes_for(p.cells_pad));
    if (rc) return rc;
    hipLaunchKernelGGL(k_reset<NB>, dim3((unsigned)p.A), dim3(64), lds_bytes_for(p.cells_pad), stream, p, tb, serial);
    SF_HIP(hipGetLastError());
    return SF_OK;
  }
  int launch_reset(const Params &p, int NB, const uint64_t *tb, const uint64_t *serial) {
    SF_HIP(hipSetDevice(device));
    switch (NB) {
      case 1: return do_reset<1>(p, tb, serial);
      case 2: return do_reset<2>(p, tb, serial);
      case 3: return do_reset<3>(p, tb, serial);
      default: return do_reset<4>(p, tb, serial);
    }
  }

  template <int NB>
  int do_step(const Params &p, const uint8_t *cmds, int k) {
    int rc = lds_attr(k_step<NB>, lds_bytes_for(p.cells_pad));
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
    hipLaunchKernelGGL(k_step<NB>, dim3((unsigned)p.A), dim3(64), lds_bytes_for(p.cells_pad), stream, p, cmds, k);
    SF_HIP(hipGetLastError());
    if (ev) SF_HIP(hipEventRecord(ev->second, stream));
    return SF_OK;
  }
  int launch_step(const Params &p, int NB, const uint8_t *cmds, int k) {
    SF_HIP(hipSetDevice(device));
    switch (NB) {
      case 1: return do_step<1>(p, cmds, k);
      case 2: return do_step<2>(p, cmds, k);
      case 3: return do_step<3>(p, cmds, k);
      default: return do_step<4>(p, cmds, k);
    }
  }
  int launch_observe(const Params &p, int, float *out) {
    SF_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_observe, dim3((unsigned)(p.A * p.n_agents)), dim3(OBS_THREADS), 0, stream, p, out);
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

}  // namespace sf

struct sf_env {
  sf::Env<sf::HipRT> e;
};

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
int sf_observe_device(sf_env *env, float *d_out) {
  SF_ENV(env);
  return env->e.observe_device(d_out);
}
int sf_results(sf_env *env, int32_t *out_host) {
  SF_ENV(env);
  return env->e.results_host(out_host);
}
int sf_results_device(sf_env *env, int32_t *d_out) {
  SF_ENV(env);
  return env->e.results_device(d_out);
}
int sf_done(sf_env *env, uint8_t *out_host) {
  SF_ENV(env);
  return env->e.done_host(out_host);
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
int sf_kernel_time(sf_env *env, int32_t enable, float *ms, int32_t *launches) {
  SF_ENV(env);
  return env->e.rt.kernel_time(enable, ms, launches);
}

}  // extern "C"
