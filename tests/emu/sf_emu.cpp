// sf_emu.cpp — TEST INFRASTRUCTURE: the device core (strikeforce_amd/csrc/sf_core.hpp) and the host API
// (sf_host.hpp) compiled for the CPU against a 64-lane wave emulator.  Exports the C-ABI of
// include/strikeforce.h with an `sfe_` prefix.  Loaded only by tests/ (CPU parity of the kernel logic
// against the oracle, ASan/UBSan runs); never part of, or a fallback for, the product library.
#include <stdlib.h>

#include <vector>

#include "wave_emu.hpp"
// clang-format off
#include "../../strikeforce_amd/csrc/sf_core.hpp"
#include "../../strikeforce_amd/csrc/sf_obs.hpp"
#include "../../strikeforce_amd/csrc/sf_host.hpp"
// clang-format on

namespace sf {

// same variant choice as the HIP launchers: big flag planes stay in "HBM" (here: the host arrays)
template <int NB, bool ZL>
static void run_reset(const Params &p, const uint64_t *tb, const uint64_t *serial) {
  std::vector<uint8_t> lds(lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P));
  for (int a = 0; a < p.A; ++a) {
    if (!hbm_plane(p.cells_pad))
      Core<WaveEmu, NB, false, true, ZL>::reset_body(lds.data(), p, a, tb, serial);
    else if (use_bitmaps(p.cells_pad))
      Core<WaveEmu, NB, true, true, ZL>::reset_body(lds.data(), p, a, tb, serial);
    else
      Core<WaveEmu, NB, true, false, ZL>::reset_body(lds.data(), p, a, tb, serial);
  }
}
template <int NB, bool ZL>
static void run_step(const Params &p, const uint8_t *cmds, int k) {
  std::vector<uint8_t> lds(lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P));
  for (int a = 0; a < p.A; ++a) {
    if (!hbm_plane(p.cells_pad))
      Core<WaveEmu, NB, false, true, ZL>::step_body(lds.data(), p, a, cmds, k);
    else if (use_bitmaps(p.cells_pad))
      Core<WaveEmu, NB, true, true, ZL>::step_body(lds.data(), p, a, cmds, k);
    else
      Core<WaveEmu, NB, true, false, ZL>::step_body(lds.data(), p, a, cmds, k);
  }
}

template <int NB, bool ZL>
static void run_step_half(const Params &p, const uint8_t *cmds, int phase) {
  std::vector<uint8_t> lds(lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P));
  for (int a = 0; a < p.A; ++a) {
    if (!hbm_plane(p.cells_pad))
      Core<WaveEmu, NB, false, true, ZL>::step_half_body(lds.data(), p, a, cmds, phase);
    else if (use_bitmaps(p.cells_pad))
      Core<WaveEmu, NB, true, true, ZL>::step_half_body(lds.data(), p, a, cmds, phase);
    else
      Core<WaveEmu, NB, true, false, ZL>::step_half_body(lds.data(), p, a, cmds, phase);
  }
}

static void run_observe(const Params &p, float *out) {
  const int W2 = SF_OBS_WINDOW * SF_OBS_WINDOW;
  std::vector<uint32_t> occ(W2);
  for (int a = 0; a < p.A; ++a)
    for (int g = 0; g < p.n_agents; ++g) {
      float *o = out + ((size_t)a * p.n_agents + g) * SF_OBS_FLOATS;
      ObsView v(p, a);
      const uint32_t hf = v.hum(HW_FLAGS, g);
      if ((hf & (HF_ALIVE | HF_CTRL)) != (HF_ALIVE | HF_CTRL)) {
        for (int i = 0; i < SF_OBS_FLOATS; ++i) o[i] = 0.f;
        continue;
      }
      const uint32_t center = v.hum(HW_POS, g);
      const int pteam = (int)((hf >> HF_TEAM_SH) & 255u);
      for (int w = 0; w < W2; ++w) occ[w] = 0;
      for (int h = 0; h < p.H; ++h)
        if (v.hum(HW_FLAGS, h) & HF_OCC) {
          int s = obs_window_slot(v.hum(HW_POS, h), center);
          if (s >= 0) occ[s] |= (uint32_t)(h + 1);
        }
      for (int z = 0; z < p.Z; ++z)
        if (v.zom(ZW_POS, z) & ZF_ALIVE) {
          int s = obs_window_slot(v.zom(ZW_POS, z) & POS_MASK, center);
          if (s >= 0) occ[s] |= (uint32_t)(z + 1) << OCC_Z_SH;
        }
      for (int b = 0; b < p.B; ++b)
        if (v.bul(BW_A, b) & BA_REF) {
          int s = obs_window_slot(v.bul(BW_A, b) & POS_MASK, center);
          if (s >= 0) occ[s] |= (uint32_t)(b + 1) << OCC_B_SH;
        }
      for (int w = 0; w < W2; ++w) {
        const int i = pos_r(center) - SF_OBS_WINDOW / 2 + w / SF_OBS_WINDOW;
        const int j = pos_c(center) - SF_OBS_WINDOW / 2 + w % SF_OBS_WINDOW;
        uint32_t fl = 0;
        int32_t cdmg = 0;
        if (i >= 0 && j >= 0 && i < p.N && j < p.M) {
          const size_t ci = (size_t)(pos_f(center) * p.N + i) * p.M + j;
          fl = p.flags[(size_t)a * p.cells_pad + ci];
          if (fl & SF_CELL_TEMP) cdmg = p.aux_dmg[(size_t)a * p.cells + ci];
        }
        for (int k = 0; k < SF_OBS_CHANNELS; ++k) o[k * W2 + w] = 0.f;
        obs_cell_emit(v, fl, cdmg, occ[w], pteam, [&](int k, float x) {
          float y;
          if (!obs_map_fast(*p.tab, x, y)) y = obs_map(x);
          o[k * W2 + w] = y;
        });
      }
    }
}

struct CpuRT {
  int init(int) { return SF_OK; }
  void shutdown() {}
  size_t max_lds() const { return 160 * 1024; }
  void *alloc(size_t n) { return calloc(n ? n : 1, 1); }
  void free(void *p) { ::free(p); }
  void h2d(void *d, const void *s, size_t n) { memcpy(d, s, n); }
  void d2h(void *d, const void *s, size_t n) { memcpy(d, s, n); }
  void d2d(void *d, const void *s, size_t n) { memcpy(d, s, n); }
  void zero(void *d, size_t n) { memset(d, 0, n); }
  int sync() { return SF_OK; }
  int launch_reset(const Params &p, int NB, const uint64_t *tb, const uint64_t *serial) {
    if (large_pools(p.Z, p.P)) {
      run_reset<4, true>(p, tb, serial);
      return SF_OK;
    }
    switch (NB) {
      case 1: run_reset<1, false>(p, tb, serial); break;
      case 2: run_reset<2, false>(p, tb, serial); break;
      case 3: run_reset<3, false>(p, tb, serial); break;
      default: run_reset<4, false>(p, tb, serial); break;
    }
    return SF_OK;
  }
  int launch_step(const Params &p, int NB, const uint8_t *cmds, int k) {
    if (large_pools(p.Z, p.P)) {
      run_step<4, true>(p, cmds, k);
      return SF_OK;
    }
    switch (NB) {
      case 1: run_step<1, false>(p, cmds, k); break;
      case 2: run_step<2, false>(p, cmds, k); break;
      case 3: run_step<3, false>(p, cmds, k); break;
      default: run_step<4, false>(p, cmds, k); break;
    }
    return SF_OK;
  }
  bool can_rank() const { return false; }  // (a launch order only matters where arenas run side by side)
  int launch_rank(const Params &, uint32_t *) { return SF_OK; }
  int launch_step_half(const Params &p, int NB, const uint8_t *cmds, int phase) {
    if (large_pools(p.Z, p.P)) {
      run_step_half<4, true>(p, cmds, phase);
      return SF_OK;
    }
    switch (NB) {
      case 1: run_step_half<1, false>(p, cmds, phase); break;
      case 2: run_step_half<2, false>(p, cmds, phase); break;
      case 3: run_step_half<3, false>(p, cmds, phase); break;
      default: run_step_half<4, false>(p, cmds, phase); break;
    }
    return SF_OK;
  }
  int launch_agent_alive(const Params &p, uint8_t *out) {
    for (int i = 0; i < p.A * p.n_agents; ++i) {
      const int a = i / p.n_agents, g = i % p.n_agents;
      const uint32_t fl = p.hum[((size_t)HW_FLAGS * (size_t)p.A + (size_t)a) * (size_t)p.H + (size_t)g];
      out[i] = (uint8_t)((fl & (HF_ALIVE | HF_CTRL)) == (HF_ALIVE | HF_CTRL));
    }
    return SF_OK;
  }
  int launch_done(const Params &p, uint8_t *out) {
    for (int i = 0; i < p.A * p.n_agents; ++i) {
      const int32_t *sc = p.scal + (size_t)(i / p.n_agents) * SC_WORDS;
      out[i] = (uint8_t)(p.auto_reset ? sc[SC_ENDED] : sc[SC_DONE]);
    }
    return SF_OK;
  }
  int launch_observe_overflow(const Params &, const uint32_t *, int, float *, float *) { return SF_ERR_DEVICE; }  // device-only form
  int launch_observe_sparse(const Params &, uint32_t *, float *, uint32_t *, float *, int) { return SF_ERR_DEVICE; }
  int launch_observe(const Params &p, int, float *out, uint32_t *, int) {  // (delta mode: same final buffer content)
    run_observe(p, out);
    return SF_OK;
  }
};

}  // namespace sf

struct sfe_env {
  sf::Env<sf::CpuRT> e;
};

extern "C" {
// op profile of the device source: out[0] = all vector ops, out[1 + ph] = vector ops in phase ph, out[1 + PH_COUNT + ph]
// = wave-uniform accesses (readlane / ballot / setlane / uniform LDS) in phase ph; clears the counters (out: 32 words)
int sfe_profile(uint64_t *out) {
  sf::EmuProf &q = sf::emu_prof();
  q.by[q.phase] += q.ops - q.mark, q.mark = q.ops;
  out[0] = q.ops;
  q.sby[q.phase] += q.sops - q.smark, q.smark = q.sops;
  for (int i = 0; i < sf::PH_COUNT; ++i) out[1 + i] = q.by[i], q.by[i] = 0;
  for (int i = 0; i < sf::PH_COUNT; ++i) out[1 + sf::PH_COUNT + i] = q.sby[i], q.sby[i] = 0;
  q.ops = q.mark = 0, q.sops = q.smark = 0;
  return sf::PH_COUNT;
}
// SF_COUNT(k) path counters of the device source; clears them (out: 8 words)
void sfe_counts(uint64_t *out) {
  sf::EmuProf &q = sf::emu_prof();
  for (int i = 0; i < 8; ++i) out[i] = q.counts[i], q.counts[i] = 0;
}
sfe_env *sfe_create(const sf_config *cfg) {
  sfe_env *env = new sfe_env();
  if (env->e.create(cfg) != SF_OK) {
    env->e.destroy();
    delete env;
    return nullptr;
  }
  return env;
}
int sfe_destroy(sfe_env *env) {
  if (env) {
    env->e.destroy();
    delete env;
  }
  return SF_OK;
}
int sfe_reset(sfe_env *env, const uint64_t *tb, const uint64_t *serial) { return env->e.reset(tb, serial); }
int sfe_step(sfe_env *env, const uint8_t *cmd) { return env->e.step_host(cmd); }
int sfe_step_many(sfe_env *env, const uint8_t *cmds, int32_t k) { return env->e.step_device(cmds, k); }
int sfe_step_begin(sfe_env *env) { return env->e.step_begin(); }
int sfe_step_end(sfe_env *env, const uint8_t *cmd) { return env->e.step_end_host(cmd); }
int sfe_agent_alive(sfe_env *env, uint8_t *out) { return env->e.agent_alive_host(out); }
int sfe_observe(sfe_env *env, float *out) { return env->e.observe_host(out); }
int sfe_results(sfe_env *env, int32_t *out) { return env->e.results_host(out); }
int sfe_done(sfe_env *env, uint8_t *out) { return env->e.done_host(out); }
int sfe_phase_draws(sfe_env *env, int32_t *out) { return env->e.phase_draws_host(out); }
int sfe_state_digest(sfe_env *env, uint64_t *out) { return env->e.state_digest(out); }
int sfe_dump_arena(sfe_env *env, int32_t a, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs, sf_bullet_rec *bs,
                   sf_portal_rec *ps, uint8_t *cf, int32_t *cd, int32_t *cp) {
  return env->e.dump_arena(a, hdr, hs, zs, bs, ps, cf, cd, cp);
}
const char *sfe_last_error(void) { return sf::last_error().c_str(); }
}
