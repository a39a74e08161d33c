// TEST INFRASTRUCTURE ONLY: the entry points of include/strikeforce.h that a host program needs (create / reset / step /
// split step / observe / done / agent_alive / digest), implemented over the device core running on the CPU wave emulator
// (sf_emu.cpp), so that C++ host code written against the C-ABI (include/sf_agent_adapter.hpp, examples/) can be
// exercised here, without a GPU, beside the reference.  Never shipped, never loaded by the product: libstrikeforce_amd.so
// has no CPU path and fails with SF_ERR_DEVICE when there is no GPU.
#include "sf_emu.cpp"

struct sf_env {
  sf::Env<sf::CpuRT> e;
};

extern "C" {
int sf_abi_version(void) { return SF_ABI_VERSION; }
const char *sf_last_error(void) { return sf::last_error().c_str(); }
void sf_config_defaults(sf_config *cfg) {
  if (cfg) sf::config_defaults(cfg);
}
int sf_create(const sf_config *cfg, sf_env **out) {
  if (!out) return SF_ERR_ARG;
  sf_env *env = new sf_env();
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
  if (env) env->e.destroy(), delete env;
  return SF_OK;
}
int sf_reset(sf_env *env, const uint64_t *tb, const uint64_t *serial) { return env->e.reset(tb, serial); }
int sf_step(sf_env *env, const uint8_t *cmd) { return env->e.step_host(cmd); }
int sf_step_begin(sf_env *env) { return env->e.step_begin(); }
int sf_step_end(sf_env *env, const uint8_t *cmd) { return env->e.step_end_host(cmd); }
int sf_observe(sf_env *env, float *out) { return env->e.observe_host(out); }
int sf_done(sf_env *env, uint8_t *out) { return env->e.done_host(out); }
int sf_agent_alive(sf_env *env, uint8_t *out) { return env->e.agent_alive_host(out); }
int sf_results(sf_env *env, int32_t *out) { return env->e.results_host(out); }
int sf_state_digest(sf_env *env, uint64_t *out) { return env->e.state_digest(out); }
int sf_phase_draws(sf_env *env, int32_t *out) { return env->e.phase_draws_host(out); }
int sf_dump_arena(sf_env *env, int32_t a, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs, sf_bullet_rec *bs,
                  sf_portal_rec *ps, uint8_t *cf, int32_t *cd, int32_t *cp) {
  return env->e.dump_arena(a, hdr, hs, zs, bs, ps, cf, cd, cp);
}
}
