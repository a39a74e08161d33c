// sf_host.hpp — host side of the C-ABI (include/strikeforce.h): configuration checks, the derived
// stat tables (Human::build), HBM allocation, launches and the parity tooling (dump / digest).
//
// Template over a runtime RT that owns memory and launches.  The product instantiates it with the HIP
// runtime (sf_api.hip); tests/emu instantiates it with a CPU runtime that runs the same device core on a
// wave emulator.  There is no CPU path in the product: HipRT fails with SF_ERR_DEVICE without a GPU.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/strikeforce.h"
#include "sf_types.hpp"
#include "sf_obs.hpp"

namespace sf {

inline std::string &last_error() {
  static thread_local std::string e;
  return e;
}
inline int fail(int code, const std::string &msg) {
  last_error() = msg;
  return code;
}

// Character.hpp:29-45 compute_damage (only ever called on per-profile constants, so it runs on the host
// once and the device reads Derived::cd_punch / cd_weapon).  Note the reference's indentation: only
// `tmp /= mid` belongs to the for.
inline int host_compute_damage(int x, int y) {
  int l = 0, r = x + 1, z = 2;
  for (; 1 < y; y >>= 1) ++z;
  while (r - l > 1) {
    const int mid = (l + r) >> 1;
    int tmp = x;
    for (int i = 0; i < z && mid; ++i) tmp /= mid;
    if (tmp)
      l = mid;
    else
      r = mid;
  }
  return l;
}

// Human::build, Character.hpp:650-709: what a freshly built human of this profile carries.
inline void derive_profile(const sf_config &cfg, const sf_profile &pr, Derived &d) {
  int def_blocks = 8, def_portals = 1;  // Character.hpp:78-79
  int md = pr.mindamage_def;
  d.hp = pr.def_hp, d.mindamage = pr.mindamage_def, d.stamina = pr.def_stamina;  // Character.hpp:667
  for (int i = 0; i < 4; ++i) d.cons[i] = pr.cons[i];
  for (int i = 0; i < 4; ++i) {
    const int lvl = pr.throw_lvl_cnt[i][0];
    const int up = lvl - 1 > 0 ? lvl - 1 : 0;  // Character.hpp:674-675
    d.thr_cnt[i] = pr.throw_lvl_cnt[i][1];
    d.thr[i][0] = cfg.items.thr[i][0];
    d.thr[i][1] = cfg.items.thr[i][1] + 50 * up;  // Weapon::upgrade Item.hpp:105-111
    d.thr[i][2] = cfg.items.thr[i][2] - 50 * up;
    d.thr[i][3] = cfg.items.thr[i][3];
  }
  for (int i = 0; i < 8; ++i) {
    const int lvl = pr.weapon_lvl[i] > 0 ? pr.weapon_lvl[i] : 0;  // Character.hpp:680-681
    d.weapon_lvl[i] = pr.weapon_lvl[i];
    d.weapon[i][0] = cfg.items.weapon[i][0];
    d.weapon[i][1] = cfg.items.weapon[i][1] + 50 * lvl;
    d.weapon[i][2] = cfg.items.weapon[i][2] - 50 * lvl;
    d.weapon[i][3] = cfg.items.weapon[i][3];
  }
  const int lv[3] = {pr.level_solo, pr.level_timer, pr.level_squad};  // Character.hpp:689-706
  for (int m = 0; m < 3; ++m)
    for (int level = 2; level <= lv[m]; ++level) {  // level_*_up Character.hpp:765-807
      md += 5;
      if (level % 2 == 1) ++def_blocks, ++def_portals;
    }
  d.mindamage_def = md;
  d.blocks = def_blocks, d.portals = def_portals;  // back_tmp Character.hpp:139-144
}

inline void finish_derived(Derived &d) {
  d.cd_punch = host_compute_damage(d.mindamage_def, 1);
  for (int i = 0; i < 8; ++i) d.cd_weapon[i] = host_compute_damage(d.weapon[i][1], d.weapon[i][3]);
}

// Tables::hatab (sf_types.hpp HT_*): obey()'s key classes (gameplay.hpp:695-821) and the per-profile stats its
// shooting / selection branches read, as one flat table that the step kernel keeps in LDS
inline void fill_hatab(Tables &t, int blocks) {
  memset(t.hatab, 0, sizeof t.hatab);
  uint8_t *cmd = reinterpret_cast<uint8_t *>(t.hatab + HT_CMD);
  auto put = [&](char c, int cls, int prm) { cmd[(unsigned char)c] = (uint8_t)(cls | (prm << 4)); };
  put('_', CL_SUICIDE, 0), put('[', CL_BLOCK, 0), put(']', CL_PORTAL, 0), put('q', CL_TURN, 0), put('e', CL_TURN, 1);
  put('s', CL_MOVE, 0), put('d', CL_MOVE, 1), put('w', CL_MOVE, 2), put('a', CL_MOVE, 3);
  const char *selc = "fghj", *selt = "kl;'", *selw = "cvbnm,./";
  for (int k = 0; k < 4; ++k) put(selc[k], CL_SELC, k), put(selt[k], CL_SELT, k);
  for (int k = 0; k < 8; ++k) put(selw[k], CL_SELW, k);
  put('u', CL_USE, 0), put('z', CL_PUNCH, 0), put('x', CL_FIRE, 0);
  for (int pr = 0; pr < blocks; ++pr) {
    uint32_t *w = t.hatab + HT_PROF + pr * HT_PROF_STRIDE;
    const Derived &d = t.der[pr];
    w[HT_P_CDPUNCH] = (uint32_t)d.cd_punch;
    for (int k = 0; k < 8; ++k) {
      w[HT_P_WLVL + k] = (uint32_t)d.weapon_lvl[k], w[HT_P_CDW + k] = (uint32_t)d.cd_weapon[k];
      for (int j = 0; j < 4; ++j) w[HT_P_WEAPON + 4 * k + j] = (uint32_t)d.weapon[k][j];
    }
    for (int k = 0; k < 4; ++k)
      for (int j = 0; j < 4; ++j) w[HT_P_THR + 4 * k + j] = (uint32_t)d.thr[k][j];
  }
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 3; ++j) t.hatab[HT_CONS + 3 * k + j] = (uint32_t)t.cons_items[k][j];
}

inline int validate(const sf_config *c) {
  if (!c) return fail(SF_ERR_ARG, "null config");
  if (c->abi_version != SF_ABI_VERSION) return fail(SF_ERR_ARG, "abi_version mismatch");
  if (!c->map) return fail(SF_ERR_ARG, "config.map is null");
  if (c->arenas < 1) return fail(SF_ERR_ARG, "arenas < 1");
  if (c->floors < 1 || c->floors > 4) return fail(SF_ERR_ARG, "floors must be 1..4");
  if (c->rows < 3 || c->cols < 3 || c->rows > SF_MAX_COORD || c->cols > SF_MAX_COORD)
    return fail(SF_ERR_ARG, "rows/cols must be 3..1024");
  if (c->cap_humans < 1 || c->cap_humans > SF_MAX_HUMANS) return fail(SF_ERR_ARG, "cap_humans must be 1..64");
  if (c->cap_zombies < 1 || c->cap_zombies > SF_MAX_ZOMBIES) return fail(SF_ERR_ARG, "cap_zombies must be 1..9000");
  if (c->cap_bullets < 1 || c->cap_bullets > SF_MAX_BULLETS) return fail(SF_ERR_ARG, "cap_bullets must be 1..256");
  if (c->cap_portals < 1 || c->cap_portals > SF_MAX_PORTALS) return fail(SF_ERR_ARG, "cap_portals must be 1..9000");
  if (c->cap_chests < 0) return fail(SF_ERR_ARG, "cap_chests < 0");
  if (c->reseed_stride < 0) return fail(SF_ERR_ARG, "reseed_stride < 0");
  if (c->ind < 0 || c->ind >= c->n_agents || (c->ind != 0 && c->mode != SF_MODE_BATTLE))
    return fail(SF_ERR_ARG, "ind must be 0, or an agent slot of a Battle match");
  if (c->n_agents < 1 || c->n_agents > SF_MAX_AGENTS || c->n_agents > c->cap_humans)
    return fail(SF_ERR_ARG, "n_agents must be 1..min(16, cap_humans)");
  if (c->level < 1 || c->level > 10) return fail(SF_ERR_ARG, "level must be 1..10");
  if (c->mode < SF_MODE_SOLO || c->mode > SF_MODE_BATTLE) return fail(SF_ERR_ARG, "unknown mode");
  if (c->mode == SF_MODE_SQUAD && (c->cap_humans < 10 || c->rows < 5 || c->cols < 12))
    return fail(SF_ERR_ARG, "Squad needs cap_humans >= 10 and a map of at least 5 x 12");
  if ((c->mode == SF_MODE_SOLO || c->mode == SF_MODE_TIMER) && c->n_agents != 1)
    return fail(SF_ERR_ARG, "Solo/Timer have exactly one agent");
  if (c->n_agent_profiles != 0 && c->n_agent_profiles != c->n_agents)
    return fail(SF_ERR_ARG, "n_agent_profiles must be 0 or n_agents");
  // per-player records are what the players of a lock-step match exchange (gameplay.hpp:120-151): Battle mode only.  In
  // Solo / Timer / Squad every commanded human is built from `player` (Squad team mates from `npc`, gameplay.hpp:1878)
  if (c->n_agent_profiles != 0 && c->mode != SF_MODE_BATTLE)
    return fail(SF_ERR_ARG, "agent_profile[] is for SF_MODE_BATTLE (one record per player of the match)");
  const sf_profile *pr[2 + SF_MAX_AGENTS] = {&c->player, &c->npc};
  for (int i = 0; i < c->n_agent_profiles; ++i) pr[2 + i] = &c->agent_profile[i];
  for (int k = 0; k < 2 + c->n_agent_profiles; ++k) {
    for (int i = 0; i < 4; ++i)
      if (pr[k]->cons[i] < 0 || pr[k]->cons[i] > 65535 || pr[k]->throw_lvl_cnt[i][1] < 0 ||
          pr[k]->throw_lvl_cnt[i][1] > 65535)
        return fail(SF_ERR_ARG, "item counts must be 0..65535");
  }
  for (int i = 0; i < 4; ++i)
    if (c->items.thr[i][3] < 0 || c->items.thr[i][3] > 65535) return fail(SF_ERR_ARG, "item range out of 0..65535");
  for (int i = 0; i < 8; ++i)
    if (c->items.weapon[i][3] < 0 || c->items.weapon[i][3] > 65535) return fail(SF_ERR_ARG, "item range out of 0..65535");
  const int cells = c->floors * c->rows * c->cols;
  for (int i = 0; i < cells; ++i) {
    const char ch = c->map[i];
    if (ch != '#' && ch != '.' && ch != 'O' && ch != '^' && ch != 'v') return fail(SF_ERR_ARG, "map holds a char outside {# . O ^ v}");
    if ((ch == '^' || ch == 'v') && (!c->map_portal || c->map_portal[i] < 0 || c->map_portal[i] >= c->cap_portals))
      return fail(SF_ERR_ARG, "portal entrance without a valid exit number in map_portal");
  }
  return SF_OK;
}

// splitmix64 finalizer; the digest is an order-independent sum of per-item hashes
inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
inline uint64_t dg(uint64_t tag, uint64_t idx, int64_t v) {
  return mix64((tag << 56) ^ (idx << 32) ^ (uint64_t)(uint32_t)v ^ ((uint64_t)(v >> 32) << 40));
}

template <class RT>
struct Env {
  sf_config cfg;
  Params p;
  Tables tab;
  int NB = 1, cells = 0;
  bool was_reset = false;
  bool mid_step = false;  // between sf_step_begin and sf_step_end
  RT rt;
  // device buffers
  Tables *d_tab = nullptr;
  uint8_t *d_map_flags = nullptr;
  int16_t *d_map_pidx = nullptr;
  uint32_t *d_map_exits = nullptr;
  uint16_t *d_logt = nullptr;
  uint32_t *d_exptab = nullptr;
  uint64_t *d_tb = nullptr, *d_serial = nullptr;
  uint8_t *d_cmd = nullptr;
  uint32_t *d_perm = nullptr;  // k_step's launch order (k_rank): arenas by population, for long launches
  int balance = 1;             // SF_BALANCE=0 switches the ordering off (A/B measurements)
  int rank_k_min = 8;          // launches of fewer steps run in arena order
  int rank_every = 100;        // steps between two orderings (SF_RANK_EVERY; measured: tools/experiments/README.md)
  int steps_since_rank = 1 << 30;  // the order is renewed every >= 100 steps (populations change by one every 20-25 steps)
  float *d_obs = nullptr;
  uint32_t *d_nzprev = nullptr;      // [A * n_agents][961] which floats of the delta-tracked buffer are non-zero
  const float *delta_ptr = nullptr;  // the buffer d_nzprev describes
  // host copies of the static map
  std::vector<uint8_t> map_flags;
  std::vector<int16_t> map_pidx;
  std::vector<uint32_t> map_exits;
  // host snapshot of the dynamic state (dump / digest)
  std::vector<uint32_t> s_hum, s_zom, s_bul, s_por, s_rng;
  std::vector<int32_t> s_scal, s_dmg;
  std::vector<int16_t> s_pidx;
  std::vector<uint8_t> s_flags;

  template <class T>
  int alloc(T *&ptr, size_t n) {
    ptr = (T *)rt.alloc(n * sizeof(T));
    return ptr ? SF_OK : fail(SF_ERR_MEMORY, "device allocation failed");
  }

  int create(const sf_config *c) {
    int rc = validate(c);
    if (rc) return rc;
    cfg = *c;
    rc = rt.init(cfg.device);
    if (rc) return rc;
    memset(&p, 0, sizeof p);
    memset(&tab, 0, sizeof tab);
    cells = cfg.floors * cfg.rows * cfg.cols;
    p.A = cfg.arenas, p.F = cfg.floors, p.N = cfg.rows, p.M = cfg.cols;
    p.cells = cells, p.cells_pad = (cells + 15) & ~15;
    p.bm_words = bm_words_for(p.cells_pad);
    p.H = cfg.cap_humans, p.Z = cfg.cap_zombies, p.B = cfg.cap_bullets, p.P = cfg.cap_portals, p.C = cfg.cap_chests;
    p.mode = cfg.mode, p.level = cfg.level, p.n_agents = cfg.n_agents, p.auto_reset = cfg.auto_reset;
    p.reseed = cfg.reseed_stride > 0 ? cfg.reseed_stride : cfg.arenas;
    p.timer_lim = cfg.level * (cfg.timer_frames_per_level > 0 ? cfg.timer_frames_per_level : 7500);
    p.squad_floor = cfg.floors > 2 ? 2 : cfg.floors - 1;
    p.ind = cfg.ind;
    // large pools (zombie / exit tables in LDS): one kernel instance, built for four bullet words (any B up to 256)
    NB = large_pools(p.Z, p.P) ? 4 : nb_for(p.B);
    {
      const char *e = getenv("SF_BALANCE");  // (A/B measurements: SF_BALANCE=0 switches k_rank's launch order off)
      balance = (e && e[0] == '0') ? 0 : 1;
      const char *r = getenv("SF_RANK_K_MIN");  // (A/B measurements: the shortest launch that is ordered)
      if (r) rank_k_min = atoi(r);
      const char *ev = getenv("SF_RANK_EVERY");
      if (ev && atoi(ev) > 0) rank_every = atoi(ev);
    }
    // tables: one shared player record (block 0) + npc (block 1), or one record per commanded human (blocks 0..15,
    // the account blobs of a lock-step match, gameplay.hpp:120-151) + npc (block 16)
    p.npc_block = cfg.n_agent_profiles > 0 ? MAX_PROFILE_BLOCKS - 1 : 1;
    p.ht_bytes = ht_bytes_for(p.npc_block + 1), p.lds_tab = lds_tab_for(p.npc_block + 1);
    if (lds_bytes_for(p.cells_pad, p.lds_tab, p.Z, p.P) > rt.max_lds())
      return fail(SF_ERR_ARG, large_pools(p.Z, p.P) ? "flag plane + zombie / exit tables (12 B per zombie slot, 4 B per exit) do not fit the CU's LDS: lower cap_zombies / cap_portals"
                                                   : "map does not fit the LDS flag plane");
    for (int i = 0; i < p.npc_block; ++i)
      derive_profile(cfg, (cfg.n_agent_profiles > 0 && i < cfg.n_agent_profiles) ? cfg.agent_profile[i] : cfg.player, tab.der[i]);
    derive_profile(cfg, cfg.npc, tab.der[p.npc_block]);
    tab.der[p.npc_block].mindamage_def += 15 * (cfg.level - 1);  // gen_human Character.hpp:882-886
    for (int i = 0; i <= p.npc_block; ++i) finish_derived(tab.der[i]);
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 3; ++k) tab.cons_items[i][k] = cfg.items.cons[i][k];
    fill_hatab(tab, p.npc_block + 1);
    for (int i = 0; i < SF_MAX_AGENTS; ++i) {
      if (cfg.agent_team[i] < 0 || cfg.agent_team[i] > 255) return fail(SF_ERR_ARG, "agent_team must be 0..255");
      tab.teams[i] = cfg.agent_team[i];
    }
    {  // observation fast map (sf_obs.hpp obs_map_fast): constants of describe(), Custom.hpp:98,110-111,117-121
      float in[16];
      int n = 0;
      auto add = [&](float x) {
        if (x == 0.f) return;
        for (int i = 0; i < n; ++i)
          if (in[i] == x) return;
        if (n < 16) in[n++] = x;
      };
      add(1.0f), add((float)0.01), add((float)(20 / 1000.0)), add((float)(10 / 1000.0));
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) add((float)(cfg.items.cons[i][k] / 1000.0));
      tab.obs_n = n;
      for (int i = 0; i < n; ++i) {
        tab.obs_in[i] = in[i];
        tab.obs_out[i] = (float)pow((double)(fabsf(in[i]) / 10), 0.2);  // Custom.hpp:157, host libm
      }
      // the records of plain static cells, through the same describe() code the kernel runs for occupied cells
      Params hp = p;
      hp.tab = &tab;
      const ObsView hv(hp, 0);
      for (int c = 0; c < 8; ++c) {
        tab.class_mask[c] = 0u;
        for (int k = 0; k < 32; ++k) tab.class_rec[c][k] = 0.f;
        obs_cell_emit(hv, obs_class_flags(c), 0, 0u, 0, [&](int k, float x) {
          float y;
          if (!obs_map_fast(tab, x, y)) y = obs_map(x);
          tab.class_rec[c][k] = y;
          if (y != 0.f) tab.class_mask[c] |= 1u << k;
        });
      }
    }
    // static map: gameplay.hpp:1249-1274
    map_flags.assign((size_t)p.cells_pad, 0);
    map_pidx.assign((size_t)cells, -1);
    map_exits.assign((size_t)p.P, 0u);
    int next_exit = 0;
    for (int ci = 0; ci < cells; ++ci) {
      const char ch = cfg.map[ci];
      if (ch == '#')
        map_flags[ci] = SF_CELL_WALL;
      else if (ch == '^')
        map_flags[ci] = SF_CELL_PIN_UP, map_pidx[ci] = cfg.map_portal[ci];
      else if (ch == 'v')
        map_flags[ci] = SF_CELL_PIN_DN, map_pidx[ci] = cfg.map_portal[ci];
      else if (ch == 'O') {
        map_flags[ci] = SF_CELL_POUT;
        if (next_exit < p.P) {  // p_ind() of an empty table: scan order
          const int f = ci / (p.N * p.M), r = (ci / p.M) % p.N, cc = ci % p.M;
          map_exits[next_exit++] = pos_pack(f, r, cc) | PF_ACTIVE;
        }
      }
    }
    p.pw0 = zw_for(next_exit);
    cfg.map = nullptr, cfg.map_portal = nullptr;  // the caller's buffers are not kept
    // RNG tables: discrete logs / powers of the generator 3 of Z/65537*
    std::vector<uint16_t> logt(LOGT_ENTRIES, 0);
    std::vector<uint32_t> exptab(512);
    {
      uint32_t v = 1;
      for (uint32_t m = 0; m < 65536; ++m) {
        logt[LOGT_OFF + v] = (uint16_t)m;                                              // t = v
        if (v + (uint32_t)LOGT_OFF >= 65537u) logt[LOGT_OFF + v - 65537u] = (uint16_t)m;  // t = v - 65537 < 0
        if (m < 256) exptab[m] = v;
        if ((m & 255u) == 0) exptab[256 + (m >> 8)] = v;
        v = (uint32_t)(((uint64_t)v * 3u) % 65537u);
      }
    }
    // device state
    const size_t A = (size_t)p.A;
    if ((rc = alloc(d_logt, (size_t)LOGT_ENTRIES)) || (rc = alloc(d_exptab, 512)) || (rc = alloc(d_tab, 1)) || (rc = alloc(d_map_flags, (size_t)p.cells_pad)) || (rc = alloc(d_map_pidx, (size_t)cells)) ||
        (rc = alloc(d_map_exits, (size_t)p.P)) || (rc = alloc(p.hum, HW_WORDS * A * p.H)) ||
        (rc = alloc(p.zom, ZW_WORDS * A * p.Z)) || (rc = alloc(p.bul, BW_WORDS * A * p.B)) ||
        (rc = alloc(p.por, A * p.P)) || (rc = alloc(p.rng, A * RNG_WORDS)) || (rc = alloc(p.rng2, A * RNG_WORDS)) || (rc = alloc(p.scal, A * SC_WORDS)) ||
        (rc = alloc(p.results, A * p.n_agents * 8)) || (rc = alloc(p.flags, A * (size_t)p.cells_pad)) ||
        (rc = alloc(p.aux_dmg, A * (size_t)cells)) || (rc = alloc(p.aux_pidx, A * (size_t)cells)) ||
        (rc = alloc(d_tb, A)) || (rc = alloc(d_serial, A)) || (rc = alloc(d_cmd, A * p.n_agents)))
      return rc;
    rt.h2d(d_tab, &tab, sizeof tab);
    rt.h2d(d_logt, logt.data(), logt.size() * sizeof(uint16_t));
    rt.h2d(d_exptab, exptab.data(), exptab.size() * sizeof(uint32_t));
    if ((rc = rt.sync())) return rc;  // the staging vectors above die with this scope
    rt.h2d(d_map_flags, map_flags.data(), map_flags.size());
    rt.h2d(d_map_pidx, map_pidx.data(), map_pidx.size() * sizeof(int16_t));
    rt.h2d(d_map_exits, map_exits.data(), map_exits.size() * sizeof(uint32_t));
    rt.zero(p.results, A * p.n_agents * 8 * sizeof(int32_t));
    rt.zero(p.scal, A * SC_WORDS * sizeof(int32_t));
    p.logt = d_logt, p.exptab = d_exptab;
    p.tab = d_tab, p.map_flags = d_map_flags, p.map_pidx = d_map_pidx, p.map_exits = d_map_exits;
    return rt.sync();
  }

  void destroy() {
    void *ptrs[] = {d_logt, d_exptab, d_tab, d_map_flags, d_map_pidx, d_map_exits, p.hum, p.zom, p.bul, p.por, p.rng, p.rng2, p.scal, p.results,
                    p.flags, p.aux_dmg, p.aux_pidx, d_tb, d_serial, d_cmd, d_obs, d_nzprev, d_perm};
    for (void *q : ptrs)
      if (q) rt.free(q);
    rt.shutdown();
  }

  int reset(const uint64_t *tb, const uint64_t *serial) {
    if (!tb || !serial) return fail(SF_ERR_ARG, "null seed array");
    rt.h2d(d_tb, tb, sizeof(uint64_t) * (size_t)p.A);
    rt.h2d(d_serial, serial, sizeof(uint64_t) * (size_t)p.A);
    rt.zero(p.results, (size_t)p.A * p.n_agents * 8 * sizeof(int32_t));
    int rc = rt.launch_reset(p, NB, d_tb, d_serial);
    if (rc) return rc;
    was_reset = true;
    mid_step = false;
    steps_since_rank = 1 << 30;
    return rt.sync();
  }

  int step_host(const uint8_t *cmd) {
    if (!cmd) return fail(SF_ERR_ARG, "null command array");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_step before sf_reset");
    if (mid_step) return fail(SF_ERR_STATE, "sf_step between sf_step_begin and sf_step_end");
    rt.h2d(d_cmd, cmd, (size_t)p.A * p.n_agents);
    return rt.launch_step(p, NB, d_cmd, 1);
  }
  int step_device(const uint8_t *d_cmds, int k) {
    if (!d_cmds || k < 1) return fail(SF_ERR_ARG, "bad command buffer / step count");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_step_device before sf_reset");
    if (mid_step) return fail(SF_ERR_STATE, "sf_step_device between sf_step_begin and sf_step_end");
    // long launches: order the arenas by population first, so that the ones sharing a SIMD are of different loads (k_rank)
    Params q = p;
    // Below 1024 arenas every arena has a SIMD of its own (the chip has 1024): nothing to order.  Measured for 2048 ...
    // 16384 arenas, an arena count that is no multiple of 1024, and the maps whose flag plane stays in HBM
    // (profiles/r04b_rank_sweep.txt): +1 % (2048) ... +14 % (5000) on configs[2], +4 % on configs[4], -0.6 % on configs[3]
    // (every arena at its caps: nothing to order, k_rank's 7 us per 100 steps is what is left)
    if (balance && k >= rank_k_min && p.A >= 1024 && rt.can_rank()) {
      int rc;
      if (!d_perm && (rc = alloc(d_perm, (size_t)p.A))) return rc;
      if (steps_since_rank >= rank_every) {
        if ((rc = rt.launch_rank(p, d_perm))) return rc;
        steps_since_rank = 0;
      }
      steps_since_rank += k;
      q.perm = d_perm;
    }
    return rt.launch_step(q, NB, d_cmds, k);
  }
  // the iteration in two halves (the reference queries the agents of humans other than `ind` between them, G:988-999)
  int step_begin() {
    if (!was_reset) return fail(SF_ERR_STATE, "sf_step_begin before sf_reset");
    if (mid_step) return fail(SF_ERR_STATE, "sf_step_begin twice without sf_step_end");
    int rc = rt.launch_step_half(p, NB, d_cmd, 1);
    if (rc) return rc;
    mid_step = true;
    return SF_OK;
  }
  int step_end_device(const uint8_t *d_cmds) {
    if (!d_cmds) return fail(SF_ERR_ARG, "null command array");
    if (!mid_step) return fail(SF_ERR_STATE, "sf_step_end without sf_step_begin");
    int rc = rt.launch_step_half(p, NB, d_cmds, 2);
    if (rc) return rc;
    mid_step = false;
    return SF_OK;
  }
  int step_end_host(const uint8_t *cmd) {
    if (!cmd) return fail(SF_ERR_ARG, "null command array");
    if (!mid_step) return fail(SF_ERR_STATE, "sf_step_end without sf_step_begin");
    rt.h2d(d_cmd, cmd, (size_t)p.A * p.n_agents);
    return step_end_device(d_cmd);
  }
  int agent_alive_device(uint8_t *d_out) {
    if (!d_out) return fail(SF_ERR_ARG, "null output buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_agent_alive before sf_reset");
    return rt.launch_agent_alive(p, d_out);
  }
  int agent_alive_host(uint8_t *out) {
    if (!out) return fail(SF_ERR_ARG, "null output buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_agent_alive before sf_reset");
    // (d_cmd is free between calls: every step entry point uploads or receives its commands anew)
    int rc = rt.launch_agent_alive(p, d_cmd);
    if (rc) return rc;
    rt.d2h(out, d_cmd, (size_t)p.A * p.n_agents);
    return rt.sync();
  }

  int observe_device(float *d_out) {
    if (!d_out) return fail(SF_ERR_ARG, "null observation buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_observe before sf_reset");
    if (d_out == delta_ptr) delta_ptr = nullptr;  // a plain write: the non-zero map no longer describes this buffer
    return rt.launch_observe(p, NB, d_out, nullptr, 0);
  }
  int observe_sparse_device(uint32_t *d_keys, float *d_vals, uint32_t *d_counts, float *d_pov, int cap) {
    if (!d_keys || !d_vals || !d_counts || !d_pov) return fail(SF_ERR_ARG, "null buffer");
    if (cap < 1) return fail(SF_ERR_ARG, "sf_observe_sparse_device: cap must be positive");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_observe before sf_reset");
    return rt.launch_observe_sparse(p, d_keys, d_vals, d_counts, d_pov, cap);
  }
  int observe_overflow_device(const uint32_t *d_counts, int cap, float *d_dense, float *d_pov) {
    if (!d_counts || !d_dense || !d_pov) return fail(SF_ERR_ARG, "null buffer");
    if (cap < 1) return fail(SF_ERR_ARG, "sf_observe_overflow_device: cap must be positive");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_observe before sf_reset");
    if (d_dense == delta_ptr) delta_ptr = nullptr;
    return rt.launch_observe_overflow(p, d_counts, cap, d_dense, d_pov);
  }
  int observe_device_delta(float *d_out) {
    if (!d_out) return fail(SF_ERR_ARG, "null observation buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_observe before sf_reset");
    int rc;
    if (!d_nzprev && (rc = alloc(d_nzprev, (size_t)p.A * p.n_agents * 961))) return rc;
    const int mode = d_out == delta_ptr ? 2 : 1;  // 1: write everything and record the map; 2: write the differences
    delta_ptr = d_out;
    return rt.launch_observe(p, NB, d_out, d_nzprev, mode);
  }
  int observe_host(float *out) {
    if (!out) return fail(SF_ERR_ARG, "null observation buffer");
    const size_t n = (size_t)p.A * p.n_agents * SF_OBS_FLOATS;
    int rc;
    if (!d_obs && (rc = alloc(d_obs, n))) return rc;
    if ((rc = observe_device(d_obs))) return rc;
    rt.d2h(out, d_obs, n * sizeof(float));
    return rt.sync();
  }

  // between sf_step_begin and sf_step_end the episode-end bookkeeping is half done (the first half clears SC_ENDED): only
  // observe / dump / digest / agent_alive may be called there, as strikeforce.h says
  int not_mid_step(const char *what) const {
    return mid_step ? fail(SF_ERR_STATE, std::string(what) + " between sf_step_begin and sf_step_end") : SF_OK;
  }
  int results_host(int32_t *out) {
    if (!out) return fail(SF_ERR_ARG, "null results buffer");
    if (int rc = not_mid_step("sf_results")) return rc;
    rt.d2h(out, p.results, (size_t)p.A * p.n_agents * 8 * sizeof(int32_t));
    return rt.sync();
  }
  int results_device(int32_t *d_out) {
    if (!d_out) return fail(SF_ERR_ARG, "null results buffer");
    if (int rc = not_mid_step("sf_results_device")) return rc;
    rt.d2d(d_out, p.results, (size_t)p.A * p.n_agents * 8 * sizeof(int32_t));
    return SF_OK;
  }
  int done_host(uint8_t *out) {
    if (!out) return fail(SF_ERR_ARG, "null done buffer");
    if (int rc0 = not_mid_step("sf_done")) return rc0;
    std::vector<int32_t> sc((size_t)p.A * SC_WORDS);
    rt.d2h(sc.data(), p.scal, sc.size() * sizeof(int32_t));
    int rc = rt.sync();
    if (rc) return rc;
    for (int a = 0; a < p.A; ++a)
      out[a] = (uint8_t)(p.auto_reset ? sc[(size_t)a * SC_WORDS + SC_ENDED] : sc[(size_t)a * SC_WORDS + SC_DONE]);
    return SF_OK;
  }

  // generator draws of every arena's last step by phase: out[A][6] = zombie_action, update_bull (1st), human_action,
  // update_bull (2nd), the next loop top's spawns, everything else (0)
  int phase_draws_host(int32_t *out) {
    if (!out) return fail(SF_ERR_ARG, "null output buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "sf_phase_draws before sf_reset");
    std::vector<int32_t> sc((size_t)p.A * SC_WORDS);
    rt.d2h(sc.data(), p.scal, sc.size() * sizeof(int32_t));
    int rc = rt.sync();
    if (rc) return rc;
    for (int a = 0; a < p.A; ++a) {
      const uint32_t w[3] = {(uint32_t)sc[(size_t)a * SC_WORDS + SC_PD01], (uint32_t)sc[(size_t)a * SC_WORDS + SC_PD23],
                             (uint32_t)sc[(size_t)a * SC_WORDS + SC_PD45]};
      int32_t *o = out + (size_t)a * 6;
      o[0] = (int32_t)(w[0] & 0xffffu), o[1] = (int32_t)(w[0] >> 16), o[2] = (int32_t)(w[1] & 0xffffu), o[3] = (int32_t)(w[1] >> 16);
      o[4] = (int32_t)(w[2] & 0xffffu), o[5] = (int32_t)(w[2] >> 16);
    }
    return SF_OK;
  }

  int done_device(uint8_t *d_out) {
    if (!d_out) return fail(SF_ERR_ARG, "null done buffer");
    if (int rc = not_mid_step("sf_done_device")) return rc;
    return rt.launch_done(p, d_out);
  }

  int done_view_device(const int32_t **d_words, int32_t *stride, int32_t *group) {
    if (!d_words || !stride || !group) return fail(SF_ERR_ARG, "null argument");
    *d_words = p.scal + (p.auto_reset ? SC_ENDED : SC_DONE), *stride = SC_WORDS, *group = p.n_agents;
    return SF_OK;
  }

  // ---- parity tooling ----------------------------------------------------------------------------
  int snapshot() {
    const size_t A = (size_t)p.A;
    s_hum.resize(HW_WORDS * A * p.H), s_zom.resize(ZW_WORDS * A * p.Z), s_bul.resize(BW_WORDS * A * p.B);
    s_por.resize(A * p.P), s_rng.resize(A * RNG_WORDS), s_scal.resize(A * SC_WORDS);
    s_flags.resize(A * (size_t)p.cells_pad), s_dmg.resize(A * (size_t)cells), s_pidx.resize(A * (size_t)cells);
    rt.d2h(s_hum.data(), p.hum, s_hum.size() * 4), rt.d2h(s_zom.data(), p.zom, s_zom.size() * 4);
    rt.d2h(s_bul.data(), p.bul, s_bul.size() * 4), rt.d2h(s_por.data(), p.por, s_por.size() * 4);
    rt.d2h(s_rng.data(), p.rng, s_rng.size() * 4), rt.d2h(s_scal.data(), p.scal, s_scal.size() * 4);
    rt.d2h(s_flags.data(), p.flags, s_flags.size()), rt.d2h(s_dmg.data(), p.aux_dmg, s_dmg.size() * 4);
    rt.d2h(s_pidx.data(), p.aux_pidx, s_pidx.size() * 2);
    return rt.sync();
  }

  // decode arena `a` of the snapshot into the canonical records of strikeforce.h
  void decode(int a, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs, sf_bullet_rec *bs, sf_portal_rec *ps,
              uint8_t *cf, int32_t *cd, int32_t *cp) const {
    const size_t A = (size_t)p.A;
    if (hdr) {
      memset(hdr, 0, sizeof *hdr);
      const int32_t *sc = &s_scal[(size_t)a * SC_WORDS];
      hdr->frame = sc[SC_FRAME], hdr->kills = sc[SC_KILLS], hdr->teams_kills = sc[SC_TKILLS], hdr->loot = sc[SC_LOOT];
      hdr->chests = sc[SC_CHESTS], hdr->jomle = (uint32_t)sc[SC_JOMLE], hdr->steps = sc[SC_STEPS];
      hdr->episodes = sc[SC_EPISODES];
      hdr->tb = (int64_t)(((uint64_t)(uint32_t)sc[SC_TB_HI] << 32) | (uint32_t)sc[SC_TB_LO]);
      hdr->serial = (int64_t)(((uint64_t)(uint32_t)sc[SC_SR_HI] << 32) | (uint32_t)sc[SC_SR_LO]);
      for (int i = 0; i < 18; ++i) hdr->rng[i] = (int32_t)(s_rng[(size_t)a * RNG_WORDS + i] & 0xfffffu);
      hdr->done = sc[SC_DONE], hdr->outcome = sc[SC_OUTCOME];
    }
    if (hs)
      for (int i = 0; i < p.H; ++i) {
        auto w = [&](int f) { return s_hum[((size_t)f * A + a) * p.H + i]; };
        sf_human_rec &o = hs[i];
        memset(&o, 0, sizeof o);
        const uint32_t fl = w(HW_FLAGS), q = w(HW_POS), bp = w(HW_BPK);
        const bool placed = (fl & (HF_ALIVE | HF_OCC)) || w(HW_HP) || (fl & HF_WAY_MASK);
        o.alive = !!(fl & HF_ALIVE), o.remote = !!(fl & HF_REMOTE), o.rnpc = !!(fl & HF_RNPC), o.profile = !!(fl & HF_PROF);
        if (placed && q != POS_NONE) o.f = pos_f(q), o.r = pos_r(q), o.c = pos_c(q);
        o.way = (int)(fl & HF_WAY_MASK), o.team = (int)((fl >> HF_TEAM_SH) & 255u);
        o.hp = (int32_t)w(HW_HP), o.stamina = (int32_t)w(HW_STAMINA), o.mindamage = (int32_t)w(HW_MINDAMAGE);
        o.kills = (int32_t)w(HW_KILLS), o.damage = (int32_t)w(HW_DAMAGE), o.effect = (int32_t)w(HW_EFFECT);
        const bool made = (fl & HF_WAY_MASK) != 0;  // slots never used stay all-zero, like the oracle's memset
        o.vec = made ? (int)((fl >> HF_VEC_SH) & 3u) - 1 : 0, o.ind = made ? (int)((fl >> HF_IND_SH) & 15u) - 1 : 0;
        o.cons[0] = w(HW_CONS01) & 0xffff, o.cons[1] = w(HW_CONS01) >> 16, o.cons[2] = w(HW_CONS23) & 0xffff,
        o.cons[3] = w(HW_CONS23) >> 16;
        o.throw_cnt[0] = w(HW_THR01) & 0xffff, o.throw_cnt[1] = w(HW_THR01) >> 16, o.throw_cnt[2] = w(HW_THR23) & 0xffff,
        o.throw_cnt[3] = w(HW_THR23) >> 16;
        o.blocks = bp & 255u, o.portals = (bp >> 8) & 255u, o.portal_ind = made ? (int)((bp >> 16) & 255u) - 1 : 0;
      }
    if (zs)
      for (int i = 0; i < p.Z; ++i) {
        auto w = [&](int f) { return s_zom[((size_t)f * A + a) * p.Z + i]; };
        memset(&zs[i], 0, sizeof zs[i]);
        const uint32_t zp = w(ZW_POS);
        if (!(zp & ZF_ALIVE)) continue;
        zs[i].alive = 1, zs[i].f = pos_f(zp), zs[i].r = pos_r(zp), zs[i].c = pos_c(zp);
        zs[i].hp = (int32_t)w(ZW_HP), zs[i].mindamage = (int32_t)w(ZW_MINDAMAGE), zs[i].super_ = !!(zp & ZF_SUPER);
      }
    if (bs)
      for (int i = 0; i < p.B; ++i) {
        auto w = [&](int f) { return s_bul[((size_t)f * A + a) * p.B + i]; };
        memset(&bs[i], 0, sizeof bs[i]);
        const uint32_t ba = w(BW_A);
        if (!(ba & BA_ALIVE)) continue;
        bs[i].alive = 1, bs[i].f = pos_f(ba), bs[i].r = pos_r(ba), bs[i].c = pos_c(ba);
        bs[i].way = (int)((ba >> BA_WAY_SH) & 3u) + 1, bs[i].traveled = (int)(w(BW_C) >> 16);
        bs[i].damage = (int32_t)w(BW_DAMAGE), bs[i].effect = (int32_t)(int16_t)(w(BW_B) & 0xffffu);
        bs[i].range = (int)(w(BW_C) & 0xffffu), bs[i].owner = (int)(w(BW_B) >> 16), bs[i].ref = !!(ba & BA_REF);
      }
    if (ps)
      for (int i = 0; i < p.P; ++i) {
        memset(&ps[i], 0, sizeof ps[i]);
        const uint32_t pp = s_por[(size_t)a * p.P + i];
        if (!(pp & PF_ACTIVE)) continue;
        ps[i].active = 1, ps[i].f = pos_f(pp), ps[i].r = pos_r(pp), ps[i].c = pos_c(pp);
      }
    for (int ci = 0; ci < cells; ++ci) {
      const uint8_t fl = s_flags[(size_t)a * p.cells_pad + ci];
      if (cf) cf[ci] = fl;
      if (cd) cd[ci] = (fl & SF_CELL_TEMP) ? s_dmg[(size_t)a * cells + ci] : 0;
      if (cp) {
        int v = -1;
        if (fl & (SF_CELL_PIN_UP | SF_CELL_PIN_DN)) v = (fl & SF_CELL_TEMP) ? s_pidx[(size_t)a * cells + ci] : map_pidx[ci];
        cp[ci] = v;
      }
    }
  }

  uint64_t digest_of(const sf_arena_hdr &hdr, const sf_human_rec *hs, const sf_zombie_rec *zs, const sf_bullet_rec *bs,
                     const sf_portal_rec *ps, const uint8_t *cf, const int32_t *cd, const int32_t *cp) const {
    uint64_t d = 0;
    const int64_t hv[9] = {hdr.frame, hdr.kills, hdr.teams_kills, hdr.loot, hdr.chests, hdr.jomle, hdr.steps, hdr.done,
                           hdr.outcome};
    for (int i = 0; i < 9; ++i) d += dg(1, (uint64_t)i, hv[i]);
    for (int i = 0; i < 18; ++i) d += dg(2, (uint64_t)i, hdr.rng[i]);
    for (int i = 0; i < p.H; ++i) {
      const int32_t *w = (const int32_t *)&hs[i];
      for (size_t k = 0; k < sizeof(sf_human_rec) / 4; ++k) d += dg(3, (uint64_t)(i * 32 + (int)k), w[k]);
    }
    for (int i = 0; i < p.Z; ++i) {
      const int32_t *w = (const int32_t *)&zs[i];
      for (size_t k = 0; k < sizeof(sf_zombie_rec) / 4; ++k) d += dg(4, (uint64_t)(i * 8 + (int)k), w[k]);
    }
    for (int i = 0; i < p.B; ++i) {
      const int32_t *w = (const int32_t *)&bs[i];
      for (size_t k = 0; k < sizeof(sf_bullet_rec) / 4; ++k) d += dg(5, (uint64_t)(i * 16 + (int)k), w[k]);
    }
    for (int i = 0; i < p.P; ++i) {
      const int32_t *w = (const int32_t *)&ps[i];
      for (size_t k = 0; k < sizeof(sf_portal_rec) / 4; ++k) d += dg(6, (uint64_t)(i * 4 + (int)k), w[k]);
    }
    for (int ci = 0; ci < cells; ++ci) {
      if (cf[ci]) d += dg(7, (uint64_t)ci, cf[ci]);
      if (cd[ci]) d += dg(8, (uint64_t)ci, cd[ci]);
      if (cp[ci] != -1) d += dg(9, (uint64_t)ci, cp[ci]);
    }
    return d;
  }

  int dump_arena(int a, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs, sf_bullet_rec *bs, sf_portal_rec *ps,
                 uint8_t *cf, int32_t *cd, int32_t *cp) {
    if (a < 0 || a >= p.A) return fail(SF_ERR_ARG, "arena out of range");
    if (!was_reset) return fail(SF_ERR_STATE, "dump before reset");
    int rc = snapshot();
    if (rc) return rc;
    decode(a, hdr, hs, zs, bs, ps, cf, cd, cp);
    return SF_OK;
  }

  int state_digest(uint64_t *out) {
    if (!out) return fail(SF_ERR_ARG, "null digest buffer");
    if (!was_reset) return fail(SF_ERR_STATE, "digest before reset");
    int rc = snapshot();
    if (rc) return rc;
    sf_arena_hdr hdr;
    std::vector<sf_human_rec> hs(p.H);
    std::vector<sf_zombie_rec> zs(p.Z);
    std::vector<sf_bullet_rec> bs(p.B);
    std::vector<sf_portal_rec> ps(p.P);
    std::vector<uint8_t> cf(cells);
    std::vector<int32_t> cd(cells), cp(cells);
    for (int a = 0; a < p.A; ++a) {
      decode(a, &hdr, hs.data(), zs.data(), bs.data(), ps.data(), cf.data(), cd.data(), cp.data());
      out[a] = digest_of(hdr, hs.data(), zs.data(), bs.data(), ps.data(), cf.data(), cd.data(), cp.data());
    }
    return SF_OK;
  }
};

// Items/ text files and character/*.txt of the reference as defaults (values: SURVEY.md Appendix B)
inline void config_defaults(sf_config *c) {
  memset(c, 0, sizeof *c);
  c->abi_version = SF_ABI_VERSION;
  static const int cons[4][3] = {{20, 0, 20}, {0, 200, 10}, {20, 50, 10}, {20, 400, 20}};
  static const int thr[4][4] = {{-15, 50, -20, 100}, {-20, 75, -200, 100}, {-30, 100, -80, 100}, {-35, 125, -80, 100}};
  static const int wp[8][4] = {{-25, 150, -50, 1},  {-40, 175, -60, 1},   {-40, 200, -70, 1},   {-45, 225, -80, 1},
                               {-50, 150, -55, 100}, {-50, 175, -65, 100}, {-50, 200, -75, 100}, {-50, 225, -85, 100}};
  memcpy(c->items.cons, cons, sizeof cons), memcpy(c->items.thr, thr, sizeof thr), memcpy(c->items.weapon, wp, sizeof wp);
  sf_profile human = {1000, 100, 1000, 1, 1, 1, 1000, 0, 0, 0, 0, {0, 0, 0, 0}, {{1, 0}, {1, 0}, {1, 0}, {1, 0}}, {0}, 1};
  sf_profile enemy = {1000, 100, 1000000, 1, 1, 1, 1000, 1, 1, 1, 1, {1, 1, 1, 1}, {{1, 1}, {1, 1}, {1, 1}, {1, 1}},
                      {1, 1, 1, 1, 1, 1, 1, 1}, 1};
  c->player = human, c->npc = enemy;
  c->cap_chests = 9000, c->cap_portals = 16, c->level = 1, c->n_agents = 1, c->arenas = 1;
  for (int i = 0; i < SF_MAX_AGENTS; ++i) c->agent_team[i] = i + 1;
}

}  // namespace sf
