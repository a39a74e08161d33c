// sf_obs.hpp — the observation encoder of gameplay::bot(), bots/bot-0.5/Custom.hpp:29-159 (CU below),
// as pure per-element functions over the HBM arena state: for one (arena, agent) the 31x31 window around
// the agent's human is described by 32 channels, channel-major, each mapped by x -> (float)pow(|x|/10, 0.2).
//
// The kernel in sf_api.hip stages, per window cell, the flag byte and an occupant word
// (human+1 | (zombie+1) << 8 | (designated bullet+1) << 16) in LDS and then evaluates obs_value() for
// every one of the 30 752 outputs with coalesced stores.  tests/emu calls the same functions in loops.
#pragma once
#include <math.h>

#include "../../include/strikeforce.h"
#include "sf_types.hpp"

namespace sf {

struct ObsView {  // read-only view of one arena's entity tables
  const uint32_t *hum_, *zom_, *bul_;
  const Tables *tab;
  int A, H, Z, B, a;
  SF_HD ObsView(const Params &p, int arena)
      : hum_(p.hum), zom_(p.zom), bul_(p.bul), tab(p.tab), A(p.A), H(p.H), Z(p.Z), B(p.B), a(arena) {}
  SF_HD uint32_t hum(int f, int i) const { return hum_[((size_t)f * A + a) * H + i]; }
  SF_HD uint32_t zom(int f, int i) const { return zom_[((size_t)f * A + a) * Z + i]; }
  SF_HD uint32_t bul(int f, int i) const { return bul_[((size_t)f * A + a) * B + i]; }
};

// window slot of packed position q for an observer centred on `c` (same floor), or -1
SF_HD inline int obs_window_slot(uint32_t q, uint32_t c) {
  if ((q >> 20) != (c >> 20)) return -1;
  const int di = pos_r(q) - pos_r(c) + SF_OBS_WINDOW / 2, dj = pos_c(q) - pos_c(c) + SF_OBS_WINDOW / 2;
  if (di < 0 || dj < 0 || di >= SF_OBS_WINDOW || dj >= SF_OBS_WINDOW) return -1;
  return di * SF_OBS_WINDOW + dj;
}

// Human::get_damage_effect Character.hpp:429-443
SF_HD inline void obs_damage_effect(const Derived &d, uint32_t fl, int32_t stamina, int32_t mindamage, int &v0, int &v1) {
  const int vec = (int)((fl >> HF_VEC_SH) & 3u) - 1, sel = (int)((fl >> HF_IND_SH) & 15u) - 1;
  int dmg = d.cd_punch > mindamage ? d.cd_punch : mindamage;
  v0 = dmg, v1 = 0;
  if (vec == 1) {
    const int32_t *b = d.thr[sel];
    if (0 <= stamina + b[0]) {
      int m = b[1];
      if (b[1] + mindamage > m) m = b[1] + mindamage;
      if (dmg > m) m = dmg;
      v0 = m, v1 = b[2];
      return;
    }
  }
  if (vec == 2) {
    const int32_t *w = d.weapon[sel];
    if (0 <= stamina + w[0]) {
      int m = d.cd_weapon[sel];
      if (w[1] + mindamage > m) m = w[1] + mindamage;
      if (dmg > m) m = dmg;
      v0 = m, v1 = w[2];
    }
  }
}

// describe() CU:29-135: feature k of a cell with flag byte `fl`, side-table damage `cdmg` and occupants `occ`,
// seen by a player of team `pteam`.  Returns the raw feature (before the pow map).
SF_HD inline float obs_feature(const ObsView &v, int k, uint32_t fl, int32_t cdmg, uint32_t occ, int pteam) {
  const int h = (int)(occ & 255u) - 1, z = (int)((occ >> 8) & 255u) - 1, b = (int)(occ >> 16) - 1;
  const bool s0 = h >= 0, s1 = z >= 0, s2 = b >= 0;
  const bool s3 = fl & SF_CELL_WALL, s4 = fl & SF_CELL_CHEST, s5 = fl & SF_CELL_PIN_UP, s6 = fl & SF_CELL_PIN_DN,
             s7 = fl & SF_CELL_POUT, s10 = fl & SF_CELL_TEMP;
  const bool blocked = s3 || s5 || s6 || s0 || s1;
  switch (k) {
    case 0: return (float)(s0 || s1);
    case 1: return (float)s2;
    case 2: return (float)s3;
    case 3: return (float)s4;
    case 4: return (float)(s5 || s6);
    case 5: return (float)s7;
    case 6: return (float)s10;
    case 7: case 8: case 9: {
      if (!s0) return 0.f;
      const int t = (int)((v.hum(HW_FLAGS, h) >> HF_TEAM_SH) & 255u);
      const int cls = !t ? 2 : (t == pteam ? 0 : 1);
      return cls == k - 7 ? 1.f : 0.f;
    }
    case 10: return (float)s1;
    case 11: return s0 ? (float)(int32_t)v.hum(HW_KILLS, h) : 0.f;
    case 12: return s0 ? (float)(int32_t)(v.hum(HW_BPK, h) & 255u) : 0.f;
    case 13: return s0 ? (float)(int32_t)((v.hum(HW_BPK, h) >> 8) & 255u) : 0.f;
    case 14: return s0 ? (float)(((v.hum(HW_BPK, h) >> 16) & 255u) != 0u) : 0.f;
    case 15: return (blocked || s7) ? 1.f : 0.f;
    case 16: return blocked ? 1.f : 0.f;
    case 17: return blocked ? (float)(s10 || s0 || s1) : 0.f;
    case 18: {
      if (!blocked) return 0.f;
      if (s0) return (float)((int32_t)v.hum(HW_HP, h) / 1000.0);
      if (s1) return (float)((int32_t)v.zom(ZW_HP, z) / 1000.0);
      if (s10) return s3 ? (float)((1100 - cdmg) / 1000.0) : (float)((1000 - cdmg) / 1000.0);
      return 0.f;
    }
    case 19: return (!s0 && !s1 && s2) ? 1.f : 0.f;
    case 20: case 21: case 22: case 23: {
      const int i = k - 20;
      if (s0) return ((int)(v.hum(HW_FLAGS, h) & HF_WAY_MASK) - 1 == i) ? 1.f : 0.f;
      if (s1) return (float)0.01;
      if (s2) {
        const uint32_t ba = v.bul(BW_A, b), bc = v.bul(BW_C, b);
        if ((int)((ba >> BA_WAY_SH) & 3u) != i) return 0.f;
        return (float)(((int)(bc & 0xffffu) - (int)(bc >> 16)) / 100.0);
      }
      return 0.f;
    }
    case 24: case 25: {
      if (s0) {
        const uint32_t hf = v.hum(HW_FLAGS, h);
        int v0, v1;
        obs_damage_effect(v.tab->der[(hf & HF_PROF) ? 1 : 0], hf, (int32_t)v.hum(HW_STAMINA, h),
                          (int32_t)v.hum(HW_MINDAMAGE, h), v0, v1);
        return k == 24 ? (float)(v0 / 1000.0) : (float)(-v1 / 1000.0);
      }
      if (s1) return k == 24 ? (float)((int32_t)v.zom(ZW_MINDAMAGE, z) / 1000.0) : 0.f;
      if (s2)
        return k == 24 ? (float)((int32_t)v.bul(BW_DAMAGE, b) / 1000.0)
                       : (float)(-(int32_t)(int16_t)(v.bul(BW_B, b) & 0xffffu) / 1000.0);
      if (s7) return k == 24 ? (float)(20 / 1000.0) : (float)(10 / 1000.0);
      return 0.f;
    }
    case 26: return s0 ? (float)((int32_t)v.hum(HW_STAMINA, h) / 1000.0) : 0.f;
    case 27: case 28: case 29: {
      if (!s4) return 0.f;
      const int32_t *c = v.tab->cons_items[(fl >> SF_CELL_CONS_SHIFT) & 3u];
      return (float)((k == 27 ? c[0] : k == 28 ? c[2] : c[1]) / 1000.0);
    }
    case 30: return s0 ? (float)((int32_t)v.hum(HW_DAMAGE, h) / 1000.0) : 0.f;
    case 31: return s0 ? (float)(-(int32_t)v.hum(HW_EFFECT, h) / 1000.0) : 0.f;
  }
  return 0.f;
}

// CU:157: obs.push_back(std::pow(std::abs(x) / 10, 0.2)) — float abs, float / int, double pow, narrowed to float
SF_HD inline float obs_map(float x) {
  if (x == 0.f) return 0.f;
  const float y = fabsf(x) / 10;
  return (float)pow((double)y, 0.2);
}

}  // namespace sf
