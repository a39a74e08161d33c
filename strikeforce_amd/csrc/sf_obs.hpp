// sf_obs.hpp — the observation encoder of gameplay::bot(), bots/bot-0.5/Custom.hpp:29-159 (CU below),
// as pure per-element functions over the HBM arena state: for one (arena, agent) the 31x31 window around
// the agent's human is described by 32 channels, channel-major, each mapped by x -> (float)pow(|x|/10, 0.2).
//
// The kernel in sf_api.hip stages, per window cell, the flag byte and an occupant word
// (human+1 | (zombie+1) << OCC_Z_SH | (designated bullet+1) << OCC_B_SH) in LDS and then evaluates obs_value() for
// every one of the 30 752 outputs with coalesced stores.  tests/emu calls the same functions in loops.
#pragma once
#include <math.h>

#include "../../include/strikeforce.h"
#include "sf_types.hpp"

namespace sf {

// occupant word of a window cell: slots + 1 of the human (<= 64: 7 bits), the zombie (<= 9000: 14 bits) and the
// designated bullet (<= 256: 9 bits) on it, 0 = none
constexpr int OCC_Z_SH = 7, OCC_B_SH = 21;
constexpr uint32_t OCC_H_MASK = 127u, OCC_Z_MASK = 16383u;
static_assert(SF_MAX_HUMANS <= (int)OCC_H_MASK && SF_MAX_ZOMBIES <= (int)OCC_Z_MASK && SF_MAX_BULLETS + 1 <= (1 << (32 - OCC_B_SH)),
              "occupant word fields");

struct ObsView {  // read-only view of one arena's entity tables
  const uint32_t *hum_, *zom_, *bul_;
  const Tables *tab;
  int A, H, Z, B, a;
  int zA, za;     // the zombie table's own arena count / arena (k_observe leaves a large table in HBM beside staged humans and bullets)
  int npc_block;  // Params::npc_block: which table block holds the NPC record
  SF_HD ObsView(const Params &p, int arena)
      : hum_(p.hum), zom_(p.zom), bul_(p.bul), tab(p.tab), A(p.A), H(p.H), Z(p.Z), B(p.B), a(arena), zA(p.A), za(arena),
        npc_block(p.npc_block) {}
  SF_HD uint32_t hum(int f, int i) const { return hum_[((size_t)f * A + a) * H + i]; }
  SF_HD uint32_t zom(int f, int i) const { return zom_[((size_t)f * zA + za) * Z + i]; }
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

// describe() CU:29-135 for a cell with flag byte `fl`, side-table damage `cdmg` and occupant word `occ`, seen by a
// player of team `pteam`: calls emit(k, x) for every feature k whose raw value x (before the pow map) can be
// non-zero; every feature not emitted is 0.  (A visitor instead of a 32-float array keeps the device kernel's
// register count low; most cells emit 3 features, most window cells none.)
template <class F>
SF_HD inline void obs_cell_emit(const ObsView &v, uint32_t fl, int32_t cdmg, uint32_t occ, int pteam, F &&emit) {
  const int h = (int)(occ & OCC_H_MASK) - 1, z = (int)((occ >> OCC_Z_SH) & OCC_Z_MASK) - 1, b = (int)(occ >> OCC_B_SH) - 1;
  const bool s0 = h >= 0, s1 = z >= 0, s2 = b >= 0;
  const bool s3 = fl & SF_CELL_WALL, s4 = fl & SF_CELL_CHEST, s5 = fl & SF_CELL_PIN_UP, s6 = fl & SF_CELL_PIN_DN,
             s7 = fl & SF_CELL_POUT, s10 = fl & SF_CELL_TEMP;
  if (s0 || s1) emit(0, 1.f);
  if (s2) emit(1, 1.f);
  if (s3) emit(2, 1.f);
  if (s4) emit(3, 1.f);
  if (s5 || s6) emit(4, 1.f);
  if (s7) emit(5, 1.f);
  if (s10) emit(6, 1.f);
  if (s1) emit(10, 1.f);
  if (s3 || s5 || s6 || s0 || s1) {
    emit(15, 1.f), emit(16, 1.f);
    if (s10 || s0 || s1) emit(17, 1.f);
    if (s0)
      emit(18, (float)((int32_t)v.hum(HW_HP, h) / 1000.0));
    else if (s1)
      emit(18, (float)((int32_t)v.zom(ZW_HP, z) / 1000.0));
    else if (s10)
      emit(18, s3 ? (float)((1100 - cdmg) / 1000.0) : (float)((1000 - cdmg) / 1000.0));
  } else if (s7)
    emit(15, 1.f);
  if (s0) {
    const uint32_t hf = v.hum(HW_FLAGS, h), hb = v.hum(HW_BPK, h);
    const int t = (int)((hf >> HF_TEAM_SH) & 255u);
    emit(!t ? 9 : (t == pteam ? 7 : 8), 1.f);
    emit(11, (float)(int32_t)v.hum(HW_KILLS, h));
    emit(12, (float)(int32_t)(hb & 255u));
    emit(13, (float)(int32_t)((hb >> 8) & 255u));
    emit(14, (float)(((hb >> 16) & 255u) != 0u));
    emit(20 + (int)(hf & HF_WAY_MASK) - 1, 1.f);
    int v0, v1;
    obs_damage_effect(v.tab->der[(hf & HF_PROF) ? v.npc_block : (int)((hf >> HF_AGP_SH) & 15u)], hf, (int32_t)v.hum(HW_STAMINA, h),
                      (int32_t)v.hum(HW_MINDAMAGE, h), v0, v1);
    emit(24, (float)(v0 / 1000.0)), emit(25, (float)(-v1 / 1000.0));
    emit(26, (float)((int32_t)v.hum(HW_STAMINA, h) / 1000.0));
    emit(30, (float)((int32_t)v.hum(HW_DAMAGE, h) / 1000.0));
    emit(31, (float)(-(int32_t)v.hum(HW_EFFECT, h) / 1000.0));
  } else if (s1) {
    for (int k = 20; k < 24; ++k) emit(k, (float)0.01);
    emit(24, (float)((int32_t)v.zom(ZW_MINDAMAGE, z) / 1000.0));
  } else if (s2) {
    const uint32_t ba = v.bul(BW_A, b), bc = v.bul(BW_C, b);
    emit(19, 1.f);
    emit(20 + (int)((ba >> BA_WAY_SH) & 3u), (float)(((int)(bc & 0xffffu) - (int)(bc >> 16)) / 100.0));
    emit(24, (float)((int32_t)v.bul(BW_DAMAGE, b) / 1000.0));
    emit(25, (float)(-(int32_t)(int16_t)(v.bul(BW_B, b) & 0xffffu) / 1000.0));
  } else if (s7) {
    emit(24, (float)(20 / 1000.0)), emit(25, (float)(10 / 1000.0));
  }
  if (s4) {
    const int32_t *c = v.tab->cons_items[(fl >> SF_CELL_CONS_SHIFT) & 3u];
    emit(27, (float)(c[0] / 1000.0)), emit(28, (float)(c[2] / 1000.0)), emit(29, (float)(c[1] / 1000.0));
  }
}

// flag byte of the plain static cell of class c (0 '#', 1 '^', 2 'v', 3 'O', 4..7 chest of type c - 4)
SF_HD inline uint32_t obs_class_flags(int c) {
  return c == 0 ? (uint32_t)SF_CELL_WALL : c == 1 ? (uint32_t)SF_CELL_PIN_UP : c == 2 ? (uint32_t)SF_CELL_PIN_DN : c == 3 ? (uint32_t)SF_CELL_POUT
       : (uint32_t)SF_CELL_CHEST | ((uint32_t)(c - 4) << SF_CELL_CONS_SHIFT);
}

// CU:157: obs.push_back(std::pow(std::abs(x) / 10, 0.2)) — float abs, float / int, double pow, narrowed to float
SF_HD inline float obs_map(float x) {
  if (x == 0.f) return 0.f;
  const float y = fabsf(x) / 10;
  return (float)pow((double)y, 0.2);
}

// The handful of raw values that make up almost every non-zero feature (1.0 flags, the 0.01 zombie facing, the
// portal-exit and chest constants) are mapped through a table the host fills with obs_map() itself, so on the
// device they cost a few compares instead of a double-precision pow.  Returns false for any other value.
SF_HD inline bool obs_map_fast(const Tables &t, float x, float &y) {
  if (x == 0.f) {
    y = 0.f;
    return true;
  }
  for (int i = 0; i < t.obs_n; ++i)
    if (x == t.obs_in[i]) {
      y = t.obs_out[i];
      return true;
    }
  return false;
}

}  // namespace sf
