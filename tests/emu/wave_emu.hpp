// wave_emu.hpp — TEST INFRASTRUCTURE: a 64-lane wavefront emulated on the CPU.
//
// Implements the wave-backend interface that strikeforce_amd/csrc/sf_core.hpp is written against
// (see wave_gfx950.hpp for the real one), so that the exact device source can be run, compared with the
// oracle and put under AddressSanitizer on a machine without a GPU.  It is built only into
// tests/emu/libsf_emu.so, which only tests/ loads; the product library never contains or calls it.
#pragma once
#include <stdint.h>
#include <string.h>

#define SF_DEV inline

namespace sf {

// Vector-op profile of the device source (one count per emulated 64-lane operation), by phase of the tick: a
// stand-in for per-phase VALU instruction counts, used to decide where the kernel's issue slots go.
enum { PH_OTHER, PH_RNG, PH_ZOMBIE, PH_PORTAL, PH_HUMAN, PH_TMP, PH_HITS, PH_BULL, PH_TOP, PH_WARM, PH_COUNT };
struct EmuProf {
  uint64_t ops = 0, mark = 0, by[PH_COUNT] = {};
  uint64_t sops = 0, smark = 0, sby[PH_COUNT] = {};  // wave-uniform accesses (readlane, ballot, setlane, uniform LDS)
  int phase = PH_OTHER;
  uint64_t counts[8] = {};  // SF_COUNT(k): how often the device source took path k (tests assert both kinds occur)
};
inline EmuProf &emu_prof() {
  static EmuProf p;
  return p;
}
struct EmuProfScope {
  int prev;
  explicit EmuProfScope(int ph) {
    EmuProf &q = emu_prof();
    q.by[q.phase] += q.ops - q.mark, q.mark = q.ops;
    q.sby[q.phase] += q.sops - q.smark, q.smark = q.sops;
    prev = q.phase, q.phase = ph;
  }
  ~EmuProfScope() {
    EmuProf &q = emu_prof();
    q.by[q.phase] += q.ops - q.mark, q.mark = q.ops;
    q.sby[q.phase] += q.sops - q.smark, q.smark = q.sops;
    q.phase = prev;
  }
};
#define SF_PROF(ph) ::sf::EmuProfScope sf_prof_scope_(::sf::ph)
#define SF_COUNT(k) (++::sf::emu_prof().counts[(k) & 7])
#define SF_STAMP_BEGIN(S)  // in-kernel phase stamps exist only in the device's diagnostic build
#define SF_STAMP(S, ph)
#define SF_STAMP_END(S, a)
#define SF_STAMP_LOADED()
#define SF_STAMP_STEPPED()
#define EMU_OP() (++::sf::emu_prof().ops)
#define EMU_SOP() (++::sf::emu_prof().sops)

struct EmuP {
  uint64_t m;
};
inline EmuP operator&(EmuP a, EmuP b) { return {a.m & b.m}; }
inline EmuP operator|(EmuP a, EmuP b) { return {a.m | b.m}; }
inline EmuP operator!(EmuP a) { return {~a.m}; }

struct EmuV {
  uint32_t v[64];
  EmuV() {}
  EmuV(uint32_t x) {
    for (int i = 0; i < 64; ++i) v[i] = x;
  }
};
#define EMU_BIN(op)                                              \
  inline EmuV operator op(const EmuV &a, const EmuV &b) {        \
    EMU_OP(); EmuV r;                                            \
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] op b.v[i];      \
    return r;                                                    \
  }                                                              \
  inline EmuV operator op(const EmuV &a, uint32_t b) {           \
    EMU_OP(); EmuV r;                                            \
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] op b;           \
    return r;                                                    \
  }
EMU_BIN(+)
EMU_BIN(-)
EMU_BIN(*)
EMU_BIN(&)
EMU_BIN(|)
EMU_BIN(^)
#undef EMU_BIN
inline EmuV operator<<(const EmuV &a, int s) {
  EMU_OP();
  EmuV r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] << s;
  return r;
}
inline EmuV operator>>(const EmuV &a, int s) {
  EMU_OP();
  EmuV r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] >> s;
  return r;
}
inline EmuV operator~(const EmuV &a) {
  EMU_OP();
  EmuV r;
  for (int i = 0; i < 64; ++i) r.v[i] = ~a.v[i];
  return r;
}
#define EMU_CMP(op)                                                                \
  inline EmuP operator op(const EmuV &a, const EmuV &b) {                          \
    EMU_OP(); EmuP r{0};                                                           \
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)(a.v[i] op b.v[i]) << i;         \
    return r;                                                                      \
  }                                                                                \
  inline EmuP operator op(const EmuV &a, uint32_t b) {                             \
    EMU_OP(); EmuP r{0};                                                           \
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)(a.v[i] op b) << i;              \
    return r;                                                                      \
  }
EMU_CMP(==)
EMU_CMP(!=)
#undef EMU_CMP

struct WaveEmu {
  using V = EmuV;
  using P = EmuP;

  static V lane() {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)i;
    return r;
  }
  static uint64_t ballot(P p) {
    EMU_SOP();
    return p.m;
  }
  static int ctz64(uint64_t m) { return __builtin_ctzll(m); }
  static int clz64(uint64_t m) { return __builtin_clzll(m); }
  static uint32_t readlane(const V &v, uint32_t idx) {
    EMU_SOP();
    return v.v[idx & 63u];
  }
  static void setlane(V &v, uint32_t idx, uint32_t val) {
    EMU_SOP();
    v.v[idx & 63u] = val;
  }
  static V select(P p, const V &a, const V &b) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((p.m >> i) & 1ull) ? a.v[i] : b.v[i];
    return r;
  }
  static V sar31(const V &a) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)((int32_t)a.v[i] >> 31);
    return r;
  }
  static P le0(const V &a) {
    EMU_OP();
    P r{0};
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)((int32_t)a.v[i] <= 0) << i;
    return r;
  }
  static P ltu(const V &a, const V &b) {
    EMU_OP();
    P r{0};
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)(a.v[i] < b.v[i]) << i;
    return r;
  }
  static P ltu(const V &a, uint32_t b) {
    EMU_OP();
    P r{0};
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)(a.v[i] < b) << i;
    return r;
  }
  static P frombits(uint64_t m) { return P{m}; }
  static P all() { return P{~0ull}; }
  static V sum18_row1(const V &a) {  // like the DPP version: every lane gets its 16-lane row's sum, row 1 also row 0's
    uint32_t rs[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) rs[i >> 4] += a.v[i];
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = rs[i >> 4] + ((i >> 4) == 1 ? rs[0] : 0u);
    return r;
  }
  static V minu(const V &a, const V &b) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] < b.v[i] ? a.v[i] : b.v[i];
    return r;
  }
  static V shrv(const V &a, const V &sh) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] >> (sh.v[i] & 31u);
    return r;
  }
  static V shlv(const V &a, const V &sh) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] << (sh.v[i] & 31u);
    return r;
  }
  static P gts(const V &a, const V &b) {
    EMU_OP();
    P r{0};
    for (int i = 0; i < 64; ++i) r.m |= (uint64_t)((int32_t)a.v[i] > (int32_t)b.v[i]) << i;
    return r;
  }
  static int popc64(uint64_t m) { return __builtin_popcountll(m); }
  static void lds_store_u8(uint8_t *lds, const V &idx, const V &val, P pred) {
    EMU_OP();
    for (int i = 0; i < 64; ++i)
      if ((pred.m >> i) & 1ull) lds[idx.v[i]] = (uint8_t)val.v[i];
  }
  static void lds_or_u32(uint32_t *base, const V &w, const V &bits, P pred) {
    EMU_OP();
    for (int i = 0; i < 64; ++i)
      if ((pred.m >> i) & 1ull) base[w.v[i]] |= bits.v[i];
  }
  static V lds_or_rtn_u32(uint32_t *base, const V &w, const V &bits, P pred) {
    EMU_OP();
    V r(0u);
    for (int i = 0; i < 64; ++i)
      if ((pred.m >> i) & 1ull) r.v[i] = base[w.v[i]], base[w.v[i]] |= bits.v[i];
    return r;
  }
  static void lds_store_u32(uint32_t *base, const V &w, const V &val, P pred) {
    EMU_OP();
    for (int i = 0; i < 64; ++i)
      if ((pred.m >> i) & 1ull) base[w.v[i]] = val.v[i];
  }
  static void lds_zero(uint32_t *base, uint32_t nwords) { memset(base, 0, (size_t)nwords * 4u); }
  static V mad24(const V &a, uint32_t b, const V &c) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] & 0xffffffu) * (b & 0xffffffu) + c.v[i];
    return r;
  }
  static V mul24(const V &a, const V &b) {  // operands below 2^24, like v_mul_u32_u24
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] & 0xffffffu) * (b.v[i] & 0xffffffu);
    return r;
  }
  static V mul24_su(uint32_t a, const V &b) { return mul24(V(a), b); }
  static V shl1(const V &a) {
    EMU_OP();
    V r;
    for (int i = 0; i < 63; ++i) r.v[i] = a.v[i + 1];
    r.v[63] = 0;
    return r;
  }
  static V lds_u8(const uint8_t *lds, const V &idx, P pred) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((pred.m >> i) & 1ull) ? lds[idx.v[i]] : 0u;
    return r;
  }
  static V lds_u8_any(const uint8_t *lds, const V &idx) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = lds[idx.v[i]];
    return r;
  }
  static uint32_t ulds_u8(const uint8_t *lds, uint32_t idx) {
    EMU_SOP();
    return lds[idx];
  }
  static V lds_u32(const uint32_t *lds, const V &idx, P pred) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((pred.m >> i) & 1ull) ? lds[idx.v[i]] : 0u;
    return r;
  }
  static uint32_t ulds_u32(const uint32_t *lds, uint32_t idx) {
    EMU_SOP();
    return lds[idx];
  }
  static void ulds_store_u32(uint32_t *lds, uint32_t idx, uint32_t val) {
    EMU_SOP();
    lds[idx] = val;
  }
  static void rng_prio_end() {}
  static constexpr bool FUSED_ROUND = false;  // (the fused asm form of the round exists on the device only)
  static V rng_round(V &, uint32_t, V, V, V, V, const uint32_t *, uint32_t &) { return V(0u); }
  static V rng_commit(const V &rl, uint32_t e, const V &la) {
    EMU_OP();
    V r;
    for (int i = 0; i < 63; ++i) r.v[i] = rl.v[i + 1];
    r.v[63] = 0;
    r.v[17] = (e & 0xffffffu) * (la.v[17] & 0xffffffu);
    r.v[18] = (e & 0xffffffu) * (la.v[18] & 0xffffffu);
    return r;
  }
  static V rng_reduce(const V &d, const V &us, const V &bias) {
    EMU_OP();
    uint32_t rs[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) {
      const int32_t a = (int32_t)(d.v[i] << 8) >> 8, b = (int32_t)(us.v[i] << 8) >> 8;  // v_mad_i32_i24: signed 24-bit factors
      rs[i >> 4] += (uint32_t)(a * b) + bias.v[i];
    }
    V r;
    for (int i = 0; i < 64; ++i) {
      const uint32_t x = rs[i >> 4] + ((i >> 4) == 1 ? rs[0] : 0u);
      r.v[i] = (((x & 0xffffu) - (x >> 16)) << 1) + bias.v[i];
    }
    return r;
  }
  static V pow_bytes(const uint32_t *xt, const V &m) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = xt[m.v[i] & 255u] * xt[256 + ((m.v[i] >> 8) & 255u)];
    return r;
  }
  static V pow_pair(const uint32_t *xt, const V &m4) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = xt[(m4.v[i] >> 2) & 255u] * xt[256 + ((m4.v[i] >> 10) & 255u)];
    return r;
  }
  static V gload_u16_at(const uint16_t *base, const V &byte_off) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = base[byte_off.v[i] >> 1];
    return r;
  }
  static V gload_u16(const uint16_t *base, const V &idx, P pred) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((pred.m >> i) & 1ull) ? base[idx.v[i]] : 0u;
    return r;
  }
  static void ulds_store_u8(uint8_t *lds, uint32_t idx, uint32_t val) { lds[idx] = (uint8_t)val; }
  static int32_t uload_i32(const int32_t *p) { return *p; }
  static void ustore_i32(int32_t *p, int32_t v) { *p = v; }
  static int32_t uload_i16(const int16_t *p) { return *p; }
  static void ustore_i16(int16_t *p, int16_t v) { *p = v; }
  static V gload(const uint32_t *base, const V &idx, P pred) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((pred.m >> i) & 1ull) ? base[idx.v[i]] : 0u;
    return r;
  }
  static V gload_u8(const uint8_t *base, const V &idx, P pred) {
    EMU_OP();
    V r;
    for (int i = 0; i < 64; ++i) r.v[i] = ((pred.m >> i) & 1ull) ? base[idx.v[i]] : 0u;
    return r;
  }
  static void gstore(uint32_t *base, const V &idx, const V &val, P pred) {
    for (int i = 0; i < 64; ++i)
      if ((pred.m >> i) & 1ull) base[idx.v[i]] = val.v[i];
  }
  static void copy_g2l(uint8_t *lds, const uint8_t *g, uint32_t n) { memcpy(lds, g, n); }
  // the copy in two halves (wave_gfx950.hpp: loads first, LDS stores later); here the source is read at the store
  template <int U>
  struct G2L {
    const uint8_t *g = nullptr;
  };
  template <int U>
  static void g2l_issue(G2L<U> &q, const uint8_t *g, uint32_t) { q.g = g; }
  template <int U>
  static void g2l_store(const G2L<U> &q, uint8_t *lds, uint32_t n) { memcpy(lds, q.g, n); }
  static void copy_l2g(uint8_t *g, const uint8_t *lds, uint32_t n) { memcpy(g, lds, n); }
};

}  // namespace sf
