// sf_core.hpp — one arena per 64-lane wavefront: the per-tick gameplay path of the reference
// (StrikeForce-client/gameplay.hpp:1443-1472 and everything it calls) on struct-of-arrays state.
//
// Execution model.  One wavefront owns one arena.  Entities live in registers, one lane per entity
// slot (humans, zombies, portals: lane = slot; bullets: NB registers per lane, slot = j*64 + lane).
// The only per-cell storage is one flag byte per cell, staged in LDS for the duration of a launch.
// "Who is on cell q?" is answered by a lane compare + wavefront ballot instead of a pointer grid;
// phases whose order matters in the reference (RNG draw order, slot-order sweeps, last-entrant-wins)
// run as wave-uniform loops over ballot bit masks, and everything else runs lane-parallel.
//
// The file is a template over a wave backend W (per-lane value type V, predicate type P, ballot,
// readlane, DPP reductions, LDS/global access).  The product instantiates it with WaveGfx950
// (wave_gfx950.hpp); tests/emu instantiates the same source with a 64-lane CPU emulator so that the
// kernel logic can be checked against the oracle (and under ASan) on machines without a GPU.
// SF_DEV is defined by the backend header included before this file.
//
// Citations: G = gameplay.hpp, CH = Character.hpp, IT = Item.hpp, RN = random.hpp.
#pragma once
#include "../../include/strikeforce.h"
#include "sf_types.hpp"

#if defined(SF_EXP_FAKE_DRAWS) && !defined(SF_EXPERIMENT_BUILD)
#error "SF_EXP_FAKE_DRAWS makes draw() return wrong numbers on purpose (tools/experiments/r03_rows): timing builds only, together with -DSF_EXPERIMENT_BUILD"
#endif

namespace sf {

enum { SH_WALL, SH_HUMAN, SH_ZOMBIE, SH_PUP, SH_PDN, SH_BULLET, SH_CHEST, SH_POUT, SH_EMPTY };
enum { LIM_PORTAL = 1000, LIM_BLOCK = 1100 };  // G:37

// HBM_PLANE: the arena's flag plane is too large to stage in LDS next to enough other wavefronts (128x128 and up):
// `lds` then points at the plane in HBM itself (L2-cached; the same accessors compile to global loads/stores) and
// only the 2 KiB power table sits in LDS.
// BITMAPS: the per-cell scratch bitmaps ("cell bitmaps" below) fit in LDS; without them the ballot loops run.
// ZL: large slot pools — more than 64 zombie slots or more than 64 portal exits (the reference's pools hold 9000 each,
// G:37,51-53, and nothing culls the herd: a Timer game on the shipped maps passes 64 live zombies after 1 650 steps and
// 64 exits, one per NPC human that ever placed its portal, after 6 000).  The zombie table and the exit table then live
// in LDS, [field][slot], and every phase that sweeps them walks 64-slot words — word-major = slot order — up to the
// highest word in use (Arena::zwn, pwn).  The register form (one lane per slot) is untouched by it: every ZL branch
// below is `if constexpr`.
template <class W, int NB, bool HBM_PLANE = false, bool BITMAPS = !HBM_PLANE, bool ZL = false>
struct Core {
  using V = typename W::V;
  using P = typename W::P;

  struct Arena {
    // humans (lane < H)
    V hpos, hfl, hhp, hst, hmd, hk, hdm, hef, hc01, hc23, ht01, ht23, hbpk, hcmd;
    // zombies (lane < Z); with ZL: unused, the table is in LDS at zl[field * zcap + slot]
    V zpos, zhp, zmd;
    uint32_t *zl;
    uint32_t zcap;  // slots per field in LDS: 64 * Params::ZW
    uint32_t zwn;   // words in use: every live zombie's slot is below 64 * zwn; LDS words from zwn on are undefined
    uint32_t zwhi;  // the largest zwn since load(): HBM words in [zwn, zwhi) still hold an earlier episode's zombies
    // portal exits with ZL: pl[slot] in LDS behind the zombie table, the same bookkeeping
    uint32_t *pl;
    uint32_t pwn, pwhi;
    // bullets (slot = j * 64 + lane)
    V ba[NB], bd[NB], bb[NB], bc[NB];
    // portals (lane < P)
    V ppos;
    // RNG (lane < 18): log_3(random[i]) (bit 16 set: random[i] == 0), us[i], seed[i]  RN:31
    V rl, rus, rseed;
    // the NEXT episode's generator, warmed up a few draws per step so that an in-kernel restart does not stall on
    // _srand's 1024 serial draws (RN:73-74): log state, seed digits, draws done so far (0..1024)
    V rl2, rseed2;
    uint32_t warm;
    uint32_t wrate;  // warm-up draws per call site (4 sites per step)
    // RNG power table in LDS: 3^i (i < 256), then 3^(256 i)
    const uint32_t *xt;
    // human_action's command-class / stat table in LDS (Tables::hatab, sf_types.hpp HT_*)
    const uint32_t *ht;
    // BM_COUNT scratch bitmaps in LDS, one bit per cell, all-zero between uses ("cell bitmaps" below); null without BITMAPS
    uint32_t *bm;
    // one-deep lookahead of draw(): the log looked up for the next draw, and whether it is valid
    V la;  // valid whenever draw() can run: (re)issued by load(), srand_(), the adoption of a warmed-up generator, draw()
    // the same for the next episode's generator (prewarm_one)
    V la2;
    uint32_t la2_ok;
    // wave-uniform scalars  G:461
    int32_t frame, kills, tkills, loot, chests, steps, episodes, done, outcome, ended;
    uint32_t jomle;
    uint32_t tb_lo, tb_hi, sr_lo, sr_hi;
    uint32_t dirty;  // the LDS flag plane differs from HBM
    // generator draws of the last step by phase (SC_PD01 .. SC_PD45, sf_phase_draws): pins the ORDER in which phases draw
    uint32_t pd01, pd23, pd45;
#ifdef SF_DIAG_STAMPS
    V dacc;  // diagnostic build only (tools/diag_stamps.sh): wave cycles per tick phase, lane = phase
    uint32_t dlast;
#endif
  };

  // ------------------------------------------------------------------------------------------------
  // RNG: 18-tap generator over Z/65537, RN:27-77.  Taps are lane-parallel, the mod-exp is wave-uniform.
  // Z/65537* is cyclic of order 65536 with generator 3, so x^e = 3^(log3(x) * e mod 65536).  The state keeps
  // log3(random[i]); a power is then two 256-entry LDS lookups (3^lo, 3^(256 hi)) and one multiply instead of a
  // 16-step square-and-multiply chain, and the new value's log is one L2 lookup in the log table (sf_types.hpp LOGT_*).
  // Bit-identical to RN:54-62 by construction (checked against the reference's known answers).
  static constexpr uint32_t RL_ZERO = 0x10000u;  // random[i] == 0 (only until the first 18 draws after _srand)
  // x mod 65537 for x < 2^32: lo16 - hi16, plus 65537 if that went negative (as unsigned: the smaller of the two)
  static SF_DEV V mod65537_v(V x) {
    const V t = (x & 0xffffu) - (x >> 16);
    return W::minu(t, t + 65537u);
  }
  static SF_DEV V pow3_v(const uint32_t *xt, V m, P pred) {  // 3^m mod 65537, m < 65536
    // 3^lo (lo < 256) is never 65536 = 3^32768, so the product cannot wrap 2^32 and needs no corner term
    return mod65537_v(W::lds_u32(xt, m & 255u, pred) * W::lds_u32(xt + 256, m >> 8, pred));
  }

  // draw_core is the general form of RN:54-62 (zero-valued taps allowed): _srand's warm-up and the first draws of a
  // fresh generator use it.  The wave-uniform tail of a draw (mod, log lookup, power) is computed on the vector unit,
  // redundantly in every lane: the scalar unit is shared by the four SIMDs of a CU and was the first bottleneck
  // measured (SQ_INSTS_SALU ~ SQ_INSTS_VALU).  The hot path is draw() / draw_issue() below.
  template <bool WANT_OUT>
  static SF_DEV V draw_core(V &rl, const V &rus, const V &rseed, uint32_t &jomle, const uint32_t *xt, const Params &p) {
    const P tap = W::ltu(W::lane(), 18u) & ((rl & RL_ZERO) == 0u);
    const V pw = pow3_v(xt, (rl * rseed) & 0xffffu, tap);  // p[random[i]][seed[i]] = random[i]^seed[i]
    // the sum stays on the vector unit: valid on lanes 16..31 (DPP row reductions + row_bcast:15), which is all that
    // is needed — the new value is inserted on lane 17 and read back from lane 17
    const V sum = W::sum18_row1(W::select(tap, rus * pw, V(0u))) + 1u;  // < 2^24
    V t = mod65537_v(sum);
    t = W::select(t == 0u, V(1u), t);  // binpow(sum + (int)(sum == 0), ...)
    jomle += 1u;
    const uint32_t e = jomle & 0xffffu;  // b %= mod - 1
    const V lg = W::gload_u16(p.logt, t + (uint32_t)LOGT_OFF, W::all());  // t <= 65536 whatever a junk lane summed
    const V lnew = (lg * e) & 0xffffu;
    rl = W::select(W::lane() == 17u, lnew, W::shl1(rl));  // the 17 swaps: rotate left, new value last
    if (WANT_OUT) return pow3_v(xt, lnew, W::all()) & 1023u;
    return V(0u);
  }
  // The draw the game logic consumes, software-pipelined one deep: the generator's sequence does not depend on the
  // game, so the table lookup for draw n+1 (the one long-latency link of the chain) is issued at the end of draw
  // n and waited for at the beginning of draw n+1; the logic in between runs under its latency.  `S.la` holds the
  // looked-up log; it is a pure function of the committed state (rl, jomle), so it is simply dropped at store time.
  //
  // The hot-path form of the first half (draw_issue) differs from draw_core in three ways, all of them safe only on a
  // warmed-up generator (no zero among random[0..17]: a new value is a power of a non-zero residue, so this holds from
  // the 18th draw after _srand on, and draw() is only reached after the 1024 warm-up draws):
  //   * no tap predicate: lanes >= 18 carry us = 0, so their term is 0 whatever their power is;
  //   * lane 18 carries seed 1 and a copy of the newest log, so the same pair of LDS lookups that powers the 18
  //     taps also yields 3^lnew = the value of the draw just made (its low 10 bits are rand()'s result, RN:61);
  //   * the power is left as the signed difference lo16 - hi16 of the table product (congruent mod 65537, magnitude
  //     < 2^16); the terms are summed signed and one constant multiple of 65537 makes the total positive before the
  //     single reduction of the sum.
  //     The constant is spread over the lanes: every lane's product gets SUM_BIAS_LANE added inside the multiply (mad),
  //     and the 32 lanes of rows 0 and 1 together add 32 * SUM_BIAS_LANE = 383 * 65537 + 1 — the generator's `sum = 1`
  //     plus a multiple of 65537 that keeps every row's partial sum positive (a row of 16 terms is > -16 * 10 * 65536),
  //     so every lane's index stays inside the table: x < 2^26, hi16 <= 563 < LOGT_OFF.  The same (even) constant is
  //     added to the byte offset 2 t, which keeps the load's unsigned register offset non-negative; log_base() moves
  //     the table pointer down by as much.
  static constexpr uint32_t SUM_BIAS_LANE = 63489u + 11u * 65537u;  // 32 * 63489 = 31 * 65537 + 1
  static_assert((32ull * SUM_BIAS_LANE) % 65537ull == 1ull, "the lanes' bias must add up to the generator's +1");
  static_assert(SUM_BIAS_LANE % 2u == 0u, "it is also a byte offset into a table of 16-bit entries");
  static_assert(16ull * SUM_BIAS_LANE > 16ull * 10ull * 65536ull, "a row's partial sum must stay positive");
  static_assert((32ull * SUM_BIAS_LANE + 18ull * 10ull * 65535ull) >> 16 < (unsigned long long)LOGT_OFF, "index below the table");
  static SF_DEV const uint16_t *log_base(const Params &p) { return p.logt + LOGT_OFF - (int)(SUM_BIAS_LANE / 2u); }
  static SF_DEV V issue_offset(const uint32_t *xt, V rl, V rseed, V rus, V &d) {
    const V m = W::mul24(rl, rseed);  // log * seed; bits above 65536 are multiples of the group order
    const V pr = W::pow_bytes(xt, m);
    d = (pr & 0xffffu) - (pr >> 16);  // == 3^m (mod 65537), in (-65536, 65536)
    return W::rng_reduce(d, rus, V(SUM_BIAS_LANE));
  }
  static SF_DEV uint32_t draw_issue(Arena &S, const Params &p) {
    V d;
    S.la = W::gload_u16_at(log_base(p), issue_offset(S.xt, S.rl, S.rseed, S.rus, d));
    W::rng_prio_end();
    const int32_t o = (int32_t)W::readlane(d, 18u);
    return (uint32_t)(o + ((o >> 31) & 65537)) & 1023u;
  }
  static SF_DEV uint32_t draw(Arena &S, const uint8_t *, const Params &p) {  // RN:54-62
#ifdef SF_EXP_FAKE_DRAWS
    // timing build only (tools/experiments/r03_rows): what a draw would cost if it came out of a ring filled by producer
    // waves — one v_readlane and a few scalar instructions.  Wrong results by construction; never the product library.
    S.jomle = S.jomle * 1664525u + 1013904223u;
    return (W::readlane(S.rl, (S.jomle >> 26) & 31u) ^ (S.jomle >> 16)) & 1023u;
#endif
    SF_PROF(PH_RNG);
    S.jomle += 1u;
    if constexpr (W::FUSED_ROUND) {  // the same round as below, as two asm blocks around the LDS lookups (wave_gfx950.hpp)
      uint32_t out;
      const V off = W::rng_round(S.rl, S.jomle, S.la, S.rseed, S.rus, V(SUM_BIAS_LANE), S.xt, out);
      S.la = W::gload_u16_at(log_base(p), off);
      return out;
    }
    // rotate left, new value last (+ copy on 18).  Neither jomle nor the product is reduced mod 2^16: every consumer of
    // the hot state — the 24-bit multiply by the seed, the table offsets, store() — takes the low 16 bits itself
    S.rl = W::rng_commit(S.rl, S.jomle, S.la);
    return draw_issue(S, p);  // draw n+1's lookup is now in flight; lane 18's power is draw n's value
  }

  static SF_DEV void seed_digits(V &digits, uint64_t x, uint32_t lane18 = 0u) {  // RN:65-68: decimal digit i of x, plus one, on lane i
    digits = V(0u);
    W::setlane(digits, 18u, lane18);  // seeds: 1 (draw_issue's output lane); us: 0
    for (int i = 0; i < 18; ++i) {
      W::setlane(digits, (uint32_t)i, (uint32_t)(x % 10u) + 1u);
      x /= 10u;
    }
  }
  static SF_DEV void srand_(Arena &S, const uint8_t *lds, const Params &p, uint64_t tb, uint64_t us) {  // RN:64-76
    S.rl = V(RL_ZERO);
    seed_digits(S.rus, us);
    seed_digits(S.rseed, tb, 1u);
    S.jomle = 18u;
    (void)lds;
    for (int i = 0; i < 1024; ++i) draw_core<false>(S.rl, S.rus, S.rseed, S.jomle, S.xt, p);
    draw_issue(S, p);
  }
  // advance the next episode's warm-up by up to n draws
  // One warm-up draw of the NEXT episode's generator.  Past its first 18 draws (no zero left in the state) it runs in
  // the same lean, split form as draw(): commit the draw whose log lookup was issued by the previous call, issue the
  // next one, so that the lookup's latency passes under whatever runs between two calls.  `S.la2` is a pure function
  // of (rl2, warm) and is dropped at store time like `S.la`.
  static SF_DEV void prewarm_issue(Arena &S, const Params &p) {
    V d;
    S.la2 = W::gload_u16_at(log_base(p), issue_offset(S.xt, S.rl2, S.rseed2, S.rus, d));
    S.la2_ok = 1u;
  }
  static SF_DEV void prewarm_one(Arena &S, const Params &p) {
#ifdef SF_EXP_FAKE_DRAWS
    S.warm = 1024u;
    return;
#endif
    SF_PROF(PH_WARM);
    if (S.warm >= 1024u) return;
#ifdef SF_EXP_FREE_WARMUP
    // timing build only (round 4): three of four warm-up draws cost nothing — the bound of running the next episode's
    // generator in the idle lanes of the game's own rounds.  Wrong numbers by construction; never the product library.
    if (S.warm & 3u) {
      ++S.warm;
      return;
    }
#endif
    uint32_t j2 = 18u + S.warm;
    if (S.warm < 18u) {
      draw_core<false>(S.rl2, S.rus, S.rseed2, j2, S.xt, p);
    } else {
      if (!S.la2_ok) prewarm_issue(S, p);
      const V lnew = W::mul24(S.la2, V((j2 + 1u) & 0xffffu)) & 0xffffu;
      S.rl2 = W::select(W::lane() == 17u, lnew, W::shl1(S.rl2));
      S.la2_ok = 0u;
      if (S.warm + 1u < 1024u) prewarm_issue(S, p);
    }
    ++S.warm;
  }
  static SF_DEV void prewarm(Arena &S, const uint8_t *lds, const Params &p, uint32_t n) {
    (void)lds;
    for (; n && S.warm < 1024u; --n) prewarm_one(S, p);
  }

  // ------------------------------------------------------------------------------------------------
  // cell queries (q wave-uniform)
  static SF_DEV int DX(int d) { return d == 0 ? 1 : (d == 2 ? -1 : 0); }  // wdx CH:47
  static SF_DEV int DY(int d) { return d == 1 ? 1 : (d == 3 ? -1 : 0); }  // wdy
  static SF_DEV uint32_t cellidx(const Params &p, int f, int r, int c) { return (uint32_t)((f * p.N + r) * p.M + c); }
  static SF_DEV uint32_t cellidx_q(const Params &p, uint32_t q) { return cellidx(p, pos_f(q), pos_r(q), pos_c(q)); }
  static SF_DEV bool inmap(const Params &p, int r, int c) {
    return (uint32_t)r < (uint32_t)p.N && (uint32_t)c < (uint32_t)p.M;
  }
  // Out-of-range cells read as indestructible wall (the reference reads without bounds checks in
  // zombie_action / update_bull / portal_damage and relies on border walls).
  static SF_DEV uint32_t flags_at(const uint8_t *lds, const Params &p, int f, int r, int c) {
    if (!inmap(p, r, c)) return SF_CELL_WALL;
    return W::ulds_u8(lds, cellidx(p, f, r, c));
  }
  static SF_DEV int human_at(const Arena &S, uint32_t q) {  // the human designated by the cell's s[0]
    uint64_t m = W::ballot(((S.hfl & HF_OCC) != 0u) & (S.hpos == q));
    return m ? W::ctz64(m) : -1;
  }
  static SF_DEV int live_human_at(const Arena &S, uint32_t q) {
    uint64_t m = W::ballot(((S.hfl & HF_ALIVE) != 0u) & (S.hpos == q));
    return m ? W::ctz64(m) : -1;
  }
  // ---- the zombie table: one lane per slot in registers, or (ZL) [field][slot] in LDS ------------------------------
  // word j = slots 64 j .. 64 j + 63, one per lane
  static SF_DEV V zl_get(const Arena &S, int f, uint32_t j) { return W::lds_u32(S.zl + (uint32_t)f * S.zcap + 64u * j, W::lane(), W::all()); }
  static SF_DEV void zl_put(Arena &S, int f, uint32_t j, const V &v, P pred) {
    W::lds_store_u32(S.zl + (uint32_t)f * S.zcap + 64u * j, W::lane(), v, pred);
  }
  // field f of slot i, wave-uniform
  static SF_DEV uint32_t z_read(const Arena &S, int f, uint32_t i) {
    if constexpr (ZL) return W::ulds_u32(S.zl + (uint32_t)f * S.zcap, i);
    return W::readlane(f == ZW_POS ? S.zpos : f == ZW_HP ? S.zhp : S.zmd, i);
  }
  static SF_DEV void z_write(Arena &S, int f, uint32_t i, uint32_t v) {
    if constexpr (ZL) {
      W::ulds_store_u32(S.zl + (uint32_t)f * S.zcap, i, v);
    } else {
      W::setlane(f == ZW_POS ? S.zpos : f == ZW_HP ? S.zhp : S.zmd, i, v);
    }
  }
  static SF_DEV int zombie_at(const Arena &S, uint32_t q) {
    if constexpr (ZL) {
      for (uint32_t j = 0; j < S.zwn; ++j) {
        const uint64_t m = W::ballot((zl_get(S, ZW_POS, j) & (ZF_ALIVE | POS_MASK)) == (ZF_ALIVE | q));
        if (m) return (int)(64u * j) + W::ctz64(m);
      }
      return -1;
    }
    uint64_t m = W::ballot((S.zpos & (ZF_ALIVE | POS_MASK)) == (ZF_ALIVE | q));
    return m ? W::ctz64(m) : -1;
  }
  static SF_DEV int refbullet_at(const Arena &S, uint32_t q) {  // the bullet designated by the cell's s[2]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint64_t m = W::ballot((S.ba[j] & (BA_REF | POS_MASK)) == (BA_REF | q));
      if (m) return j * 64 + W::ctz64(m);
    }
    return -1;
  }
  // node::showit() G:321-346 as a class code
  static SF_DEV int showit(const Arena &S, const uint8_t *lds, const Params &p, int f, int r, int c, uint32_t &fl) {
    fl = flags_at(lds, p, f, r, c);
    if (fl & SF_CELL_WALL) return SH_WALL;
    const uint32_t q = pos_pack(f, r, c);
    if (human_at(S, q) >= 0) return SH_HUMAN;
    if (zombie_at(S, q) >= 0) return SH_ZOMBIE;
    if (fl & SF_CELL_PIN_UP) return SH_PUP;
    if (fl & SF_CELL_PIN_DN) return SH_PDN;
    if (refbullet_at(S, q) >= 0) return SH_BULLET;
    if (fl & SF_CELL_CHEST) return SH_CHEST;
    if (fl & SF_CELL_POUT) return SH_POUT;
    return SH_EMPTY;
  }
  static SF_DEV int showit_q(const Arena &S, const uint8_t *lds, const Params &p, uint32_t q, uint32_t &fl) {
    return showit(S, lds, p, pos_f(q), pos_r(q), pos_c(q), fl);
  }
  // `(sit != '#' && sit != 'v' && sit != '^') || s[10]`  G:812,1086
  static SF_DEV bool bullet_may_enter(int sit, uint32_t fl) {
    return (sit != SH_WALL && sit != SH_PDN && sit != SH_PUP) || (fl & SF_CELL_TEMP);
  }

  // ------------------------------------------------------------------------------------------------
  // slot allocators G:209-235
  static SF_DEV uint64_t capmask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
  static SF_DEV int b_ind(const Arena &S, const Params &p) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int left = p.B - 64 * j;
      if (left > 0) {
        uint64_t fr = ~W::ballot((S.ba[j] & BA_ALIVE) != 0u) & capmask(left);
        if (fr) return j * 64 + W::ctz64(fr);
      }
    }
    return -1;
  }
  static SF_DEV int z_ind(const Arena &S, const Params &p) {
    if constexpr (ZL) {
      for (uint32_t j = 0; j < S.zwn; ++j) {
        const uint64_t fr = ~W::ballot((zl_get(S, ZW_POS, j) & ZF_ALIVE) != 0u) & capmask(p.Z - (int)(64u * j));
        if (fr) return (int)(64u * j) + W::ctz64(fr);
      }
      return S.zwn < (uint32_t)zw_for(p.Z) ? (int)(64u * S.zwn) : -1;  // the first slot of a word not in use yet (z_take opens it)
    }
    uint64_t fr = ~W::ballot((S.zpos & ZF_ALIVE) != 0u) & capmask(p.Z);
    return fr ? W::ctz64(fr) : -1;
  }
  // ---- the exit table: one lane per exit in registers, or (ZL) pl[slot] in LDS ------------------------------------
  static SF_DEV V pl_get(const Arena &S, uint32_t j) { return W::lds_u32(S.pl + 64u * j, W::lane(), W::all()); }
  static SF_DEV uint32_t p_read(const Arena &S, uint32_t i) {
    if constexpr (ZL) return (i >> 6) < S.pwn ? W::ulds_u32(S.pl, i) : 0u;  // (an exit number the map never defined: inactive)
    return W::readlane(S.ppos, i);
  }
  static SF_DEV void p_write(Arena &S, uint32_t i, uint32_t v) {
    if constexpr (ZL) {
      if ((i >> 6) >= S.pwn) {  // a word not in use yet is opened with every exit inactive
        W::lds_store_u32(S.pl + 64u * S.pwn, W::lane(), V(0u), W::all());
        ++S.pwn;
        if (S.pwhi < S.pwn) S.pwhi = S.pwn;
      }
      W::ulds_store_u32(S.pl, i, v);
    } else {
      W::setlane(S.ppos, i, v);
    }
  }
  // ZL: slot i is about to be filled; a word that was not in use is opened with every slot dead
  static SF_DEV void z_take(Arena &S, uint32_t i) {
    if constexpr (ZL) {
      if ((i >> 6) >= S.zwn) {
        zl_put(S, ZW_POS, S.zwn, V(0u), W::all());
        zl_put(S, ZW_HP, S.zwn, V(0u), W::all());
        zl_put(S, ZW_MINDAMAGE, S.zwn, V(0u), W::all());
        ++S.zwn;
        if (S.zwhi < S.zwn) S.zwhi = S.zwn;
      }
    }
  }
  static SF_DEV int h_ind(const Arena &S, const Params &p) {  // skips `ind` and remote slots
    uint64_t fr = ~W::ballot((S.hfl & (HF_ALIVE | HF_REMOTE)) != 0u) & capmask(p.H) & ~(1ull << p.ind);
    return fr ? W::ctz64(fr) : -1;
  }
  static SF_DEV int p_ind(const Arena &S, const Params &p) {
    if constexpr (ZL) {
      for (uint32_t j = 0; j < S.pwn; ++j) {
        const uint64_t fr = ~W::ballot((pl_get(S, j) & PF_ACTIVE) != 0u) & capmask(p.P - (int)(64u * j));
        if (fr) return (int)(64u * j) + W::ctz64(fr);
      }
      return S.pwn < (uint32_t)zw_for(p.P) ? (int)(64u * S.pwn) : -1;  // (p_write opens the word)
    }
    uint64_t fr = ~W::ballot((S.ppos & PF_ACTIVE) != 0u) & capmask(p.P);
    return fr ? W::ctz64(fr) : -1;
  }

  // Activate bullet slot `idx` on cell q and make it the cell's designated bullet (Bullet::shot IT:156-163
  // + `pix->bullet = &bull[index]; pix->s[2] = 1; mb[index] = true`).  A bullet already designated there
  // is orphaned.
  static SF_DEV void bullet_put(Arena &S, int idx, uint32_t q, int way, int dmg, int eff, int range, int owner) {
    const uint32_t l = (uint32_t)idx & 63u;
    const uint32_t a = q | ((uint32_t)(way - 1) << BA_WAY_SH) | BA_ALIVE | BA_REF;
    const uint32_t b = ((uint32_t)eff & 0xffffu) | ((uint32_t)owner << 16);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      S.ba[j] = W::select((S.ba[j] & (BA_REF | POS_MASK)) == (BA_REF | q), S.ba[j] & ~BA_REF, S.ba[j]);
      if (j == (idx >> 6)) {
        W::setlane(S.ba[j], l, a);
        W::setlane(S.bd[j], l, (uint32_t)dmg);
        W::setlane(S.bb[j], l, b);
        W::setlane(S.bc[j], l, (uint32_t)range);
      }
    }
  }
  // ------------------------------------------------------------------------------------------------
  // human field helpers (wave-uniform access to slot i)
  static SF_DEV int h_way(uint32_t fl) { return (int)(fl & HF_WAY_MASK); }
  static SF_DEV int h_team(uint32_t fl) { return (int)((fl >> HF_TEAM_SH) & 255u); }
  static SF_DEV int h_vec(uint32_t fl) { return (int)((fl >> HF_VEC_SH) & 3u) - 1; }
  static SF_DEV int h_sel(uint32_t fl) { return (int)((fl >> HF_IND_SH) & 15u) - 1; }
  // the table block of the record this human was built from: the NPC record, or the commanded human's own
  static SF_DEV int h_block(const Params &p, uint32_t fl) { return (fl & HF_PROF) ? p.npc_block : (int)((fl >> HF_AGP_SH) & 15u); }
  static SF_DEV uint32_t set_vec_sel(uint32_t fl, int vec, int sel) {
    fl &= ~((3u << HF_VEC_SH) | (15u << HF_IND_SH));
    return fl | ((uint32_t)(vec + 1) << HF_VEC_SH) | ((uint32_t)(sel + 1) << HF_IND_SH);
  }
  static SF_DEV uint32_t get16(const V &lo, const V &hi, uint32_t lane, int k) {  // k in 0..3
    // both halves are read and one is picked: selecting the *register* by a run-time k would make the
    // compiler spill the pair to scratch memory as an indexable array
    const uint32_t wl = W::readlane(lo, lane), wh = W::readlane(hi, lane);
    const uint32_t w = k < 2 ? wl : wh;
    return (w >> ((k & 1) * 16)) & 0xffffu;
  }
  static SF_DEV void set16(V &lo, V &hi, uint32_t lane, int k, uint32_t val) {
    const int sh = (k & 1) * 16;
    const uint32_t wl = W::readlane(lo, lane), wh = W::readlane(hi, lane);
    const uint32_t w = k < 2 ? wl : wh;
    const uint32_t nw = (w & ~(0xffffu << sh)) | ((val & 0xffffu) << sh);
    W::setlane(lo, lane, k < 2 ? nw : wl);  // both registers are rewritten (see get16)
    W::setlane(hi, lane, k < 2 ? wh : nw);
  }
  static SF_DEV void add_lane(V &v, uint32_t lane, int32_t d) {
    W::setlane(v, lane, W::readlane(v, lane) + (uint32_t)d);
  }

  // Build a fresh human of profile `prof` in slot i (Human::build CH:650-709 / gen_human CH:873-888).
  // prof 1: the NPC record; prof 0: a commanded human's record (agent record `agp` when every agent has its own)
  static SF_DEV void human_make(Arena &S, const Params &p, uint32_t i, int prof, uint32_t q, int way, int team,
                                uint32_t extra_flags, int agp = 0) {
    const Derived &d = p.tab->der[prof ? p.npc_block : agp];
    extra_flags |= prof ? 0u : ((uint32_t)agp << HF_AGP_SH);
    W::setlane(S.hpos, i, q);
    uint32_t fl = (uint32_t)way | ((uint32_t)team << HF_TEAM_SH) | HF_ALIVE | HF_OCC | (prof ? HF_PROF : 0u) | extra_flags;
    W::setlane(S.hfl, i, set_vec_sel(fl, -1, -1));
    W::setlane(S.hhp, i, (uint32_t)d.hp);
    W::setlane(S.hst, i, (uint32_t)d.stamina);
    W::setlane(S.hmd, i, (uint32_t)d.mindamage);
    W::setlane(S.hk, i, 0u), W::setlane(S.hdm, i, 0u), W::setlane(S.hef, i, 0u);
    W::setlane(S.hc01, i, (uint32_t)d.cons[0] | ((uint32_t)d.cons[1] << 16));
    W::setlane(S.hc23, i, (uint32_t)d.cons[2] | ((uint32_t)d.cons[3] << 16));
    W::setlane(S.ht01, i, (uint32_t)d.thr_cnt[0] | ((uint32_t)d.thr_cnt[1] << 16));
    W::setlane(S.ht23, i, (uint32_t)d.thr_cnt[2] | ((uint32_t)d.thr_cnt[3] << 16));
    W::setlane(S.hbpk, i, (uint32_t)d.blocks | ((uint32_t)d.portals << 8));  // portal_ind = -1
    W::setlane(S.hcmd, i, (uint32_t)'+');
  }

  // ------------------------------------------------------------------------------------------------
  // spawns G:532-572 (loop top G:1444-1449)
  // Code-size note: this kernel's hot code has to stay resident in a 64 KiB instruction cache shared by two CUs
  // while 32 wavefronts walk different parts of it, and every inlined draw() is ~0.4 KiB.  Phases that the
  // reference writes out several times (three spawns, two half-ticks, three coordinate draws) are therefore kept as
  // ONE body inside a non-unrolled loop; the order of draws is unchanged.
#define SF_NOUNROLL _Pragma("clang loop unroll(disable)")

  // `rand() % F, rand() % N, rand() % M` in that order (G:534,546,561,1850) as a packed position
  static SF_DEV uint32_t draw_cell(Arena &S, const uint8_t *lds, const Params &p) {
    uint32_t q = 0u;
    SF_NOUNROLL for (int i = 0; i < 3; ++i) {
      const uint32_t m = (uint32_t)(i == 0 ? p.F : i == 1 ? p.N : p.M);
      q = (q << 10) | (draw(S, lds, p) % m);
    }
    return q;  // ((f << 10) | r) << 10 | c == pos_pack(f, r, c)
  }

  // spawn_chest / spawn_zombie_npc / spawn_human_npc G:532-572, at the loop top when frame % {30,40,50} <= 1
  // (G:1444-1449), in that order
  static SF_DEV void spawns(Arena &S, uint8_t *lds, const Params &p) {
    const uint32_t due = (S.frame % 30 <= 1 ? 1u : 0u) | (S.frame % 40 <= 1 ? 2u : 0u) | (S.frame % 50 <= 1 ? 4u : 0u);
    if (!due) return;
    SF_NOUNROLL for (int kind = 0; kind < 3; ++kind) {
      if (!((due >> kind) & 1u)) continue;
      if (kind == 0 && p.C <= S.chests) continue;  // `if(C <= chest) return;` before any draw
      const uint32_t q = draw_cell(S, lds, p);
      uint32_t fl;
      if (showit_q(S, lds, p, q, fl) != SH_EMPTY) continue;
      int index = 0;
      if (kind == 1) index = z_ind(S, p);
      if (kind == 2) index = h_ind(S, p);
      if (index == -1) continue;
      if (kind == 2) {
        human_make(S, p, (uint32_t)index, 1, q, 1, 0, HF_RNPC);  // gen_human CH:873-888
        continue;
      }
      const uint32_t d4 = draw(S, lds, p) % 4u;
      if (kind == 0) {  // `cons = gen_item(rand() % 4)`
        W::ulds_store_u8(lds, cellidx_q(p, q), fl | SF_CELL_CHEST | (d4 << SF_CELL_CONS_SHIFT));
        S.dirty = 1u;
        ++S.chests;
      } else {  // `super = (rand() % 4 == 0)`, Zombie::gen_npc CH:850-857
        const uint32_t super_ = d4 == 0u ? 1u : 0u;
        z_take(S, (uint32_t)index);
        z_write(S, ZW_POS, (uint32_t)index, q | ZF_ALIVE | (super_ ? ZF_SUPER : 0u));
        z_write(S, ZW_HP, (uint32_t)index, (super_ + 1u) * 400u);
        z_write(S, ZW_MINDAMAGE, (uint32_t)index, (super_ + 1u) * 100u);
      }
    }
  }

  // ------------------------------------------------------------------------------------------------
  // zombie_action G:654-693.  What a zombie may do depends on: a designated bullet on its own cell (skip), a
  // human on a neighbour cell (punch instead of moving), and for a move the target being '.', i.e. a clear flag
  // byte with no human, zombie or designated bullet on it.  Humans, flags and pre-existing bullets do not change
  // during the phase and the punches it creates land on human cells, which border only zombies that do not move;
  // so everything except "is another zombie there now" is computed once, lane-parallel, one lane per zombie, and
  // the slot-ordered loop that fixes the RNG draw order only tests bits, draws, and asks one ballot per move.
  static SF_DEV void zombie_action(Arena &S, uint8_t *lds, const Params &p) {
    if constexpr (ZL) {
      zombie_action_zl(S, lds, p);
      return;
    }
    SF_PROF(PH_ZOMBIE);
    const P zalive = ((S.zpos & ZF_ALIVE) != 0u) & W::ltu(W::lane(), (uint32_t)p.Z);
    uint64_t zm = W::ballot(zalive);
    if (!zm) return;
    const V zq = S.zpos & POS_MASK;
    const V zr = (zq >> 10) & 1023u, zc = zq & 1023u;
    // neighbour d of each zombie: packed position and "clear flag byte" bit
    V freebits = V(0u);
    V hnear = V(0u);  // bit d: a human (cell designation s[0]) stands on neighbour d
    const V ci0 = ((zq >> 20) * (uint32_t)p.N + zr) * (uint32_t)p.M + zc;  // the zombie's own cell index
    const V qn0 = zq + 1024u, qn1 = zq + 1u, qn2 = zq - 1024u, qn3 = zq - 1u;  // DX/DY order: down, right, up, left
    uint64_t skip = 0ull;
    // the four neighbours' flag bytes first, all four loads in flight at once (with the plane in HBM each is an L2 round
    // trip): a neighbour off the map reads the zombie's own cell instead and is masked afterwards
    V nfl[4];
    P ninb[4];
    for (int d = 0; d < 4; ++d) {
      const V rr = zr + (uint32_t)DX(d), cc = zc + (uint32_t)DY(d);
      ninb[d] = zalive & W::ltu(rr, (uint32_t)p.N) & W::ltu(cc, (uint32_t)p.M);
      nfl[d] = W::lds_u8_any(lds, W::select(ninb[d], ci0 + (uint32_t)(DX(d) * p.M + DY(d)), W::select(zalive, ci0, V(0u))));
    }
    const bool use_bm = BITMAPS;
    if (use_bm) {
      // humans and designated bullets scatter their cells into bitmaps; every zombie tests its own cell (a bullet
      // there: skip) and its four neighbours (a bullet: not '.'; a human: punch)
      const P hocc = (S.hfl & HF_OCC) != 0u;
      const V hci = cell_index_v(p, S.hpos);
      bm_set(S, p, BM_HUM, hci, hocc);
      bm_bullets(S, p, BM_REF, true);
      skip = W::ballot(zalive & bm_test(S, p, BM_REF, ci0, zalive));
      for (int d = 0; d < 4; ++d) {
        const P inb = ninb[d];
        const V ci = ci0 + (uint32_t)(DX(d) * p.M + DY(d));
        const V fl = nfl[d];
        const P clear = inb & (fl == 0u) & (!bm_test(S, p, BM_REF, ci, inb));
        freebits = freebits | W::select(clear, V(1u << d), V(0u));
        hnear = hnear | W::select(inb & bm_test(S, p, BM_HUM, ci, inb), V(1u << d), V(0u));
      }
      bm_clear(S, p, BM_HUM, hci, hocc);
      bm_bullets(S, p, BM_REF, false);
    } else {
      for (int d = 0; d < 4; ++d) freebits = freebits | W::select(ninb[d] & (nfl[d] == 0u), V(1u << d), V(0u));
      // designated bullets present at phase start: own cell -> skip; neighbour cell -> not '.'
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        uint64_t bm = W::ballot((S.ba[j] & BA_REF) != 0u);
        while (bm) {
          const uint32_t l = (uint32_t)W::ctz64(bm);
          bm &= bm - 1ull;
          const uint32_t q = W::readlane(S.ba[j], l) & POS_MASK;
          skip |= W::ballot(zalive & (zq == q));
          const V hit = W::select(qn0 == q, V(1u), V(0u)) | W::select(qn1 == q, V(2u), V(0u)) |
                        W::select(qn2 == q, V(4u), V(0u)) | W::select(qn3 == q, V(8u), V(0u));
          freebits = freebits & ~hit;
        }
      }
      // humans next to a zombie (the packed compare can alias across a row end: the loop below re-checks exactly)
      uint64_t hm = W::ballot((S.hfl & HF_OCC) != 0u);
      while (hm) {
        const uint32_t h = (uint32_t)W::ctz64(hm);
        hm &= hm - 1ull;
        const uint32_t q = W::readlane(S.hpos, h);
        hnear = hnear | W::select(zalive & ((qn0 == q) | (qn1 == q) | (qn2 == q) | (qn3 == q)), V(15u), V(0u));
      }
    }
    // Two loops instead of one.  A zombie next to a human punches and draws nothing (G:664-677); every other zombie draws
    // and may move (G:678-690).  The two kinds do not see each other's effects: punches land on human cells — never a
    // move's target, which must show '.' — and take bullet slots, which movers do not; movers change zombie cells, which
    // punches do not look at; and who is next to a human is fixed for the phase (humans stand still, a zombie acts from
    // where it started).  So the punches of all such zombies, in slot order, can come first, and the draw loop — the hot
    // one: the reference's draw order is its slot order — carries neither their code nor the tests for it.
    SF_STAMP(S, 13);  // (diagnostic build: the pre-computation above is charged to its own phase)
    const uint64_t todo = zm & ~skip;  // `if(themap[i][j][k].s[2]) continue;`
    uint64_t nearm = W::ballot(hnear != 0u) & todo;
    uint64_t falls = 0ull;  // "near" by the packed compare but no human there (row-end aliasing): they draw like the others
    while (nearm) {
      const uint32_t z = (uint32_t)W::ctz64(nearm);
      const uint64_t bit = nearm & (0ull - nearm);
      nearm ^= bit;
      // exact neighbour scan (the packed compare above can alias across a row end; this cannot)
      const uint32_t q0 = W::readlane(zq, z);
      const int f = pos_f(q0), r = pos_r(q0), c = pos_c(q0);
      const int zmd = (int)W::readlane(S.zmd, z);
      const uint32_t hn = W::readlane(hnear, z);
      bool b = false;
      for (int i1 = 0; i1 < 4; ++i1) {
        if (!((hn >> i1) & 1u)) continue;
        const int rr = r + DX(i1), cc = c + DY(i1);
        if (!inmap(p, rr, cc)) continue;
        const uint32_t q = pos_pack(f, rr, cc);
        if (use_bm || human_at(S, q) >= 0) {
          int index = b_ind(S, p);
          if (refbullet_at(S, q) < 0 && index != -1)  // Zombie::punch CH:838-844
            bullet_put(S, index, q, i1 + 1, zmd > 0 ? zmd : 0, 0, 1, 0);
          b = true;
        }
      }
      if (!b) falls |= bit;
    }
    uint64_t movers = (todo & ~W::ballot(hnear != 0u)) | falls;
#ifdef SF_EXP_ZLOOP_FREE
    // timing build only (round 4): the draw loop keeps its draws (one, two or three per zombie, in about the real
    // proportions) and loses everything else — slot lookup, free-cell test, "is a zombie there now", the move: the bound
    // of ANY restructuring of the per-zombie decision logic (lane-parallel speculation over the draw stream included).
    // Nobody moves: wrong by construction; never the product library.
    for (int nz_ = W::popc64(movers); nz_ > 0; --nz_) {
      if (mod5(draw(S, lds, p)) < 2u) continue;
      if (draw(S, lds, p) & 1u) continue;
      (void)draw(S, lds, p);
    }
    movers = 0ull;
#endif
    while (movers) {
      const uint32_t z = (uint32_t)W::ctz64(movers);
      movers &= movers - 1ull;
      if (mod5(draw(S, lds, p)) < 2u) continue;
      const uint32_t fb = W::readlane(freebits, z);
      const uint32_t zp = W::readlane(S.zpos, z);
      SF_NOUNROLL for (int i1 = 0; i1 < 2; ++i1) {
        const uint32_t i2 = draw(S, lds, p) & 3u;
        if (!((fb >> i2) & 1u)) continue;
        // +1024, +1, -1024, -1 for directions 0..3 (DX / DY), out of one packed constant: no branches
        const uint32_t q = (zp & POS_MASK) + (uint32_t)(int32_t)(int16_t)(0xFFFFFC0000010400ull >> (i2 * 16u));
        // a human cannot be there: a zombie with a human neighbour never reaches this point
        if (W::ballot((S.zpos & (ZF_ALIVE | POS_MASK)) == (ZF_ALIVE | q)) != 0ull) continue;
        W::setlane(S.zpos, z, (zp & ~POS_MASK) | q);
        break;
      }
    }
  }


  // The same phase over a zombie table in LDS (ZL), one 64-slot word after the other.  Within a word the scheme above
  // applies unchanged (punches first, then the draw loop); between words the reference's slot order is kept as it is.
  // That the words may be evaluated one after the other against the state as it is then — instead of everything against
  // the phase's start — follows from the same facts: humans and flags do not change; a punch lands on a human cell,
  // which is no zombie's own cell and no mover's target (a zombie next to a human punches), so neither `skip` nor a
  // mover's choice can see it; only "is another zombie there now" depends on earlier zombies, and it is asked at the
  // zombie's turn, against the table itself.  The humans' and designated bullets' bitmaps are therefore built once, as of
  // the phase's start, and serve every word.
  static SF_DEV void zombie_action_zl(Arena &S, uint8_t *lds, const Params &p) {
    SF_PROF(PH_ZOMBIE);
    if (!S.zwn) return;
    const P hocc = (S.hfl & HF_OCC) != 0u;
    const V hci = cell_index_v(p, S.hpos);
    if (BITMAPS) {
      bm_set(S, p, BM_HUM, hci, hocc);
      bm_bullets(S, p, BM_REF, true);
    }
    for (uint32_t j = 0; j < S.zwn; ++j) {
      const V zpw = zl_get(S, ZW_POS, j);
      const P zalive = (zpw & ZF_ALIVE) != 0u;
      const uint64_t zm = W::ballot(zalive);
      if (!zm) continue;
      const V zq = zpw & POS_MASK;
      const V zr = (zq >> 10) & 1023u, zc = zq & 1023u;
      V freebits = V(0u), hnear = V(0u);
      const V ci0 = ((zq >> 20) * (uint32_t)p.N + zr) * (uint32_t)p.M + zc;
      const V qn0 = zq + 1024u, qn1 = zq + 1u, qn2 = zq - 1024u, qn3 = zq - 1u;
      uint64_t skip = 0ull;
      V nfl[4];
      P ninb[4];
      for (int d = 0; d < 4; ++d) {
        const V rr = zr + (uint32_t)DX(d), cc = zc + (uint32_t)DY(d);
        ninb[d] = zalive & W::ltu(rr, (uint32_t)p.N) & W::ltu(cc, (uint32_t)p.M);
        nfl[d] = W::lds_u8_any(lds, W::select(ninb[d], ci0 + (uint32_t)(DX(d) * p.M + DY(d)), W::select(zalive, ci0, V(0u))));
      }
      if (BITMAPS) {
        skip = W::ballot(zalive & bm_test(S, p, BM_REF, ci0, zalive));
        for (int d = 0; d < 4; ++d) {
          const P inb = ninb[d];
          const V ci = ci0 + (uint32_t)(DX(d) * p.M + DY(d));
          const P clear = inb & (nfl[d] == 0u) & (!bm_test(S, p, BM_REF, ci, inb));
          freebits = freebits | W::select(clear, V(1u << d), V(0u));
          hnear = hnear | W::select(inb & bm_test(S, p, BM_HUM, ci, inb), V(1u << d), V(0u));
        }
      } else {
        for (int d = 0; d < 4; ++d) freebits = freebits | W::select(ninb[d] & (nfl[d] == 0u), V(1u << d), V(0u));
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          uint64_t bm = W::ballot((S.ba[jb] & BA_REF) != 0u);
          while (bm) {
            const uint32_t l = (uint32_t)W::ctz64(bm);
            bm &= bm - 1ull;
            const uint32_t q = W::readlane(S.ba[jb], l) & POS_MASK;
            skip |= W::ballot(zalive & (zq == q));
            const V hit = W::select(qn0 == q, V(1u), V(0u)) | W::select(qn1 == q, V(2u), V(0u)) |
                          W::select(qn2 == q, V(4u), V(0u)) | W::select(qn3 == q, V(8u), V(0u));
            freebits = freebits & ~hit;
          }
        }
        uint64_t hm = W::ballot(hocc);
        while (hm) {
          const uint32_t h = (uint32_t)W::ctz64(hm);
          hm &= hm - 1ull;
          const uint32_t q = W::readlane(S.hpos, h);
          hnear = hnear | W::select(zalive & ((qn0 == q) | (qn1 == q) | (qn2 == q) | (qn3 == q)), V(15u), V(0u));
        }
      }
      const uint64_t todo = zm & ~skip;
      uint64_t nearm = W::ballot(hnear != 0u) & todo;
      uint64_t falls = 0ull;
      while (nearm) {
        const uint32_t z = (uint32_t)W::ctz64(nearm);
        const uint64_t bit = nearm & (0ull - nearm);
        nearm ^= bit;
        const uint32_t q0 = W::readlane(zq, z);
        const int f = pos_f(q0), r = pos_r(q0), c = pos_c(q0);
        const int zmd = (int)z_read(S, ZW_MINDAMAGE, 64u * j + z);
        const uint32_t hn = W::readlane(hnear, z);
        bool b = false;
        for (int i1 = 0; i1 < 4; ++i1) {
          if (!((hn >> i1) & 1u)) continue;
          const int rr = r + DX(i1), cc = c + DY(i1);
          if (!inmap(p, rr, cc)) continue;
          const uint32_t q = pos_pack(f, rr, cc);
          if (BITMAPS || human_at(S, q) >= 0) {
            int index = b_ind(S, p);
            if (refbullet_at(S, q) < 0 && index != -1) bullet_put(S, index, q, i1 + 1, zmd > 0 ? zmd : 0, 0, 1, 0);
            b = true;
          }
        }
        if (!b) falls |= bit;
      }
      uint64_t movers = (todo & ~W::ballot(hnear != 0u)) | falls;
      while (movers) {
        const uint32_t z = (uint32_t)W::ctz64(movers);
        movers &= movers - 1ull;
        if (mod5(draw(S, lds, p)) < 2u) continue;
        const uint32_t fb = W::readlane(freebits, z);
        const uint32_t zp = W::readlane(zpw, z);  // (its own entry of the table: nobody else has written it)
        SF_NOUNROLL for (int i1 = 0; i1 < 2; ++i1) {
          const uint32_t i2 = draw(S, lds, p) & 3u;
          if (!((fb >> i2) & 1u)) continue;
          const uint32_t q = (zp & POS_MASK) + (uint32_t)(int32_t)(int16_t)(0xFFFFFC0000010400ull >> (i2 * 16u));
          if (zombie_at(S, q) >= 0) continue;  // the table as it is now: earlier zombies have moved
          z_write(S, ZW_POS, 64u * j + z, (zp & ~POS_MASK) | q);
          break;
        }
      }
    }
    if (BITMAPS) {
      bm_clear(S, p, BM_HUM, hci, hocc);
      // (the punches above have designated new bullets and orphaned old ones: the whole bitmap, not the present bullets' words)
      W::lds_zero(S.bm + BM_REF * p.bm_words, (uint32_t)p.bm_words);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // portal_damage G:1279-1297
  static SF_DEV void portal_damage(Arena &S, uint8_t *lds, const Params &p) {
    if constexpr (ZL) {
      portal_damage_zl(S, lds, p);
      return;
    }
    SF_PROF(PH_PORTAL);
    const uint64_t pm = W::ballot((S.ppos & PF_ACTIVE) != 0u) & capmask(p.P);
    if (!pm) return;
    // an exit radiates unless it shows 'O': 'O' flag, no wall / entrance / chest flag, nobody and no designated
    // bullet on it (showit order G:321-346).  All exits at once: their flag bytes by one gather (one L2 round trip
    // for a plane kept in HBM, instead of one per exit), occupancy by the cell bitmaps or, for a single exit and on
    // maps without bitmaps, by three ballots per exit
    const P act = ((S.ppos & PF_ACTIVE) != 0u) & W::ltu(W::lane(), (uint32_t)p.P);
    const V pci = cell_index_v(p, S.ppos);
    const V fl = W::lds_u8(lds, pci, act);
    const P plain = act & ((fl & (uint32_t)(SF_CELL_WALL | SF_CELL_PIN_UP | SF_CELL_PIN_DN | SF_CELL_CHEST | SF_CELL_POUT)) ==
                           (uint32_t)SF_CELL_POUT);
    uint64_t need = pm;
    uint64_t plm = W::ballot(plain);
    if (BITMAPS && (plm & (plm - 1ull))) {
      const P hocc = (S.hfl & HF_OCC) != 0u, zlive = (S.zpos & ZF_ALIVE) != 0u;
      const V hci = cell_index_v(p, S.hpos), zci = cell_index_v(p, S.zpos);
      bm_set(S, p, BM_HUM, hci, hocc);
      if constexpr (ZL) bm_zombies_zl(S, p, BM_ZOM, true); else bm_set(S, p, BM_ZOM, zci, zlive);
      bm_bullets(S, p, BM_REF, true);
      const P covered = bm_test(S, p, BM_HUM, pci, plain) | bm_test(S, p, BM_ZOM, pci, plain) | bm_test(S, p, BM_REF, pci, plain);
      bm_clear(S, p, BM_HUM, hci, hocc);
      if constexpr (ZL) bm_zombies_zl(S, p, BM_ZOM, false); else bm_clear(S, p, BM_ZOM, zci, zlive);
      bm_bullets(S, p, BM_REF, false);
      need = pm & ~W::ballot(plain & (!covered));
    } else {
      while (plm) {
        const uint32_t i = (uint32_t)W::ctz64(plm);
        const uint64_t bit = plm & (0ull - plm);
        plm ^= bit;
        const uint32_t q = W::readlane(S.ppos, i) & POS_MASK;
        if (human_at(S, q) < 0 && zombie_at(S, q) < 0 && refbullet_at(S, q) < 0) need &= ~bit;
      }
    }
    while (need) {
      const uint32_t i = (uint32_t)W::ctz64(need);
      need &= need - 1ull;
      int index = b_ind(S, p);
      if (index == -1) return;
      bullet_put(S, index, W::readlane(S.ppos, i) & POS_MASK, 3, 20, -10, 1, 0);  // radiation.ready(20, -10, 1); shot(v, 3, radiation, 0)
    }
  }


  // The same phase over an exit table in LDS (ZL), word by word in slot order.  Whether an exit is covered does not
  // depend on the radiation bullets of earlier exits (they land on those exits' own cells), so each word is evaluated
  // when its turn comes; the occupancy bitmaps are built once, at the first word that has a plain exit.
  static SF_DEV void portal_damage_zl(Arena &S, uint8_t *lds, const Params &p) {
    SF_PROF(PH_PORTAL);
    const P hocc = (S.hfl & HF_OCC) != 0u;
    const V hci = cell_index_v(p, S.hpos);
    bool built = false, dry = false;
    for (uint32_t j = 0; j < S.pwn && !dry; ++j) {
      const V ppw = pl_get(S, j);
      const P act = (ppw & PF_ACTIVE) != 0u;
      const uint64_t pm = W::ballot(act);
      if (!pm) continue;
      const V pci = cell_index_v(p, ppw);
      const V fl = W::lds_u8(lds, pci, act);
      const P plain = act & ((fl & (uint32_t)(SF_CELL_WALL | SF_CELL_PIN_UP | SF_CELL_PIN_DN | SF_CELL_CHEST | SF_CELL_POUT)) ==
                             (uint32_t)SF_CELL_POUT);
      uint64_t need = pm;
      uint64_t plm = W::ballot(plain);
      if (BITMAPS && plm) {
        if (!built) {
          bm_set(S, p, BM_HUM, hci, hocc);
          bm_zombies_zl(S, p, BM_ZOM, true);
          bm_bullets(S, p, BM_REF, true);
          built = true;
        }
        const P covered = bm_test(S, p, BM_HUM, pci, plain) | bm_test(S, p, BM_ZOM, pci, plain) | bm_test(S, p, BM_REF, pci, plain);
        need = pm & ~W::ballot(plain & (!covered));
      } else {
        while (plm) {
          const uint32_t i = (uint32_t)W::ctz64(plm);
          const uint64_t bit = plm & (0ull - plm);
          plm ^= bit;
          const uint32_t q = W::readlane(ppw, i) & POS_MASK;
          if (human_at(S, q) < 0 && zombie_at(S, q) < 0 && refbullet_at(S, q) < 0) need &= ~bit;
        }
      }
      while (need) {
        const uint32_t i = (uint32_t)W::ctz64(need);
        need &= need - 1ull;
        int index = b_ind(S, p);
        if (index == -1) {  // `return;` G:1289
          dry = true;
          break;
        }
        bullet_put(S, index, W::readlane(ppw, i) & POS_MASK, 3, 20, -10, 1, 0);
      }
    }
    if (built) {
      bm_clear(S, p, BM_HUM, hci, hocc);
      bm_zombies_zl(S, p, BM_ZOM, false);
      W::lds_zero(S.bm + BM_REF * p.bm_words, (uint32_t)p.bm_words);  // (the radiation bullets took over their cells' designation)
    }
  }

  // ------------------------------------------------------------------------------------------------
  // update_tmp G:1343-1381.  Bullets standing on destructible '#' / '^' cells are absorbed; objects whose
  // accumulated damage crossed the limit are removed.  Only a cell that absorbed a bullet in this call can
  // cross its limit in this call, so the reference's sweep over `temp` reduces to those cells.
  static SF_DEV void break_if_spent(Arena &S, uint8_t *lds, const Params &p, int a, uint32_t q) {
    uint32_t fl;
    const int sit = showit_q(S, lds, p, q, fl);
    if (!(fl & SF_CELL_TEMP)) return;
    const uint32_t ci = cellidx_q(p, q);
    int32_t *dmg = p.aux_dmg + (size_t)a * (size_t)p.cells;
    const int32_t d = W::uload_i32(dmg + ci);
    if (sit == SH_PUP && d >= LIM_PORTAL) {
      const int i = (int)W::uload_i16(p.aux_pidx + (size_t)a * (size_t)p.cells + ci);
      const uint32_t e1 = p_read(S, (uint32_t)i) & POS_MASK;
      const uint32_t c1 = cellidx_q(p, e1);
      W::ulds_store_u8(lds, c1, W::ulds_u8(lds, c1) & ~(uint32_t)(SF_CELL_POUT | SF_CELL_TEMP));
      W::ulds_store_u8(lds, ci, fl & ~(uint32_t)(SF_CELL_PIN_UP | SF_CELL_TEMP));
      W::ustore_i32(dmg + ci, 0);
      p_write(S, (uint32_t)i, 0u);
      S.dirty = 1u;
    } else if (sit == SH_WALL && d >= LIM_BLOCK) {
      W::ulds_store_u8(lds, ci, fl & ~(uint32_t)(SF_CELL_WALL | SF_CELL_TEMP));
      W::ustore_i32(dmg + ci, 0);
      S.dirty = 1u;
    }
  }

  static SF_DEV void update_tmp(Arena &S, uint8_t *lds, const Params &p, int a) {
    SF_PROF(PH_TMP);
    uint64_t cand[NB];
    V cell[NB];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) any = any || W::ballot((S.ba[j] & BA_ALIVE) != 0u) != 0ull;
    if (!any) return;  // no bullet in flight: nothing can be absorbed
    any = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const P alive = (S.ba[j] & BA_ALIVE) != 0u;
      cell[j] = S.ba[j] & POS_MASK;
      const V ci = ((cell[j] >> 20) * (uint32_t)p.N + ((cell[j] >> 10) & 1023u)) * (uint32_t)p.M + (cell[j] & 1023u);
      const V fl = W::lds_u8(lds, ci, alive);
      cand[j] = W::ballot(alive & ((fl & SF_CELL_TEMP) != 0u) & ((fl & (SF_CELL_WALL | SF_CELL_PIN_UP)) != 0u));
      any = any || cand[j] != 0ull;
    }
    if (!any) return;
    int32_t *dmg = p.aux_dmg + (size_t)a * (size_t)p.cells;
    // pass 1, slot order: absorb.  `s[2] = 0` un-designates the cell's bullet; every bullet standing on
    // this cell is absorbed by this same loop, so clearing each absorbed bullet's own flags is equivalent.
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint64_t m = cand[j];
      while (m) {
        const uint32_t l = (uint32_t)W::ctz64(m);
        m &= m - 1ull;
        const uint32_t q = W::readlane(cell[j], l);
        uint32_t fl;
        const int sit = showit_q(S, lds, p, q, fl);
        if (sit == SH_PUP || sit == SH_WALL) {
          const uint32_t ci = cellidx_q(p, q);
          W::ustore_i32(dmg + ci, W::uload_i32(dmg + ci) + (int32_t)W::readlane(S.bd[j], l));
          W::setlane(S.ba[j], l, 0u);
        } else {
          cand[j] &= ~(1ull << l);  // a human / zombie stands on the '^': the bullet hits them instead
        }
      }
    }
    // pass 2: `for(auto e: temp)` restricted to the cells that took damage (idempotent per cell)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint64_t m = cand[j];
      while (m) {
        const uint32_t l = (uint32_t)W::ctz64(m);
        m &= m - 1ull;
        break_if_spent(S, lds, p, a, W::readlane(cell[j], l));
      }
    }
  }

  // ------------------------------------------------------------------------------------------------
  // Cell bitmaps.  "Is somebody on cell q" for ONE cell is a lane compare + ballot; for many cells at once (every
  // bullet against every character, every zombie's neighbourhood against humans and bullets) that would be a loop of
  // ballots.  With the flag plane in LDS there is room for scratch bitmaps of one bit per cell next to it: one kind of
  // entity scatters its cells into a bitmap (LDS atomic OR, one instruction for all lanes), the other kind tests its
  // own cells (one gather), the bits are cleared again.  The bitmaps are all-zero outside such a build / test / clear
  // bracket.  Maps whose bitmaps would not fit (256 x 256) keep the ballot loops (BITMAPS false).
  enum { BM_HUM = 0, BM_ZOM = 1, BM_REF = 2 };
  static SF_DEV V cell_index_v(const Params &p, const V &q) {  // packed position (flag bits above it ignored) -> cell
    return W::mad24(W::mad24((q >> 20) & 3u, (uint32_t)p.N, (q >> 10) & 1023u), (uint32_t)p.M, q & 1023u);
  }
  static SF_DEV void bm_set(Arena &S, const Params &p, int which, const V &ci, P pred) {
    W::lds_or_u32(S.bm + which * p.bm_words, ci >> 5, W::shlv(V(1u), ci & 31u), pred);
  }
  static SF_DEV void bm_clear(Arena &S, const Params &p, int which, const V &ci, P pred) {
    W::lds_store_u32(S.bm + which * p.bm_words, ci >> 5, V(0u), pred);
  }
  static SF_DEV P bm_test(const Arena &S, const Params &p, int which, const V &ci, P pred) {
    return (W::shrv(W::lds_u32(S.bm + which * p.bm_words, ci >> 5, pred), ci & 31u) & 1u) != 0u;
  }
  // set bit, telling whether it was set already (by an earlier build or by another lane of this very call)
  static SF_DEV P bm_claim(Arena &S, const Params &p, int which, const V &ci, P pred) {
    const V bit = W::shlv(V(1u), ci & 31u);
    return pred & ((W::lds_or_rtn_u32(S.bm + which * p.bm_words, ci >> 5, bit, pred) & bit) != 0u);
  }
  static SF_DEV void bm_bullets(Arena &S, const Params &p, int which, bool set) {  // the designated bullets' cells
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const P ref = (S.ba[j] & BA_REF) != 0u;
      const V ci = cell_index_v(p, S.ba[j]);
      if (set)
        bm_set(S, p, which, ci, ref);
      else
        bm_clear(S, p, which, ci, ref);
    }
  }

  static SF_DEV void bm_zombies_zl(Arena &S, const Params &p, int which, bool set) {  // ZL: every live zombie's cell
    for (uint32_t j = 0; j < S.zwn; ++j) {
      const V zpw = zl_get(S, ZW_POS, j);
      const P zlive = (zpw & ZF_ALIVE) != 0u;
      const V zci = cell_index_v(p, zpw);
      if (set)
        bm_set(S, p, which, zci, zlive);
      else
        bm_clear(S, p, which, zci, zlive);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // hit_human + hit_zombie G:574-652.  A cell holds at most one character and a hit consumes only the
  // cell's designated bullet, so the two slot-ordered sweeps reduce to: (1) humans already at Hp <= 0 die
  // (lane-parallel); (2) one wave-uniform pass over designated bullets that share a cell with a live
  // character.  All cross-entity effects are additive (owner damage/effect/kills, loot, kill counters).
  // one hit: the designated bullet in slot `slot` on human hv (>= 0) or zombie zv
  static SF_DEV void hit_one(Arena &S, const Params &p, uint32_t my_team, int slot, int hv, int zv) {
    const uint32_t l = (uint32_t)slot & 63u;
    uint32_t dmgu = 0u, bb = 0u;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j == (slot >> 6)) {
        dmgu = W::readlane(S.bd[j], l), bb = W::readlane(S.bb[j], l);
        W::setlane(S.ba[j], l, 0u);  // pix->s[2] = 0; mb[...] = false
      }
    const int32_t dmg = (int32_t)dmgu;
    const int32_t eff = (int32_t)(int16_t)(bb & 0xffffu);
    const int owner = (int)(bb >> 16);  // human slot + 1, 0 = none
    const uint32_t ofl = owner ? W::readlane(S.hfl, (uint32_t)(owner - 1)) : 0u;
    const uint32_t owner_team = (uint32_t)h_team(ofl);
    if (hv >= 0) {  // human_damage G:611-634
      const uint32_t vfl = W::readlane(S.hfl, (uint32_t)hv);
      const uint32_t vteam = (uint32_t)h_team(vfl);
      const int32_t hp = (int32_t)W::readlane(S.hhp, (uint32_t)hv) - dmg;  // Character::hit CH:242-246
      W::setlane(S.hhp, (uint32_t)hv, (uint32_t)hp);
      add_lane(S.hmd, (uint32_t)hv, eff);
      const bool cross = owner && vteam != owner_team;
      if (cross) {
        add_lane(S.hdm, (uint32_t)(owner - 1), dmg);
        add_lane(S.hef, (uint32_t)(owner - 1), eff);
      }
      if (hp <= 0) {
        W::setlane(S.hfl, (uint32_t)hv, hv == p.ind ? (vfl & ~HF_ALIVE) : (vfl & ~(HF_ALIVE | HF_CTRL | HF_OCC)));
        if (owner && owner_team == my_team && vteam != my_team) {
          ++S.tkills, S.loot += 100;
          if (owner == p.ind + 1) S.loot += 900, ++S.kills;
        }
        if (cross) add_lane(S.hk, (uint32_t)(owner - 1), 1);
      }
    } else {  // zombie_damage G:574-598
      const uint32_t zp = z_read(S, ZW_POS, (uint32_t)zv);
      const int32_t hp = (int32_t)z_read(S, ZW_HP, (uint32_t)zv) - dmg;
      z_write(S, ZW_HP, (uint32_t)zv, (uint32_t)hp);
      z_write(S, ZW_MINDAMAGE, (uint32_t)zv, z_read(S, ZW_MINDAMAGE, (uint32_t)zv) + (uint32_t)eff);
      if (owner) {
        add_lane(S.hdm, (uint32_t)(owner - 1), dmg);
        add_lane(S.hef, (uint32_t)(owner - 1), eff);
      }
      if (hp <= 0) {
        z_write(S, ZW_POS, (uint32_t)zv, 0u);
        if (owner && owner_team == my_team) {
          const int pts = 500 + ((zp & ZF_SUPER) ? 250 : 0);
          ++S.tkills, S.loot += pts / 10;
          if (owner == p.ind + 1) S.loot += pts * 9 / 10, ++S.kills;
        }
        if (owner) add_lane(S.hk, (uint32_t)(owner - 1), 1);
      }
    }
  }
  static SF_DEV void hits(Arena &S, const Params &p) {
    SF_PROF(PH_HITS);
    {
      const P dying = ((S.hfl & HF_ALIVE) != 0u) & W::le0(S.hhp);                  // G:641-645
      // s[0] = (human == &hum[ind]); deleteAgent() only for i != ind  G:643,648-649
      S.hfl = W::select(dying, W::select(W::lane() == (uint32_t)p.ind, S.hfl & ~HF_ALIVE, S.hfl & ~(HF_ALIVE | HF_CTRL | HF_OCC)), S.hfl);
    }
    const uint32_t my_team = (uint32_t)h_team(W::readlane(S.hfl, (uint32_t)p.ind));
    if (BITMAPS) {
      // who stands on a designated bullet: the bullets scatter their cells, the characters test their own
      bool any = false;
#pragma unroll
      for (int j = 0; j < NB; ++j) any = any || W::ballot((S.ba[j] & BA_REF) != 0u) != 0ull;
      if (!any) return;
      if constexpr (ZL) {
        // humans first, then the zombies in slot order, word by word.  The bitmap stays as built meanwhile: a hit takes
        // its cell's bullet, and no second character stands on that cell to test the stale bit
        bm_bullets(S, p, BM_REF, true);
        const P hlive = (S.hfl & HF_ALIVE) != 0u;
        uint64_t hm = W::ballot(bm_test(S, p, BM_REF, cell_index_v(p, S.hpos), hlive) & hlive);
        while (hm) {
          const uint32_t i = (uint32_t)W::ctz64(hm);
          hm &= hm - 1ull;
          hit_one(S, p, my_team, refbullet_at(S, W::readlane(S.hpos, i)), (int)i, -1);
        }
        for (uint32_t j = 0; j < S.zwn; ++j) {
          const V zpw = zl_get(S, ZW_POS, j);
          const P zlive = (zpw & ZF_ALIVE) != 0u;
          uint64_t zmk = W::ballot(bm_test(S, p, BM_REF, cell_index_v(p, zpw), zlive) & zlive);
          while (zmk) {
            const uint32_t i = (uint32_t)W::ctz64(zmk);
            zmk &= zmk - 1ull;
            const int slot = refbullet_at(S, W::readlane(zpw, i) & POS_MASK);
            if (slot >= 0) hit_one(S, p, my_team, slot, -1, (int)(64u * j + i));
          }
        }
        W::lds_zero(S.bm + BM_REF * p.bm_words, (uint32_t)p.bm_words);
        return;
      }
      {
      bm_bullets(S, p, BM_REF, true);
      const P hlive = (S.hfl & HF_ALIVE) != 0u;
      const P zlive = (S.zpos & ZF_ALIVE) != 0u;
      uint64_t hm = W::ballot(bm_test(S, p, BM_REF, cell_index_v(p, S.hpos), hlive) & hlive);
      uint64_t zmk = W::ballot(bm_test(S, p, BM_REF, cell_index_v(p, S.zpos), zlive) & zlive);
      bm_bullets(S, p, BM_REF, false);
      while (hm) {
        const uint32_t i = (uint32_t)W::ctz64(hm);
        hm &= hm - 1ull;
        hit_one(S, p, my_team, refbullet_at(S, W::readlane(S.hpos, i)), (int)i, -1);
      }
      while (zmk) {
        const uint32_t i = (uint32_t)W::ctz64(zmk);
        zmk &= zmk - 1ull;
        hit_one(S, p, my_team, refbullet_at(S, W::readlane(S.zpos, i) & POS_MASK), -1, (int)i);
      }
      return;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint64_t m = W::ballot((S.ba[j] & BA_REF) != 0u);
      while (m) {
        const uint32_t l = (uint32_t)W::ctz64(m);
        m &= m - 1ull;
        const uint32_t q = W::readlane(S.ba[j], l) & POS_MASK;
        const int hv = live_human_at(S, q);
        const int zv = hv >= 0 ? -1 : zombie_at(S, q);
        if (hv < 0 && zv < 0) continue;
        hit_one(S, p, my_team, j * 64 + (int)l, hv, zv);
      }
    }
  }

  // ------------------------------------------------------------------------------------------------
  // update_bull G:1059-1100.  Expiry / advance is lane-parallel; "the last bullet to enter a cell becomes
  // the cell's designated bullet" is resolved per distinct destination with ballots.
  static SF_DEV void update_bull(Arena &S, uint8_t *lds, const Params &p) {
    SF_PROF(PH_BULL);
    bool any = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) any = any || W::ballot((S.ba[j] & BA_ALIVE) != 0u) != 0ull;
    const uint32_t r = draw(S, lds, p) & 1u;  // drawn even when no bullet is alive
    if (!any) return;
    uint64_t moved[NB];
    V nci[NB];
    P passed[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const P alive = (S.ba[j] & BA_ALIVE) != 0u;
      const V range = S.bc[j] & 0xffffu, trav = S.bc[j] >> 16;
      const P live = alive & W::ltu(trav + 1u, range);  // Bullet::expire IT:165-168: dist + 1 >= range
      const V q = S.ba[j] & POS_MASK;
      const V d = (S.ba[j] >> BA_WAY_SH) & 3u;
      const V rr = ((q >> 10) & 1023u) + W::select(d == 0u, V(1u), W::select(d == 2u, V(0xffffffffu), V(0u)));
      const V cc = (q & 1023u) + W::select(d == 1u, V(1u), W::select(d == 3u, V(0xffffffffu), V(0u)));
      const P inb = live & W::ltu(rr, (uint32_t)p.N) & W::ltu(cc, (uint32_t)p.M);
      const V ci = ((q >> 20) * (uint32_t)p.N + rr) * (uint32_t)p.M + cc;
      const V fl = W::lds_u8(lds, ci, inb);
      const P temp = (fl & SF_CELL_TEMP) != 0u, wall = (fl & SF_CELL_WALL) != 0u,
              pin = (fl & (SF_CELL_PIN_UP | SF_CELL_PIN_DN)) != 0u;
      const V nq = (q & (3u << 20)) | (rr << 10) | cc;
      // a '^' / 'v' cell shows the character standing on it instead (showit order), which lets bullets in
      uint64_t amb = W::ballot(inb & (!temp) & (!wall) & pin);
      uint64_t covered = 0ull;
      while (amb) {
        const uint32_t l = (uint32_t)W::ctz64(amb);
        amb &= amb - 1ull;
        const uint32_t qq = W::readlane(nq, l);
        if (human_at(S, qq) >= 0 || zombie_at(S, qq) >= 0) covered |= 1ull << l;
      }
      const P pass = inb & (temp | ((!wall) & ((!pin) | W::frombits(covered))));
      moved[j] = W::ballot(pass);
      nci[j] = ci, passed[j] = pass;
      // survivors advance and lose their designation; everything else that was alive dies
      const V na = (S.ba[j] & ~(POS_MASK | BA_REF)) | nq;
      S.ba[j] = W::select(pass, na, V(0u));
      S.bc[j] = W::select(pass, S.bc[j] + 0x10000u, S.bc[j]);
    }
    if (BITMAPS) {
      // every surviving bullet has just entered its cell.  If no two of them entered the same cell (they claim their
      // cells in a scratch bitmap), each one is its cell's last entrant and is designated, whatever the sweep order
      uint64_t dup = 0ull;
#pragma unroll
      for (int j = 0; j < NB; ++j) dup |= W::ballot(bm_claim(S, p, BM_HUM, nci[j], passed[j]));
#pragma unroll
      for (int j = 0; j < NB; ++j) bm_clear(S, p, BM_HUM, nci[j], passed[j]);
      if (!dup) {
#pragma unroll
        for (int j = 0; j < NB; ++j) S.ba[j] = W::select(passed[j], S.ba[j] | BA_REF, S.ba[j]);
        return;
      }
    }
    // designation: r = 1 sweeps slots ascending (last entrant = highest slot), r = 0 descending
#pragma unroll
    for (int j0 = 0; j0 < NB; ++j0) {
      while (moved[j0]) {
        const uint32_t l = (uint32_t)W::ctz64(moved[j0]);
        uint32_t q = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (j == j0) q = W::readlane(S.ba[j], l) & POS_MASK;
        uint64_t same[NB];
        int win = -1;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          same[j] = W::ballot(((S.ba[j] & BA_ALIVE) != 0u) & ((S.ba[j] & POS_MASK) == q)) & moved[j];
          moved[j] &= ~same[j];
        }
        if (r) {
#pragma unroll
          for (int j = 0; j < NB; ++j)
            if (same[j]) win = j * 64 + 63 - W::clz64(same[j]);
        } else {
#pragma unroll
          for (int j = NB - 1; j >= 0; --j)
            if (same[j]) win = j * 64 + W::ctz64(same[j]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (j == (win >> 6)) W::setlane(S.ba[j], (uint32_t)win & 63u, W::readlane(S.ba[j], (uint32_t)win & 63u) | BA_REF);
      }
    }
  }

  // ------------------------------------------------------------------------------------------------
  // claim_chest G:507-515, teleport G:517-530
  static SF_DEV void claim_chest(Arena &S, uint8_t *lds, const Params &p, uint32_t i) {
    const uint32_t q = W::readlane(S.hpos, i);
    const uint32_t ci = cellidx_q(p, q);
    const uint32_t fl = W::ulds_u8(lds, ci);
    if (fl & SF_CELL_CHEST) {
      const int32_t *c = p.tab->cons_items[(fl >> SF_CELL_CONS_SHIFT) & 3u];  // Human::claim_chest CH:372-377
      add_lane(S.hst, i, c[0]);
      add_lane(S.hhp, i, c[1]);
      add_lane(S.hmd, i, c[2]);
      W::ulds_store_u8(lds, ci, fl & ~(uint32_t)(SF_CELL_CHEST | (3u << SF_CELL_CONS_SHIFT)));
      S.dirty = 1u;
      --S.chests;
    }
  }
  static SF_DEV void teleport(Arena &S, uint8_t *lds, const Params &p, int a, uint32_t i) {
    const uint32_t q = W::readlane(S.hpos, i);
    const uint32_t ci = cellidx_q(p, q);
    const uint32_t fl = W::ulds_u8(lds, ci);
    if (!(fl & (SF_CELL_PIN_UP | SF_CELL_PIN_DN))) return;  // portal_ind == -1
    const int index = (fl & SF_CELL_TEMP) ? (int)W::uload_i16(p.aux_pidx + (size_t)a * (size_t)p.cells + ci)
                                          : (int)W::uload_i16(p.map_pidx + ci);
    if (index < 0 || index >= p.P) return;
    const uint32_t e = p_read(S, (uint32_t)index) & POS_MASK;
    uint32_t efl;
    if (showit_q(S, lds, p, e, efl) != SH_POUT) return;
    W::setlane(S.hpos, i, e);
  }

  // ------------------------------------------------------------------------------------------------
  // obey G:695-821 for human slot i (wave-uniform)
  static SF_DEV void obey(Arena &S, uint8_t *lds, const Params &p, int a, uint32_t c, uint32_t i) {
    if (c == '+') return;
    const uint32_t fl0 = W::readlane(S.hfl, i);
    const uint32_t q0 = W::readlane(S.hpos, i);
    const int f = pos_f(q0), r = pos_r(q0), cc0 = pos_c(q0);
    const int way = h_way(fl0);
    if (c == '_') {
      W::setlane(S.hhp, i, 0u);
      return;
    }
    if (c == '[' || c == ']') {
      const int rr = r + DX(way - 1), cc = cc0 + DY(way - 1);
      if (!inmap(p, rr, cc)) return;
      uint32_t fl;
      if (showit(S, lds, p, f, rr, cc, fl) != SH_EMPTY) return;
      const uint32_t ci = cellidx(p, f, rr, cc);
      const uint32_t bp = W::readlane(S.hbpk, i);
      const uint32_t blocks = bp & 255u, portals = (bp >> 8) & 255u;
      const int pind = (int)((bp >> 16) & 255u) - 1;
      int32_t *dmg = p.aux_dmg + (size_t)a * (size_t)p.cells;
      if (c == '[') {
        if (blocks) {
          W::ulds_store_u8(lds, ci, fl | SF_CELL_TEMP | SF_CELL_WALL);
          W::ustore_i32(dmg + ci, 0);
          W::setlane(S.hbpk, i, bp - 1u);
          S.dirty = 1u;
        }
        return;
      }
      if (pind != -1) {
        W::ulds_store_u8(lds, ci, fl | SF_CELL_TEMP | SF_CELL_PIN_UP);
        W::ustore_i32(dmg + ci, 0);
        W::ustore_i16(p.aux_pidx + (size_t)a * (size_t)p.cells + ci, (int16_t)pind);
        W::setlane(S.hbpk, i, bp & 0xffffu);
        S.dirty = 1u;
      } else if (portals) {
        const int index = p_ind(S, p);
        if (index == -1) return;
        W::ulds_store_u8(lds, ci, fl | SF_CELL_TEMP | SF_CELL_POUT);
        W::ustore_i32(dmg + ci, 0);
        W::setlane(S.hbpk, i, (bp - 256u) | ((uint32_t)(index + 1) << 16));
        p_write(S, (uint32_t)index, pos_pack(f, rr, cc) | PF_ACTIVE);
        S.dirty = 1u;
      }
      return;
    }
    if (c == 'q' || c == 'e') {  // turn_l / turn_r CH:745-759
      const int nw = c == 'e' ? (way == 1 ? 4 : way - 1) : (way == 4 ? 1 : way + 1);
      W::setlane(S.hfl, i, (fl0 & ~HF_WAY_MASK) | (uint32_t)nw);
      return;
    }
    if (c == 's' || c == 'd' || c == 'w' || c == 'a') {
      const int d = c == 's' ? 0 : (c == 'd' ? 1 : (c == 'w' ? 2 : 3));
      const int rr = r + DX(d), cc = cc0 + DY(d);
      if (!inmap(p, rr, cc)) return;
      uint32_t fl;
      const int sit = showit(S, lds, p, f, rr, cc, fl);
      if (sit == SH_CHEST || sit == SH_PUP || sit == SH_PDN || sit == SH_EMPTY || sit == SH_BULLET)
        W::setlane(S.hpos, i, pos_pack(f, rr, cc));
      return;
    }
    int k = -1;
    if ((k = (c == 'f' ? 0 : c == 'g' ? 1 : c == 'h' ? 2 : c == 'j' ? 3 : -1)) >= 0) {
      if (get16(S.hc01, S.hc23, i, k)) W::setlane(S.hfl, i, set_vec_sel(fl0, 0, k));
      return;
    }
    if ((k = (c == 'k' ? 0 : c == 'l' ? 1 : c == ';' ? 2 : c == '\'' ? 3 : -1)) >= 0) {
      if (get16(S.ht01, S.ht23, i, k)) W::setlane(S.hfl, i, set_vec_sel(fl0, 1, k));
      return;
    }
    if ((k = (c == 'c' ? 0 : c == 'v' ? 1 : c == 'b' ? 2 : c == 'n' ? 3 : c == 'm' ? 4 : c == ',' ? 5 : c == '.' ? 6
                                                                                              : c == '/' ? 7 : -1)) >= 0) {
      if (p.tab->der[h_block(p, fl0)].weapon_lvl[k]) W::setlane(S.hfl, i, set_vec_sel(fl0, 2, k));
      return;
    }
    const int vec = h_vec(fl0), sel = h_sel(fl0);
    if (c == 'u') {  // Human::use CH:379-389
      if (vec != 0) return;
      const uint32_t n = get16(S.hc01, S.hc23, i, sel);
      if (n < 1u) return;
      const int32_t *ci = p.tab->cons_items[sel];
      add_lane(S.hst, i, ci[0]);
      add_lane(S.hhp, i, ci[1]);
      add_lane(S.hmd, i, ci[2]);
      set16(S.hc01, S.hc23, i, sel, n - 1u);
      if (n - 1u < 1u) W::setlane(S.hfl, i, set_vec_sel(fl0, -1, sel));
      return;
    }
    if (c == 'z' || c == 'x') {
      const int rr = r + DX(way - 1), cc = cc0 + DY(way - 1);
      const int index = b_ind(S, p);
      if (index == -1 || !inmap(p, rr, cc)) return;
      const Derived &d = p.tab->der[h_block(p, fl0)];
      const int32_t md = (int32_t)W::readlane(S.hmd, i);
      const int32_t st = (int32_t)W::readlane(S.hst, i);
      bool can;
      int dmg = 0, eff = 0, range = 1;
      if (c == 'z') {  // Human::punch CH:391-397
        dmg = d.cd_punch > md ? d.cd_punch : md;
        can = true;
      } else if (vec == 1) {  // Human::throw_it CH:410-427
        const int32_t *t = d.thr[sel];
        const uint32_t n = get16(S.ht01, S.ht23, i, sel);
        dmg = t[1] > t[1] + md ? t[1] : t[1] + md;
        eff = t[2], range = t[3];
        if (st + t[0] < 0)
          can = false;
        else if (n < 1u) {
          W::setlane(S.hfl, i, set_vec_sel(fl0, -1, sel));
          can = false;
        } else {
          W::setlane(S.hst, i, (uint32_t)(st + t[0]));
          set16(S.ht01, S.ht23, i, sel, n - 1u);
          if (n - 1u < 1u) W::setlane(S.hfl, i, set_vec_sel(fl0, -1, sel));
          can = true;
        }
      } else if (vec == 2) {  // Human::shot_it CH:399-408
        const int32_t *w = d.weapon[sel];
        if (st + w[0] < 0)
          can = false;
        else {
          W::setlane(S.hst, i, (uint32_t)(st + w[0]));
          dmg = d.cd_weapon[sel] > w[1] + md ? d.cd_weapon[sel] : w[1] + md;
          eff = w[2], range = w[3];
          can = true;
        }
      } else
        return;
      uint32_t fl;
      const int sit = showit(S, lds, p, f, rr, cc, fl);
      if (can && bullet_may_enter(sit, fl)) bullet_put(S, index, pos_pack(f, rr, cc), way, dmg, eff, range, (int)i + 1);
      return;
    }
  }

  // human_rnpc_bot G:1927-1940
  // x % 5 and x % 7 for a draw (x < 1024): multiply-shift quotients that are exact on that range (checked for every x),
  // four scalar instructions where the compiler's general 32-bit form takes seven
  static SF_DEV uint32_t mod5(uint32_t x) { return x - ((x * 205u) >> 10) * 5u; }
  static SF_DEV uint32_t mod7(uint32_t x) { return x - ((x * 1171u) >> 13) * 7u; }
  // the command tables of G:1929,1936,1939 as packed constants (char k in bits 8k..8k+7): a shift and a mask in scalar
  // registers instead of a byte load from constant memory on every NPC's path
  static constexpr uint64_t pack8(const char *t, int n) {
    uint64_t v = 0;
    for (int k = 0; k < n; ++k) v |= (uint64_t)(uint8_t)t[k] << (8 * k);
    return v;
  }
  static SF_DEV uint32_t human_rnpc_bot(Arena &S, const uint8_t *lds, const Params &p, bool pick_weapon) {
    // every path starts with one draw; the non-'x' path always draws twice more (G:1928-1939)
    constexpr uint64_t T_WEAPON = pack8("cvbnm,./", 8), T_MOVE = pack8("12awsdp", 7), T_MISC = pack8("+ufghj[]", 8);
    const uint32_t d1 = draw(S, lds, p);
    if (pick_weapon) return (uint32_t)(T_WEAPON >> ((d1 & 7u) * 8u)) & 255u;  // frame % 50 <= 1
    if (mod5(d1) < 3u) return 'x';
    const uint32_t d2 = draw(S, lds, p);
    const uint32_t d3 = draw(S, lds, p);
    return mod5(d2) < 3u ? (uint32_t)(T_MOVE >> (mod7(d3) * 8u)) & 255u : (uint32_t)(T_MISC >> ((d3 & 7u) * 8u)) & 255u;
  }

  // human_action G:965-1012.  S.hcmd holds this step's external commands on lanes < n_agents.
  //
  // The reference sweeps the live humans in slot order (ascending or descending by one draw): obey; teleport;
  // claim_chest for each.  What one human does can matter to a later one only through a cell both touch (a move's
  // target or origin, a placed block or portal, a fresh bullet) or through the bullet / portal slot pools.  So the
  // sweep is first evaluated for all humans at once, one lane each, against the state at the start of the sweep, and
  // the cells every human reads or writes are compared: if no two humans meet on a cell, nobody stands on or steps
  // onto a portal entrance, and the bullet pool cannot run dry, the lane-parallel result IS the sweep's (moves,
  // selections, consumables, stamina and ammunition are applied per lane; the bullets are allocated by a short
  // slot-ordered loop, because `b_ind` hands out slots in sweep order; the rare block / portal placements run through
  // obey() itself, in sweep order).  Otherwise the sweep runs one human at a time (human_action_serial: obey /
  // teleport / claim_chest as the reference has them), from the same untouched state.
  static SF_DEV void human_action_serial(Arena &S, uint8_t *lds, const Params &p, int a, uint64_t alive, uint32_t r) {
    uint64_t m = alive;
    while (m) {
      const uint32_t i = r ? (uint32_t)W::ctz64(m) : (uint32_t)(63 - W::clz64(m));
      m &= ~(1ull << i);
      // a human killed by obey() of an earlier one is impossible (hits land in hit_human), so `alive` is stable
      obey(S, lds, p, a, W::readlane(S.hcmd, i), i);
      SF_STAMP(S, 10);
      teleport(S, lds, p, a, i);
      SF_STAMP(S, 11);
      claim_chest(S, lds, p, i);
      SF_STAMP(S, 12);
    }
  }
  static SF_DEV V get16v(const V &lo, const V &hi, const V &k) {  // k in 0..3, per lane
    return W::shrv(W::select(W::ltu(k, 2u), lo, hi), (k & 1u) << 4) & 0xffffu;
  }
  static SF_DEV void set16v(V &lo, V &hi, const V &k, const V &val, P pred) {
    const V sh = (k & 1u) << 4;
    const V keep = ~W::shlv(V(0xffffu), sh);
    const V ins = W::shlv(val & 0xffffu, sh);
    const P low = W::ltu(k, 2u);
    lo = W::select(pred & low, (lo & keep) | ins, lo);
    hi = W::select(pred & (!low), (hi & keep) | ins, hi);
  }
  static SF_DEV V set_vec_sel_v(const V &fl, uint32_t vec_plus1, const V &sel) {  // sel per lane, 0-based
    return (fl & ~((3u << HF_VEC_SH) | (15u << HF_IND_SH))) | (vec_plus1 << HF_VEC_SH) | ((sel + 1u) << HF_IND_SH);
  }

  static SF_DEV void human_action(Arena &S, uint8_t *lds, const Params &p, int a) {
    SF_PROF(PH_HUMAN);
    const uint64_t alive = W::ballot((S.hfl & HF_ALIVE) != 0u) & capmask(p.H);
    // get_command G:929-937, slot order, for i != ind: remote keep theirs, rnpc draw, agents keep theirs, others '+'
    {
      const P ext = ((S.hfl & (HF_REMOTE | HF_CTRL)) != 0u) | (W::lane() == (uint32_t)p.ind);
      S.hcmd = W::select(ext, S.hcmd, V((uint32_t)'+'));
      uint64_t m = alive & ~(1ull << p.ind) & W::ballot((S.hfl & (HF_RNPC | HF_REMOTE)) == HF_RNPC);
      const bool pick_weapon = m && S.frame % 50 <= 1;  // the same for every NPC of this sweep
      while (m) {
        const uint32_t i = (uint32_t)W::ctz64(m);
        m &= m - 1ull;
        W::setlane(S.hcmd, i, human_rnpc_bot(S, lds, p, pick_weapon));
      }
    }
    SF_STAMP(S, 9);
    const uint32_t r = draw(S, lds, p) & 1u;
    if (!alive) {
      S.hcmd = V((uint32_t)'+');
      return;
    }
    if (!(alive & (alive - 1ull))) {  // one human: nobody to meet, and one obey() costs less than the all-lanes form
      human_action_serial(S, lds, p, a, alive, r);
      S.hcmd = V((uint32_t)'+');
      return;
    }
    // ---- one lane per human: what would it do, and to which cell --------------------------------------------
    const P live = W::frombits(alive);
    const V cw = W::lds_u32(S.ht, (S.hcmd >> 2) & 31u, live & W::ltu(S.hcmd, 128u));
    const V c8 = W::shrv(cw, (S.hcmd & 3u) << 3) & 255u;
    const V cls = c8 & 15u, prm = c8 >> 4;
    const V q0 = S.hpos;
    const V hr = (q0 >> 10) & 1023u, hc = q0 & 1023u, hf = (q0 >> 20) & 3u;
    const V way = S.hfl & HF_WAY_MASK;
    const P is_move = live & (cls == (uint32_t)CL_MOVE);
    const P is_place = live & ((cls == (uint32_t)CL_BLOCK) | (cls == (uint32_t)CL_PORTAL));
    const P is_shoot = live & ((cls == (uint32_t)CL_PUNCH) | (cls == (uint32_t)CL_FIRE));
    const V dir = W::select(is_move, prm, way - 1u);
    const V rr = hr + W::select(dir == 0u, V(1u), W::select(dir == 2u, V(0xffffffffu), V(0u)));
    const V cc = hc + W::select(dir == 1u, V(1u), W::select(dir == 3u, V(0xffffffffu), V(0u)));
    const P inb = (is_move | is_place | is_shoot) & W::ltu(rr, (uint32_t)p.N) & W::ltu(cc, (uint32_t)p.M);
    const V tq = (q0 & (3u << 20)) | (rr << 10) | cc;
    const V oci = (hf * (uint32_t)p.N + hr) * (uint32_t)p.M + hc;
    const V tci = (hf * (uint32_t)p.N + rr) * (uint32_t)p.M + cc;
    const V tfl = W::lds_u8(lds, tci, inb);
    const V ofl = W::lds_u8(lds, oci, live);
    // who is on the target cells; do two humans meet on a cell
    const uint64_t placers = W::ballot(is_place & inb);
    const P needb = inb & (((tfl & SF_CELL_POUT) != 0u) | is_place);
    P oH, oZ, oB;
    bool slow = W::ballot(live & ((ofl & (SF_CELL_PIN_UP | SF_CELL_PIN_DN | SF_CELL_CHEST)) != 0u)) != 0ull;
    if (BITMAPS) {
      // occupants scatter their cells into bitmaps, the acting humans test their target cells.  A fourth bitmap
      // holds the cells of humans that may walk away in this sweep, then the claims of the targets themselves
      const P hocc = (S.hfl & HF_OCC) != 0u, zlive = (S.zpos & ZF_ALIVE) != 0u;
      const V hci = cell_index_v(p, S.hpos), zci = cell_index_v(p, S.zpos);
      const bool bul = W::ballot(needb) != 0ull;
      bm_set(S, p, BM_HUM, hci, hocc);
      if constexpr (ZL) bm_zombies_zl(S, p, BM_ZOM, true); else bm_set(S, p, BM_ZOM, zci, zlive);
      if (bul) bm_bullets(S, p, BM_REF, true);
      oH = bm_test(S, p, BM_HUM, tci, inb), oZ = bm_test(S, p, BM_ZOM, tci, inb);
      oB = needb & bm_test(S, p, BM_REF, tci, needb);
      bm_clear(S, p, BM_HUM, hci, hocc);
      if constexpr (ZL) bm_zombies_zl(S, p, BM_ZOM, false); else bm_clear(S, p, BM_ZOM, zci, zlive);
      if (bul) bm_bullets(S, p, BM_REF, false);
      // (the bitmaps are free again) the cells of humans that may walk away, then the claims of the targets
      bm_set(S, p, BM_HUM, oci, is_move);
      const P shared = bm_claim(S, p, BM_ZOM, tci, inb);
      const P leaves = inb & bm_test(S, p, BM_HUM, tci, inb);
      bm_clear(S, p, BM_HUM, oci, is_move);
      bm_clear(S, p, BM_ZOM, tci, inb);
      if (W::ballot(leaves | shared)) slow = true;
    } else {
      const uint64_t movers = W::ballot(is_move);
      const uint64_t needbm = W::ballot(needb);
      uint64_t occH = 0ull, occZ = 0ull, occB = 0ull;
      uint64_t m = W::ballot(inb);
      while (m) {
        const uint32_t i = (uint32_t)W::ctz64(m);
        const uint64_t bit = m & (0ull - m);
        m ^= bit;
        const uint32_t q = W::readlane(tq, i);
        const uint64_t hs = W::ballot(((S.hfl & HF_OCC) != 0u) & (S.hpos == q));
        const uint64_t same = W::ballot(inb & (tq == q));
        if (hs) occH |= bit;
        if constexpr (ZL) {
          if (zombie_at(S, q) >= 0) occZ |= bit;
        } else {
          if (W::ballot((S.zpos & (ZF_ALIVE | POS_MASK)) == (ZF_ALIVE | q))) occZ |= bit;
        }
        if ((needbm & bit) && refbullet_at(S, q) >= 0) occB |= bit;
        // the human standing there may move away in this sweep; another human aims at the same cell
        if ((hs & movers) || (same & (same - 1ull))) slow = true;
      }
      oH = W::frombits(occH), oZ = W::frombits(occZ), oB = W::frombits(occB);
    }
    const P t_wall = (tfl & SF_CELL_WALL) != 0u, t_pin = (tfl & (SF_CELL_PIN_UP | SF_CELL_PIN_DN)) != 0u;
    // showit() of the target is one of '?', '^', 'v', '.', '*' (G:750-756): not a wall, nobody on it, and not a bare 'O'
    const P bare_out = ((tfl & SF_CELL_POUT) != 0u) & (!t_pin) & ((tfl & SF_CELL_CHEST) == 0u) & (!oB);
    const P pass = is_move & inb & (!t_wall) & (!oH) & (!oZ) & (!bare_out);
    if (W::ballot(pass & t_pin)) slow = true;  // steps onto a portal entrance: teleport() follows
    const uint64_t shooters = W::ballot(is_shoot & inb);
    if (shooters) {  // `index = b_ind(); if(index == -1) return;` must not trigger for anybody
      int freeb = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int left = p.B - 64 * j;
        if (left > 0) freeb += W::popc64(~W::ballot((S.ba[j] & BA_ALIVE) != 0u) & capmask(left));
      }
      if (freeb < W::popc64(shooters)) slow = true;
    }
    SF_COUNT(slow ? 1 : 0);  // (test build: how often the sweep falls back to one human at a time)
    if (slow) {
      human_action_serial(S, lds, p, a, alive, r);
      S.hcmd = V((uint32_t)'+');
      return;
    }
    // ---- the sweep, all humans at once -------------------------------------------------------------------------
    const V prof_base = W::select((S.hfl & HF_PROF) != 0u, V((uint32_t)(HT_PROF + HT_PROF_STRIDE * p.npc_block)),
                                  W::mad24((S.hfl >> HF_AGP_SH) & 15u, (uint32_t)HT_PROF_STRIDE, V((uint32_t)HT_PROF)));
    const V vec1 = (S.hfl >> HF_VEC_SH) & 3u;          // backpack.vec + 1
    const V sel = ((S.hfl >> HF_IND_SH) & 15u) - 1u;   // backpack.ind (0xffffffff: none)
    V nfl = S.hfl;
    // '_'  G:696-699
    S.hhp = W::select(live & (cls == (uint32_t)CL_SUICIDE), V(0u), S.hhp);
    // turn_l / turn_r CH:745-759
    {
      const V nw = W::select(prm != 0u, W::select(way == 1u, V(4u), way - 1u), W::select(way == 4u, V(1u), way + 1u));
      nfl = W::select(live & (cls == (uint32_t)CL_TURN), (nfl & ~HF_WAY_MASK) | nw, nfl);
    }
    // selections G:759-791: a consumable / throwable needs stock, a weapon its level
    {
      const P sc = live & (cls == (uint32_t)CL_SELC) & (get16v(S.hc01, S.hc23, prm) != 0u);
      const P st = live & (cls == (uint32_t)CL_SELT) & (get16v(S.ht01, S.ht23, prm) != 0u);
      const P is_w = live & (cls == (uint32_t)CL_SELW);
      const P sw = is_w & (W::lds_u32(S.ht, prof_base + (uint32_t)HT_P_WLVL + prm, is_w) != 0u);
      nfl = W::select(sc, set_vec_sel_v(nfl, 1u, prm), nfl);
      nfl = W::select(st, set_vec_sel_v(nfl, 2u, prm), nfl);
      nfl = W::select(sw, set_vec_sel_v(nfl, 3u, prm), nfl);
    }
    // 'u': Human::use CH:379-389
    {
      const V k = sel & 3u;
      const V n = get16v(S.hc01, S.hc23, k);
      const P use = live & (cls == (uint32_t)CL_USE) & (vec1 == 1u) & (n != 0u);
      if (W::ballot(use)) {
        const V ci = k * 3u + (uint32_t)HT_CONS;
        S.hst = S.hst + W::lds_u32(S.ht, ci, use);
        S.hhp = S.hhp + W::lds_u32(S.ht, ci + 1u, use);
        S.hmd = S.hmd + W::lds_u32(S.ht, ci + 2u, use);
        set16v(S.hc01, S.hc23, k, n - 1u, use);
        nfl = W::select(use & (n == 1u), nfl & ~(3u << HF_VEC_SH), nfl);  // vec = -1, ind kept
      }
    }
    // moves G:742-758, then claim_chest G:507-515 on the cell reached
    if (W::ballot(pass)) {
      S.hpos = W::select(pass, tq, S.hpos);
      const P got = pass & ((tfl & SF_CELL_CHEST) != 0u);
      const uint64_t gm = W::ballot(got);
      if (gm) {
        const V ci = ((tfl >> SF_CELL_CONS_SHIFT) & 3u) * 3u + (uint32_t)HT_CONS;  // Human::claim_chest CH:372-377
        S.hst = S.hst + W::lds_u32(S.ht, ci, got);
        S.hhp = S.hhp + W::lds_u32(S.ht, ci + 1u, got);
        S.hmd = S.hmd + W::lds_u32(S.ht, ci + 2u, got);
        W::lds_store_u8(lds, tci, tfl & ~(uint32_t)(SF_CELL_CHEST | (3u << SF_CELL_CONS_SHIFT)), got);
        S.dirty = 1u;
        S.chests -= W::popc64(gm);
      }
    }
    // punch / throw / shoot G:796-819: stamina and ammunition are spent whenever a slot is free and the target is on
    // the map, even if the bullet then cannot enter the cell
    V bdmg = V(0u), beff = V(0u), brange = V(1u);
    uint64_t fire = 0ull;
    if (shooters) {
      const P sh = is_shoot & inb;
      const P punch = sh & (cls == (uint32_t)CL_PUNCH);
      const P thr = sh & (cls == (uint32_t)CL_FIRE) & (vec1 == 2u);
      const P gun = sh & (cls == (uint32_t)CL_FIRE) & (vec1 == 3u);
      const V md = S.hmd;
      // Human::punch CH:391-397
      const V cdp = W::lds_u32(S.ht, prof_base + (uint32_t)HT_P_CDPUNCH, punch);
      bdmg = W::select(punch, W::select(W::gts(cdp, md), cdp, md), bdmg);
      P can = punch;
      // Human::throw_it CH:410-427
      if (W::ballot(thr)) {
        const V k = sel & 3u;
        const V tb_ = prof_base + (uint32_t)HT_P_THR + (k << 2);
        const V t0 = W::lds_u32(S.ht, tb_, thr), t1 = W::lds_u32(S.ht, tb_ + 1u, thr);
        const V t2 = W::lds_u32(S.ht, tb_ + 2u, thr), t3 = W::lds_u32(S.ht, tb_ + 3u, thr);
        const V n = get16v(S.ht01, S.ht23, k);
        const P tired = W::gts(V(0u), S.hst + t0);  // stamina + cost < 0
        const P empty = thr & (!tired) & (n == 0u);
        const P ok = thr & (!tired) & (n != 0u);
        nfl = W::select(empty | (ok & (n == 1u)), nfl & ~(3u << HF_VEC_SH), nfl);
        S.hst = W::select(ok, S.hst + t0, S.hst);
        set16v(S.ht01, S.ht23, k, n - 1u, ok);
        bdmg = W::select(thr, W::select(W::gts(t1, t1 + md), t1, t1 + md), bdmg);
        beff = W::select(thr, t2, beff), brange = W::select(thr, t3, brange);
        can = can | ok;
      }
      // Human::shot_it CH:399-408
      if (W::ballot(gun)) {
        const V k = sel & 7u;
        const V wb = prof_base + (uint32_t)HT_P_WEAPON + (k << 2);
        const V w0 = W::lds_u32(S.ht, wb, gun), w1 = W::lds_u32(S.ht, wb + 1u, gun);
        const V w2 = W::lds_u32(S.ht, wb + 2u, gun), w3 = W::lds_u32(S.ht, wb + 3u, gun);
        const V cdw = W::lds_u32(S.ht, prof_base + (uint32_t)HT_P_CDW + k, gun);
        const P ok = gun & (!W::gts(V(0u), S.hst + w0));
        S.hst = W::select(ok, S.hst + w0, S.hst);
        bdmg = W::select(gun, W::select(W::gts(cdw, w1 + md), cdw, w1 + md), bdmg);
        beff = W::select(gun, w2, beff), brange = W::select(gun, w3, brange);
        can = can | ok;
      }
      // `(sit != '#' && sit != 'v' && sit != '^') || s[10]`  G:812: a character on an entrance hides it (showit order)
      const P enter = ((tfl & SF_CELL_TEMP) != 0u) | ((!t_wall) & ((!t_pin) | oH | oZ));
      fire = W::ballot(can & enter);
    }
    S.hfl = nfl;
    // the order-dependent rest, in sweep order: slot allocation of the bullets; block / portal placement through obey()
    uint64_t m = fire | placers;
    while (m) {
      const uint32_t i = r ? (uint32_t)W::ctz64(m) : (uint32_t)(63 - W::clz64(m));
      const uint64_t bit = 1ull << i;
      m &= ~bit;
      if (placers & bit) {
        obey(S, lds, p, a, W::readlane(S.hcmd, i), i);
      } else {
        const int index = b_ind(S, p);  // never -1 here (checked above)
        bullet_put(S, index, W::readlane(tq, i), (int)W::readlane(way, i), (int)W::readlane(bdmg, i),
                   (int)(int32_t)W::readlane(beff, i), (int)W::readlane(brange, i), (int)i + 1);
      }
    }
    S.hcmd = V((uint32_t)'+');
  }

  // ------------------------------------------------------------------------------------------------
  // check_end G:1102-1229 (logic only)
  static SF_DEV bool rivals_are_dead(const Arena &S, uint32_t my_team) {  // G:497-505
    const V team = (S.hfl >> HF_TEAM_SH) & 255u;
    return W::ballot(((S.hfl & HF_ALIVE) != 0u) & (team != 0u) & (team != my_team)) == 0ull;
  }
  static SF_DEV int check_end(const Arena &S, const Params &p) {
    const uint32_t my_team = (uint32_t)h_team(W::readlane(S.hfl, (uint32_t)p.ind));
    if (p.mode == SF_MODE_BATTLE && rivals_are_dead(S, my_team)) return SF_WON;
    if ((int32_t)W::readlane(S.hhp, (uint32_t)p.ind) <= 0) return SF_DIED;
    if (p.mode == SF_MODE_TIMER) {
      if (S.frame - 1 >= p.timer_lim) return S.kills < p.level * 5 ? SF_TIME_LOST : SF_TIME_WON;
      return SF_RUNNING;
    }
    if (p.level * 5 <= S.kills && p.mode == SF_MODE_SOLO) return SF_WON;
    if (p.level * 10 <= S.tkills && rivals_are_dead(S, my_team) && p.mode == SF_MODE_SQUAD) return SF_WON;
    return SF_RUNNING;
  }

  static SF_DEV void latch_results(const Arena &S, const Params &p, int a) {
    // [kills, teams_kills, loot, damage, effect, Hp, frames, outcome] per agent
    const P ag = W::ltu(W::lane(), (uint32_t)p.n_agents);
    const V base = (V((uint32_t)a * (uint32_t)p.n_agents) + W::lane()) * 8u;
    uint32_t *res = (uint32_t *)p.results;
    W::gstore(res, base + 0u, S.hk, ag);
    W::gstore(res, base + 1u, V((uint32_t)S.tkills), ag);
    W::gstore(res, base + 2u, V((uint32_t)S.loot), ag);
    W::gstore(res, base + 3u, S.hdm, ag);
    W::gstore(res, base + 4u, S.hef, ag);
    W::gstore(res, base + 5u, S.hhp, ag);
    W::gstore(res, base + 6u, V((uint32_t)S.frame), ag);
    W::gstore(res, base + 7u, V((uint32_t)S.outcome), ag);
  }

  // top of play()'s while(true): G:1444-1450
  static SF_DEV void loop_top(Arena &S, uint8_t *lds, const Params &p, int a) {
    SF_PROF(PH_TOP);
    spawns(S, lds, p);
    const int out = check_end(S, p);
    if (out != SF_RUNNING) {
      S.done = 1, S.outcome = out;
      latch_results(S, p, a);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // setup() G:1231-1277 + load_data() G:1741-1925 for this arena (the caller then does `++frame` and the first loop top).
  // `adopt`: the generator state for this seed was already warmed up in S.rl2 (see prewarm)
  static SF_DEV void reset_state(Arena &S, uint8_t *lds, const Params &p, uint64_t tb, uint64_t serial, bool adopt) {
    S.frame = S.kills = S.tkills = S.loot = S.chests = S.steps = 0;
    S.done = 0, S.outcome = SF_RUNNING;
    S.tb_lo = (uint32_t)tb, S.tb_hi = (uint32_t)(tb >> 32), S.sr_lo = (uint32_t)serial, S.sr_hi = (uint32_t)(serial >> 32);
    S.hpos = V(POS_NONE), S.hfl = V(0u), S.hhp = V(0u), S.hst = V(0u), S.hmd = V(0u), S.hk = V(0u), S.hdm = V(0u);
    S.hef = V(0u), S.hc01 = V(0u), S.hc23 = V(0u), S.ht01 = V(0u), S.ht23 = V(0u), S.hbpk = V(0u);
    S.hcmd = V((uint32_t)'+');
    S.zpos = V(0u), S.zhp = V(0u), S.zmd = V(0u);
    if constexpr (ZL) S.zwn = 0u;  // (store() clears the words the episode before had in use, zwhi)
#pragma unroll
    for (int j = 0; j < NB; ++j) S.ba[j] = V(0u), S.bd[j] = V(0u), S.bb[j] = V(0u), S.bc[j] = V(0u);
    if constexpr (ZL) {
      S.ppos = V(0u);
      S.pwn = (uint32_t)p.pw0;  // the map's own exits fill the first words
      for (uint32_t j = 0; j < S.pwn; ++j) {
        const V sl = W::lane() + 64u * j;
        W::lds_store_u32(S.pl + 64u * j, W::lane(), W::gload(p.map_exits, sl, W::ltu(sl, (uint32_t)p.P)), W::all());
      }
      if (S.pwhi < S.pwn) S.pwhi = S.pwn;
    } else {
      S.ppos = W::gload(p.map_exits, W::lane(), W::ltu(W::lane(), (uint32_t)p.P));
    }
    W::copy_g2l(lds, p.map_flags, (uint32_t)p.cells_pad);
    S.dirty = 1u;
    if (adopt) {
      prewarm(S, lds, p, 1024u);  // whatever is still missing
      S.rl = S.rl2, S.rseed = S.rseed2, S.jomle = 18u + 1024u;
      draw_issue(S, p);
    } else {
      srand_(S, lds, p, tb, serial);
    }
    if (p.auto_reset) {  // arm the warm-up of the episode after this one
      seed_digits(S.rseed2, tb + (uint64_t)(uint32_t)p.reseed, 1u);
      S.rl2 = V(RL_ZERO);
      S.warm = 0u, S.la2_ok = 0u;
    }
    // load_data(): who stands where.  Solo/Timer: the player at (0,1,1) (G:1905-1920).  Squad: the player at (0,3,1),
    // four team-mates at (0,1,2..5), five opponents at (squad_floor,1,6..10), all built from the NPC record
    // (G:1861-1903).  Battle: every commanded human, placed below on random '.' cells (G:1846-1859).
    const int count = p.mode == SF_MODE_BATTLE ? p.n_agents : (p.mode == SF_MODE_SQUAD ? 10 : 1);
    SF_NOUNROLL for (int i = 0; i < count; ++i) {
      int prof = 0, team = 1;
      uint32_t q = pos_pack(0, 1, 1), fl = HF_CTRL;
      if (p.mode == SF_MODE_BATTLE) {
        q = POS_NONE, team = p.tab->teams[i], fl = HF_CTRL | (i != p.ind ? HF_REMOTE : 0u);
      } else if (p.mode == SF_MODE_SQUAD) {
        prof = i ? 1 : 0;
        team = i < 5 ? 1 : 2;
        q = i == 0 ? pos_pack(0, 3, 1) : pos_pack(i < 5 ? 0 : p.squad_floor, 1, i + 1);
        fl = (i == 0 || i < p.n_agents) ? HF_CTRL : 0u;  // USE_AGENT_IN_SQUAD_NPCS G:1883-1885
      }
      // every commanded human of a Battle match is built from its own record when the match brought them
      human_make(S, p, (uint32_t)i, prof, q, 1, team, fl, (p.mode == SF_MODE_BATTLE && p.npc_block > 1) ? i : 0);
    }
    if (p.mode == SF_MODE_BATTLE) {
      S.hfl = S.hfl & ~HF_OCC;  // not on the map until placed
      SF_NOUNROLL for (int i = 0; i < p.n_agents; ++i) {
        const uint32_t way = draw(S, lds, p) % 4u + 1u;
        SF_NOUNROLL for (int guard = 0; guard < (1 << 20); ++guard) {  // `while(true)` with an exit every wave reaches
          const uint32_t q = draw_cell(S, lds, p);
          uint32_t fl;
          if (showit_q(S, lds, p, q, fl) == SH_EMPTY) {
            W::setlane(S.hpos, (uint32_t)i, q);
            W::setlane(S.hfl, (uint32_t)i, ((W::readlane(S.hfl, (uint32_t)i) & ~HF_WAY_MASK) | way) | HF_OCC);
            break;
          }
        }
      }
    }
  }

  // One iteration of the loop body G:1452-1471 followed by the next loop top.
  // PHASE 0: all of it (the throughput path).  PHASE 1 / 2: the iteration cut where the reference queries the agents of
  // humans other than `ind` (get_command inside human_action, G:988-999): 1 = zombie_action ... the first update_bull
  // (G:1455-1463), 2 = human_action ... the second update_bull and the next loop top (G:1464-1471,1444-1450).
  template <int PHASE = 0>
  static SF_DEV void step(Arena &S, uint8_t *lds, const Params &p, int a) {
    if (S.done) return;
    // the two half-ticks share `update_tmp; hit_human; hit_zombie; ++frame; update_bull` (G:1457-1463,1465-1471)
    SF_STAMP(S, 0);
    uint32_t rest = 0u;  // draws outside the five drawing phases (none: portal_damage, update_tmp and the hits draw nothing)
    SF_NOUNROLL for (int half = (PHASE == 2 ? 1 : 0); half < (PHASE == 1 ? 1 : 2); ++half) {
      const uint32_t j0 = S.jomle;
      if (half == 0) {
        zombie_action(S, lds, p);
        SF_STAMP(S, 1);
      } else {
        human_action(S, lds, p, a);
        SF_STAMP(S, 3);
      }
      const uint32_t j1 = S.jomle;
      if (half == 0) {
        portal_damage(S, lds, p);
        SF_STAMP(S, 2);
      }
      if (p.auto_reset) prewarm(S, lds, p, S.wrate);  // warm-up draws of the next episode, spread over the step so that
      SF_STAMP(S, 4);
      update_tmp(S, lds, p, a);             // each one's table lookup is in flight while the tick goes on
      SF_STAMP(S, 5);
      hits(S, p);
      SF_STAMP(S, 6);
      ++S.frame;  // updmap G:489-495 clears render-only bits
      if (p.auto_reset) prewarm(S, lds, p, S.wrate);
      SF_STAMP(S, 4);
      const uint32_t j2 = S.jomle;
      update_bull(S, lds, p);
      SF_STAMP(S, 7);
      const uint32_t w = ((j1 - j0) & 0xffffu) | ((S.jomle - j2) << 16);
      rest += j2 - j1;
      if (half == 0)
        S.pd01 = w;
      else
        S.pd23 = w;
    }
    if (PHASE == 1) return;
    ++S.steps;
    // the loop top; when the episode ends and auto_reset is on, once more for the episode that begins
    SF_NOUNROLL for (int pass = 0; pass < 2; ++pass) {
      const uint32_t j3 = S.jomle;
      loop_top(S, lds, p, a);
      if (pass == 0) S.pd45 = ((S.jomle - j3) & 0xffffu) | (rest << 16);
      if (!S.done || pass == 1) break;
      if (S.ended < 255) ++S.ended;  // episodes that ended during this launch (sf_done with auto_reset)
      ++S.episodes;
      if (!p.auto_reset) break;
      const uint64_t tb = (((uint64_t)S.tb_hi << 32) | S.tb_lo) + (uint64_t)(uint32_t)p.reseed;
      const uint64_t sr = ((uint64_t)S.sr_hi << 32) | S.sr_lo;
      const int32_t ep = S.episodes;
      reset_state(S, lds, p, tb, sr, true);
      S.episodes = ep;
      ++S.frame;  // G:1441
    }
    SF_STAMP(S, 8);
  }

  // ------------------------------------------------------------------------------------------------
  // HBM <-> registers / LDS
  // copy_plane = false: the caller has the flag plane's loads in flight already (step_body) and stores them itself
  static SF_DEV void load(Arena &S, uint8_t *lds, const Params &p, int a, bool copy_plane = true) {
    const V ln = W::lane();
    const size_t AH = (size_t)p.A * (size_t)p.H, AZ = (size_t)p.A * (size_t)p.Z, AB = (size_t)p.A * (size_t)p.B;
    {
      const P in = W::ltu(ln, (uint32_t)p.H);
      const uint32_t *h = p.hum + (size_t)a * (size_t)p.H;
      S.hpos = W::gload(h + HW_POS * AH, ln, in), S.hfl = W::gload(h + HW_FLAGS * AH, ln, in);
      S.hhp = W::gload(h + HW_HP * AH, ln, in), S.hst = W::gload(h + HW_STAMINA * AH, ln, in);
      S.hmd = W::gload(h + HW_MINDAMAGE * AH, ln, in), S.hk = W::gload(h + HW_KILLS * AH, ln, in);
      S.hdm = W::gload(h + HW_DAMAGE * AH, ln, in), S.hef = W::gload(h + HW_EFFECT * AH, ln, in);
      S.hc01 = W::gload(h + HW_CONS01 * AH, ln, in), S.hc23 = W::gload(h + HW_CONS23 * AH, ln, in);
      S.ht01 = W::gload(h + HW_THR01 * AH, ln, in), S.ht23 = W::gload(h + HW_THR23 * AH, ln, in);
      S.hbpk = W::gload(h + HW_BPK * AH, ln, in);
      S.hcmd = V((uint32_t)'+');
    }
    if constexpr (ZL) {
      S.zpos = V(0u), S.zhp = V(0u), S.zmd = V(0u);
      const uint32_t *z = p.zom + (size_t)a * (size_t)p.Z;
      S.zwn = (uint32_t)W::uload_i32(p.scal + (size_t)a * SC_WORDS + SC_ZWN);
      if (S.zwn > (uint32_t)zw_for(p.Z)) S.zwn = (uint32_t)zw_for(p.Z);
      S.zwhi = S.zwn;
      for (uint32_t j = 0; j < S.zwn; ++j) {
        const V sl = ln + 64u * j;
        const P in = W::ltu(sl, (uint32_t)p.Z);
        zl_put(S, ZW_POS, j, W::gload(z + ZW_POS * AZ, sl, in), W::all());
        zl_put(S, ZW_HP, j, W::gload(z + ZW_HP * AZ, sl, in), W::all());
        zl_put(S, ZW_MINDAMAGE, j, W::gload(z + ZW_MINDAMAGE * AZ, sl, in), W::all());
      }
    } else {
      const P in = W::ltu(ln, (uint32_t)p.Z);
      const uint32_t *z = p.zom + (size_t)a * (size_t)p.Z;
      S.zpos = W::gload(z + ZW_POS * AZ, ln, in), S.zhp = W::gload(z + ZW_HP * AZ, ln, in);
      S.zmd = W::gload(z + ZW_MINDAMAGE * AZ, ln, in);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const V sl = ln + (uint32_t)(64 * j);
      const P in = W::ltu(sl, (uint32_t)p.B);
      const uint32_t *b = p.bul + (size_t)a * (size_t)p.B;
      S.ba[j] = W::gload(b + BW_A * AB, sl, in), S.bd[j] = W::gload(b + BW_DAMAGE * AB, sl, in);
      S.bb[j] = W::gload(b + BW_B * AB, sl, in), S.bc[j] = W::gload(b + BW_C * AB, sl, in);
    }
    if constexpr (ZL) {
      S.ppos = V(0u);
      S.pwn = (uint32_t)W::uload_i32(p.scal + (size_t)a * SC_WORDS + SC_PWN);
      if (S.pwn > (uint32_t)zw_for(p.P)) S.pwn = (uint32_t)zw_for(p.P);
      S.pwhi = S.pwn;
      for (uint32_t j = 0; j < S.pwn; ++j) {
        const V sl = ln + 64u * j;
        W::lds_store_u32(S.pl + 64u * j, ln, W::gload(p.por + (size_t)a * (size_t)p.P, sl, W::ltu(sl, (uint32_t)p.P)), W::all());
      }
    } else {
      S.ppos = W::gload(p.por + (size_t)a * (size_t)p.P, ln, W::ltu(ln, (uint32_t)p.P));
    }
    {
      // one dword per tap: random[i] | us[i] << 20 | seed[i] << 24  (RN:31; us/seed are decimal digits + 1)
      const P in = W::ltu(ln, 18u);
      const V rw = W::gload(p.rng + (size_t)a * RNG_WORDS, ln, in);
      const V val = rw & 0xfffffu;
      S.rus = (rw >> 20) & 15u, S.rseed = W::select(ln == 18u, V(1u), (rw >> 24) & 15u);
        const P nz = in & (val != 0u);
      S.rl = W::select(nz, W::gload_u16(p.logt, val + (uint32_t)LOGT_OFF, nz), V(RL_ZERO));
      const V rw2 = W::gload(p.rng2 + (size_t)a * RNG_WORDS, ln, in);  // log form: never dumped
      S.rl2 = rw2 & 0x1ffffu, S.rseed2 = W::select(ln == 18u, V(1u), rw2 >> 24);
    }
    const V sc = W::gload((const uint32_t *)p.scal + (size_t)a * SC_WORDS, ln, W::ltu(ln, (uint32_t)SC_WORDS));
    S.frame = (int32_t)W::readlane(sc, SC_FRAME), S.kills = (int32_t)W::readlane(sc, SC_KILLS);
    S.tkills = (int32_t)W::readlane(sc, SC_TKILLS), S.loot = (int32_t)W::readlane(sc, SC_LOOT);
    S.chests = (int32_t)W::readlane(sc, SC_CHESTS), S.jomle = W::readlane(sc, SC_JOMLE);
    S.steps = (int32_t)W::readlane(sc, SC_STEPS), S.episodes = (int32_t)W::readlane(sc, SC_EPISODES);
    S.done = (int32_t)W::readlane(sc, SC_DONE), S.outcome = (int32_t)W::readlane(sc, SC_OUTCOME);
    S.ended = (int32_t)W::readlane(sc, SC_ENDED);
    S.tb_lo = W::readlane(sc, SC_TB_LO), S.tb_hi = W::readlane(sc, SC_TB_HI);
    S.sr_lo = W::readlane(sc, SC_SR_LO), S.sr_hi = W::readlane(sc, SC_SR_HI);
    S.warm = W::readlane(sc, SC_WARM);
    S.pd01 = W::readlane(sc, SC_PD01), S.pd23 = W::readlane(sc, SC_PD23), S.pd45 = W::readlane(sc, SC_PD45);
    if (!HBM_PLANE && copy_plane) W::copy_g2l(lds, p.flags + (size_t)a * (size_t)p.cells_pad, (uint32_t)p.cells_pad);
    S.dirty = 0u;
    if (copy_plane) draw_issue(S, p);  // the lookup of the next draw (S.la) is not part of the stored state
  }

  static SF_DEV void store(const Arena &S, const uint8_t *lds, const Params &p, int a) {
    const V ln = W::lane();
    const size_t AH = (size_t)p.A * (size_t)p.H, AZ = (size_t)p.A * (size_t)p.Z, AB = (size_t)p.A * (size_t)p.B;
    {
      const P in = W::ltu(ln, (uint32_t)p.H);
      uint32_t *h = p.hum + (size_t)a * (size_t)p.H;
      W::gstore(h + HW_POS * AH, ln, S.hpos, in), W::gstore(h + HW_FLAGS * AH, ln, S.hfl, in);
      W::gstore(h + HW_HP * AH, ln, S.hhp, in), W::gstore(h + HW_STAMINA * AH, ln, S.hst, in);
      W::gstore(h + HW_MINDAMAGE * AH, ln, S.hmd, in), W::gstore(h + HW_KILLS * AH, ln, S.hk, in);
      W::gstore(h + HW_DAMAGE * AH, ln, S.hdm, in), W::gstore(h + HW_EFFECT * AH, ln, S.hef, in);
      W::gstore(h + HW_CONS01 * AH, ln, S.hc01, in), W::gstore(h + HW_CONS23 * AH, ln, S.hc23, in);
      W::gstore(h + HW_THR01 * AH, ln, S.ht01, in), W::gstore(h + HW_THR23 * AH, ln, S.ht23, in);
      W::gstore(h + HW_BPK * AH, ln, S.hbpk, in);
    }
    uint32_t zlive_n = 0u;
    (void)zlive_n;
    if constexpr (ZL) {
      uint32_t *z = p.zom + (size_t)a * (size_t)p.Z;
      for (uint32_t j = 0; j < S.zwhi; ++j) {  // words an earlier episode of this launch had in use are cleared in HBM
        const V sl = ln + 64u * j;
        const P in = W::ltu(sl, (uint32_t)p.Z);
        const bool used = j < S.zwn;
        const V zpw = used ? zl_get(S, ZW_POS, j) : V(0u);
        W::gstore(z + ZW_POS * AZ, sl, zpw, in);
        W::gstore(z + ZW_HP * AZ, sl, used ? zl_get(S, ZW_HP, j) : V(0u), in);
        W::gstore(z + ZW_MINDAMAGE * AZ, sl, used ? zl_get(S, ZW_MINDAMAGE, j) : V(0u), in);
        zlive_n += (uint32_t)W::popc64(W::ballot((zpw & ZF_ALIVE) != 0u));
      }
    } else {
      const P in = W::ltu(ln, (uint32_t)p.Z);
      uint32_t *z = p.zom + (size_t)a * (size_t)p.Z;
      W::gstore(z + ZW_POS * AZ, ln, S.zpos, in), W::gstore(z + ZW_HP * AZ, ln, S.zhp, in);
      W::gstore(z + ZW_MINDAMAGE * AZ, ln, S.zmd, in);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const V sl = ln + (uint32_t)(64 * j);
      const P in = W::ltu(sl, (uint32_t)p.B);
      uint32_t *b = p.bul + (size_t)a * (size_t)p.B;
      W::gstore(b + BW_A * AB, sl, S.ba[j], in), W::gstore(b + BW_DAMAGE * AB, sl, S.bd[j], in);
      W::gstore(b + BW_B * AB, sl, S.bb[j], in), W::gstore(b + BW_C * AB, sl, S.bc[j], in);
    }
    if constexpr (ZL) {
      for (uint32_t j = 0; j < S.pwhi; ++j) {
        const V sl = ln + 64u * j;
        W::gstore(p.por + (size_t)a * (size_t)p.P, sl, j < S.pwn ? pl_get(S, j) : V(0u), W::ltu(sl, (uint32_t)p.P));
      }
    } else {
      W::gstore(p.por + (size_t)a * (size_t)p.P, ln, S.ppos, W::ltu(ln, (uint32_t)p.P));
    }
    {
      // a stored generator is always warmed up (k_reset runs _srand's 1024 draws), so no tap is zero; bit 16 of the
      // hot state is not a flag (see draw())
      const P in = W::ltu(ln, 18u);
      const V val = pow3_v(S.xt, S.rl & 0xffffu, in);
      W::gstore(p.rng + (size_t)a * RNG_WORDS, ln, val | (S.rus << 20) | (S.rseed << 24), in);
      W::gstore(p.rng2 + (size_t)a * RNG_WORDS, ln, (S.rl2 & 0x1ffffu) | (S.rseed2 << 24), in);
    }
    V sc = V(0u);
    W::setlane(sc, SC_FRAME, (uint32_t)S.frame), W::setlane(sc, SC_KILLS, (uint32_t)S.kills);
    W::setlane(sc, SC_TKILLS, (uint32_t)S.tkills), W::setlane(sc, SC_LOOT, (uint32_t)S.loot);
    W::setlane(sc, SC_CHESTS, (uint32_t)S.chests), W::setlane(sc, SC_JOMLE, S.jomle);
    W::setlane(sc, SC_STEPS, (uint32_t)S.steps), W::setlane(sc, SC_EPISODES, (uint32_t)S.episodes);
    W::setlane(sc, SC_DONE, (uint32_t)S.done), W::setlane(sc, SC_OUTCOME, (uint32_t)S.outcome);
    W::setlane(sc, SC_ENDED, (uint32_t)S.ended);
    W::setlane(sc, SC_TB_LO, S.tb_lo), W::setlane(sc, SC_TB_HI, S.tb_hi);
    W::setlane(sc, SC_SR_LO, S.sr_lo), W::setlane(sc, SC_SR_HI, S.sr_hi);
    W::setlane(sc, SC_DRAWS, S.jomle - (18u + 1024u));  // _rand() calls since the episode's _srand: jomle counts them
    W::setlane(sc, SC_WARM, S.warm);
    W::setlane(sc, SC_PD01, S.pd01), W::setlane(sc, SC_PD23, S.pd23), W::setlane(sc, SC_PD45, S.pd45);
    if constexpr (ZL) {
      W::setlane(sc, SC_LOAD, zlive_n + (uint32_t)W::popc64(W::ballot(((S.hfl & HF_ALIVE) != 0u) & W::ltu(ln, (uint32_t)p.H))));
      W::setlane(sc, SC_ZWN, S.zwn);
      W::setlane(sc, SC_PWN, S.pwn);
    } else {
      W::setlane(sc, SC_LOAD, (uint32_t)(W::popc64(W::ballot(((S.zpos & ZF_ALIVE) != 0u) & W::ltu(ln, (uint32_t)p.Z))) +
                                         W::popc64(W::ballot(((S.hfl & HF_ALIVE) != 0u) & W::ltu(ln, (uint32_t)p.H)))));
    }
    W::gstore((uint32_t *)p.scal + (size_t)a * SC_WORDS, ln, sc, W::ltu(ln, (uint32_t)SC_WORDS));
    if (!HBM_PLANE && S.dirty) W::copy_l2g(p.flags + (size_t)a * (size_t)p.cells_pad, lds, (uint32_t)p.cells_pad);
  }

  // ------------------------------------------------------------------------------------------------
  // kernel bodies
  // LDS layout of a workgroup (= one wavefront = one arena): [power table : 2 KiB][flag plane : cells_pad] (the table
  // first, so that its address is a compile-time DS offset); with HBM_PLANE only the power table, and `lds` (the
  // plane) is the arena's slice of Params::flags
  static SF_DEV uint8_t *tables(Arena &S, uint8_t *lds, const Params &p, int a, bool copy = true) {
    uint8_t *tab = lds;
    if (copy) {
      W::copy_g2l(tab, reinterpret_cast<const uint8_t *>(p.exptab), (uint32_t)LDS_EXP_BYTES);
      W::copy_g2l(tab + LDS_EXP_BYTES, reinterpret_cast<const uint8_t *>(p.tab->hatab), (uint32_t)p.ht_bytes);
    }
    S.xt = reinterpret_cast<const uint32_t *>(tab);
    S.ht = reinterpret_cast<const uint32_t *>(tab + LDS_EXP_BYTES);
    S.bm = nullptr;
    if (BITMAPS) {
      S.bm = reinterpret_cast<uint32_t *>(lds + p.lds_tab + (HBM_PLANE ? 0 : p.cells_pad));
      W::lds_zero(S.bm, (uint32_t)(BM_COUNT * p.bm_words));
    }
    S.zl = nullptr, S.zcap = 0u, S.zwn = 0u, S.zwhi = 0u;
    S.pl = nullptr, S.pwn = 0u, S.pwhi = 0u;
    if constexpr (ZL) {
      S.zl = reinterpret_cast<uint32_t *>(lds + p.lds_tab + (HBM_PLANE ? 0 : p.cells_pad) + (BITMAPS ? BM_COUNT * 4 * p.bm_words : 0));
      S.zcap = 64u * (uint32_t)zw_for(p.Z);
      S.pl = S.zl + ZW_WORDS * S.zcap;
    }
    S.la = V(0u);
    S.la2 = V(0u), S.la2_ok = 0u;
    return HBM_PLANE ? p.flags + (size_t)a * (size_t)p.cells_pad : lds + p.lds_tab;
  }

  static SF_DEV void reset_body(uint8_t *lds, const Params &p, int a, const uint64_t *tb, const uint64_t *serial) {
    Arena S;
    S.episodes = 0, S.ended = 0;
    S.pd01 = S.pd23 = S.pd45 = 0u;
    lds = tables(S, lds, p, a);
    S.rl2 = V(RL_ZERO), S.rseed2 = V(0u), S.warm = 0u, S.wrate = 1u;
    if constexpr (ZL) S.zwhi = (uint32_t)zw_for(p.Z), S.pwhi = (uint32_t)zw_for(p.P);  // store() writes the whole tables once: slots beyond zwn / pwn are zero in HBM from here on
    reset_state(S, lds, p, tb[a], serial[a], false);
    ++S.frame;  // G:1441
    loop_top(S, lds, p, a);
    store(S, lds, p, a);
  }

  // cmds: [k][A][n_agents]
  static SF_DEV void step_body(uint8_t *lds, const Params &p, int a, const uint8_t *cmds, int k) {
    Arena S;
    // The launch's loads — the two tables, the flag plane's first 4 KB, the arena's state — are all issued before the
    // first of them is waited for: one round trip to memory (two with the generator's log lookup, which needs its state
    // word) where the copies used to wait one by one, ~5 us of every launch (a third of a one-step launch's overhead).
    typename W::template G2L<LDS_EXP_BYTES / 1024> q_exp;
    typename W::template G2L<2> q_ht;
    typename W::template G2L<4> q_pl;
    uint8_t *const tab0 = lds;
    const uint32_t ht_n = (uint32_t)p.ht_bytes < 2048u ? (uint32_t)p.ht_bytes : 2048u;
    const uint32_t pl_n = HBM_PLANE ? 0u : ((uint32_t)p.cells_pad < 4096u ? (uint32_t)p.cells_pad : 4096u);
    const uint8_t *const pl_g = p.flags + (size_t)a * (size_t)p.cells_pad;
    W::g2l_issue(q_exp, reinterpret_cast<const uint8_t *>(p.exptab), (uint32_t)LDS_EXP_BYTES);
    W::g2l_issue(q_ht, reinterpret_cast<const uint8_t *>(p.tab->hatab), ht_n);
    W::g2l_issue(q_pl, pl_g, pl_n);
    lds = tables(S, lds, p, a, false);
    load(S, lds, p, a, false);
    W::g2l_store(q_exp, tab0, (uint32_t)LDS_EXP_BYTES);
    W::g2l_store(q_ht, tab0 + LDS_EXP_BYTES, ht_n);
    if ((uint32_t)p.ht_bytes > ht_n)
      W::copy_g2l(tab0 + LDS_EXP_BYTES + ht_n, reinterpret_cast<const uint8_t *>(p.tab->hatab) + ht_n, (uint32_t)p.ht_bytes - ht_n);
    if (!HBM_PLANE) {
      W::g2l_store(q_pl, lds, pl_n);
      if ((uint32_t)p.cells_pad > pl_n) W::copy_g2l(lds + pl_n, pl_g + pl_n, (uint32_t)p.cells_pad - pl_n);
    }
    draw_issue(S, p);  // the lookup of the next draw (S.la) is not part of the stored state; it reads the tables in LDS
    const P ag = W::ltu(W::lane(), (uint32_t)p.n_agents);
    // An episode shorter than its successor's warm-up stalls its own restart on the missing draws, and a launch is as
    // slow as its slowest arena: 17 % of configs[1]'s episodes are shorter than the 256 steps that 4 draws per step
    // would need, 0.7 % shorter than the 64 steps of 16 per step.  Measured over 4 / 8 / 16 / 32 / 64 draws per step:
    // 16 is the best for long launches (+1.7 % over 4) and for one-step launches (p90 128 -> 28 us).
    S.wrate = 4u;
    S.ended = 0;
    SF_STAMP_LOADED();
    SF_STAMP_BEGIN(S);
    for (int s = 0; s < k; ++s) {
      const uint8_t *c = cmds + ((size_t)s * (size_t)p.A + (size_t)a) * (size_t)p.n_agents;
      S.hcmd = W::select(ag, W::gload_u8(c, W::lane(), ag), V((uint32_t)'+'));
      step(S, lds, p, a);
    }
    SF_STAMP_END(S, a);
    SF_STAMP_STEPPED();
    store(S, lds, p, a);
  }

  // One half of one iteration (sf_step_begin / sf_step_end); cmds: [A][n_agents], read by the second half only.
  static SF_DEV void step_half_body(uint8_t *lds, const Params &p, int a, const uint8_t *cmds, int phase) {
    Arena S;
    lds = tables(S, lds, p, a);
    load(S, lds, p, a);
    S.wrate = 4u;
    S.ended = 0;
    SF_STAMP_BEGIN(S);
    if (phase == 1) {
      S.hcmd = V((uint32_t)'+');  // (nothing reads a command before human_action)
      step<1>(S, lds, p, a);
    } else {
      const P ag = W::ltu(W::lane(), (uint32_t)p.n_agents);
      S.hcmd = W::select(ag, W::gload_u8(cmds + (size_t)a * (size_t)p.n_agents, W::lane(), ag), V((uint32_t)'+'));
      step<2>(S, lds, p, a);
    }
    SF_STAMP_END(S, a);
    store(S, lds, p, a);
  }
};

}  // namespace sf
