// sf_types.hpp — HBM data layout of the batched arena state, shared by the device core and the host API.
//
// Everything an arena owns is struct-of-arrays, arena-major / slot-minor, so that the wavefront that
// owns arena `a` loads field f of all its entities with one coalesced instruction
// (address = base_f + (a * CAP + lane) * 4).  See DESIGN.md §Layout.
#pragma once
#include <stddef.h>
#include <stdint.h>

// SF_HD: usable from host code and from gfx950 device code (hipcc defines the qualifiers; the
// test-only wave emulator under tests/emu is built with g++ and sees none).
#if defined(__HIPCC__)
#define SF_HD __host__ __device__
#else
#define SF_HD
#endif

namespace sf {

// ---- packed position: f << 20 | r << 10 | c (coordinates < 1024, random.hpp:62) ----------------------
constexpr uint32_t POS_MASK = 0x3FFFFFu;
constexpr uint32_t POS_NONE = 0x3FFFFFu;
SF_HD inline uint32_t pos_pack(int f, int r, int c) {
  return ((uint32_t)f << 20) | ((uint32_t)r << 10) | (uint32_t)c;
}
SF_HD inline int pos_f(uint32_t p) { return (int)((p >> 20) & 3u); }
SF_HD inline int pos_r(uint32_t p) { return (int)((p >> 10) & 1023u); }
SF_HD inline int pos_c(uint32_t p) { return (int)(p & 1023u); }

// ---- human record: 13 dwords -----------------------------------------------------------------------
enum { HW_POS = 0, HW_FLAGS, HW_HP, HW_STAMINA, HW_MINDAMAGE, HW_KILLS, HW_DAMAGE, HW_EFFECT, HW_CONS01, HW_CONS23,
       HW_THR01, HW_THR23, HW_BPK, HW_WORDS };
// HW_FLAGS bit layout
constexpr uint32_t HF_WAY_MASK = 7u;          // way 1..4 (gameplay.hpp:43)
constexpr int HF_TEAM_SH = 3;                 // 8 bits
constexpr uint32_t HF_ALIVE = 1u << 11;       // mh[i]
constexpr uint32_t HF_REMOTE = 1u << 12;      // remote[i]
constexpr uint32_t HF_RNPC = 1u << 13;        // Human::rnpc
constexpr uint32_t HF_PROF = 1u << 14;        // 0 player profile, 1 npc profile
constexpr uint32_t HF_CTRL = 1u << 15;        // active_agent: commanded through sf_step
constexpr int HF_VEC_SH = 16;                 // 2 bits: backpack.vec + 1
constexpr int HF_IND_SH = 18;                 // 4 bits: backpack.ind + 1
constexpr uint32_t HF_OCC = 1u << 22;         // the cell's s[0] designates this human (alive, or dead `ind`)
constexpr int HF_AGP_SH = 23;                 // 4 bits: which agent record a commanded human was built from (HF_PROF clear)
// HW_BPK: blocks | portals << 8 | (portal_ind + 1) << 16

// ---- zombie record: 3 dwords -----------------------------------------------------------------------
enum { ZW_POS = 0, ZW_HP, ZW_MINDAMAGE, ZW_WORDS };
constexpr uint32_t ZF_SUPER = 1u << 30, ZF_ALIVE = 1u << 31;

// ---- bullet record: 4 dwords -----------------------------------------------------------------------
enum { BW_A = 0, BW_DAMAGE, BW_B, BW_C, BW_WORDS };
// BW_A: pos | (way-1) << 22 | alive << 24 | ref << 25 ; BW_B: (uint16)effect | owner << 16 ; BW_C: range | traveled << 16
constexpr int BA_WAY_SH = 22;
constexpr uint32_t BA_ALIVE = 1u << 24, BA_REF = 1u << 25;

// ---- portal record: 1 dword: pos | active << 31 -----------------------------------------------------
constexpr uint32_t PF_ACTIVE = 1u << 31;

// ---- per-arena scalar block: 24 dwords ----------------------------------------------------------------
enum { SC_FRAME = 0, SC_KILLS, SC_TKILLS, SC_LOOT, SC_CHESTS, SC_JOMLE, SC_STEPS, SC_EPISODES, SC_DONE, SC_OUTCOME,
       SC_ENDED, SC_TB_LO, SC_TB_HI, SC_SR_LO, SC_SR_HI, SC_DRAWS, SC_WARM,
       SC_LOAD /* live zombies + live humans when the state was stored: k_rank's key, not part of the game state */,
       SC_ZWN /* large slot pools (sf_core.hpp ZL): 64-slot words of the zombie table in use */,
       SC_PWN /* ... and of the exit table */,
       // generator draws of the arena's last step by phase (sf_phase_draws), two 16-bit counts per word: zombie_action |
       // first update_bull << 16; human_action | second update_bull << 16; the next loop top's spawns | everything else << 16
       SC_PD01, SC_PD23, SC_PD45,
       SC_WORDS = 24 };
constexpr int RNG_WORDS = 18;

// ---- derived per-profile tables (Human::build, Character.hpp:650-709) ---------------------------------
struct Derived {
  int32_t hp, mindamage, stamina, mindamage_def;
  int32_t cd_punch;          // compute_damage(mindamage_def, 1)      Character.hpp:393
  int32_t blocks, portals;
  int32_t cons[4], thr_cnt[4];
  int32_t thr[4][4];         // stamina, damage, effect, range
  int32_t weapon[8][4];
  int32_t weapon_lvl[8];
  int32_t cd_weapon[8];      // compute_damage(w.damage, w.range)     Character.hpp:404
};

// ---- human_action's decode / stat table, staged in LDS (sf_core.hpp human_action) ------------------------------
// words: [HT_CMD .. +32)   128 bytes: command char -> class | param << 4 (obey's key classes, gameplay.hpp:695-821)
//        [HT_CONS .. +12)  cons_items[4][3]
//        [HT_PROF + 72 block .. ): cd_punch, weapon_lvl[8], cd_weapon[8], weapon[8][4], thr[4][4]   (Derived)
//        blocks: one per character record.  One shared player record: block 0 = player, block 1 = npc.  One record per
//        commanded human (sf_config.agent_profile): blocks 0..15 = agents, block 16 = npc.  Params::npc_block says which.
enum { HT_CMD = 0, HT_CONS = 32, HT_PROF = 44, HT_PROF_STRIDE = 72, HT_P_CDPUNCH = 0, HT_P_WLVL = 1, HT_P_CDW = 9,
       HT_P_WEAPON = 17, HT_P_THR = 49, MAX_PROFILE_BLOCKS = 17, HT_WORDS_MAX = HT_PROF + MAX_PROFILE_BLOCKS * HT_PROF_STRIDE };
inline int ht_bytes_for(int blocks) { return ((HT_PROF + blocks * HT_PROF_STRIDE) * 4 + 15) & ~15; }
enum { CL_NOP = 0, CL_SUICIDE, CL_BLOCK, CL_PORTAL, CL_TURN, CL_MOVE, CL_SELC, CL_SELT, CL_SELW, CL_USE, CL_PUNCH, CL_FIRE };

struct Tables {
  int32_t cons_items[4][3];  // stamina, Hp, effect
  Derived der[MAX_PROFILE_BLOCKS];  // by block: see HT_* (blocks above npc_block are unused)
  int32_t teams[16];         // BATTLE mode team of agent i
  int32_t obs_n;             // observation fast map: obs_out[i] = obs_map(obs_in[i]), filled on the host
  float obs_in[16], obs_out[16];
  // the finished 32-feature records of the eight kinds of plain static cell ('#', '^', 'v', 'O', chest types 0-3 with
  // nothing on them) and which of their features are non-zero: the same for every arena and every call
  float class_rec[8][32];
  uint32_t class_mask[8];
  uint32_t hatab[HT_WORDS_MAX];  // see HT_* above; the used part (Params::ht_bytes) is copied to LDS by every launch
};

// ---- everything a kernel needs -----------------------------------------------------------------------
struct Params {
  int32_t A, F, N, M, cells, cells_pad;
  int32_t bm_words;  // words per cell bitmap (bm_words_for(cells_pad))
  int32_t H, Z, B, P, C;
  int32_t mode, level, n_agents, auto_reset, reseed, timer_lim, squad_floor;
  int32_t ind;     // the human slot this process plays (`ind`, gameplay.hpp:39); 0 except in Battle matches
  int32_t npc_block;  // table block of the NPC record: 1 (one shared player record in block 0) or 16 (one per agent)
  int32_t ht_bytes;   // used bytes of Tables::hatab
  int32_t lds_tab;    // bytes of [exptab][hatab] at the start of LDS: the flag plane follows
  int32_t pw0;        // large pools (sf_core.hpp ZL): 64-slot words of the exit table that the map's own exits fill.  (In the
                      // four bytes that padded `tab` to its alignment: the struct, and with it every kernel's argument
                      // offsets and register allocation, stays as it was before the large pools existed)
  const Tables *tab;
  uint32_t *hum;   // [HW_WORDS][A][H]
  uint32_t *zom;   // [ZW_WORDS][A][Z]
  uint32_t *bul;   // [BW_WORDS][A][B]
  uint32_t *por;   // [A][P]
  uint32_t *rng;   // [A][RNG_WORDS]
  uint32_t *rng2;  // [A][RNG_WORDS]  the next episode's generator being warmed up (log form)
  int32_t *scal;   // [A][SC_WORDS]
  int32_t *results;// [A][n_agents][8]
  uint8_t *flags;  // [A][cells_pad]
  int32_t *aux_dmg;   // [A][cells]   valid only where SF_CELL_TEMP
  int16_t *aux_pidx;  // [A][cells]   valid only where SF_CELL_TEMP|SF_CELL_PIN_UP
  const uint8_t *map_flags;  // [cells_pad]
  const int16_t *map_pidx;   // [cells]
  const uint32_t *map_exits; // [P]
  const uint16_t *logt;      // [LOGT_OFF + 65537] log_3 of the residue (j - LOGT_OFF) mod 65537 at index j, 0 for residue 0
                             // (RNG, sf_core.hpp draw())
  const uint32_t *exptab;    // [512] 3^i (i < 256) then 3^(256 i): staged in LDS behind the flag plane
  // k_step only: workgroup i steps arena perm[i] (null: arena i).  Arenas are independent, so the order changes nothing
  // but which arenas share a SIMD: k_rank orders them by population, busiest first, so that every SIMD gets one arena
  // of each load quartile instead of whatever the arena numbers bring together (sf_api.hip)
  const uint32_t *perm;
};

// The log table is indexed by the half-reduced tap sum t = lo16(x) - hi16(x), x < 2^25, i.e. t in (-512, 65536):
// entry t + LOGT_OFF holds log_3(t mod 65537), and the entry of residue 0 holds log_3(1) = 0, which is RN:58's
// `sum + (int)(sum == 0)`.  A value v in [1, 65536] is looked up at v + LOGT_OFF.
constexpr int LOGT_OFF = 1024;  // >= 532: the largest hi16 of a draw's biased tap sum (sf_core.hpp SUM_BIAS_LANE)
constexpr int LOGT_ENTRIES = LOGT_OFF + 65537;

// Per-arena scratch bitmaps in LDS, one bit per cell (sf_core.hpp "cell bitmaps"): only with the flag plane in LDS
constexpr int BM_COUNT = 3;
inline int bm_words_for(int cells);
// bitmaps are used when the three of them take at most 6 KiB of LDS per arena (maps up to 128 x 128 cells): with the
// flag plane in LDS (<= 12 KiB) that always holds, with the plane in HBM it admits 128 x 128 but not 256 x 256
inline bool use_bitmaps(int cells_pad) { return BM_COUNT * 4 * bm_words_for(cells_pad) <= 6 * 1024; }
inline int bm_words_for(int cells) { return ((cells + 31) / 32 + 3) & ~3; }  // words per bitmap, 16-byte multiple

inline int nb_for(int B) { return (B + 63) / 64; }
// Zombie or exit pools of more than 64 slots (the reference's hold 9000, gameplay.hpp:37,51-53) live in LDS instead of
// one register lane per slot (Core<.., ZL>): [ZW_WORDS][64 * zombie words] + [64 * exit words] dwords behind the bitmaps
SF_HD inline bool large_pools(int Z, int P) { return Z > 64 || P > 64; }
SF_HD inline int zw_for(int n) { return (n + 63) / 64; }  // 64-slot words of a table of n slots
inline size_t zl_bytes_for(int Z, int P) {
  return large_pools(Z, P) ? 4u * 64u * ((size_t)ZW_WORDS * (size_t)zw_for(Z) + (size_t)zw_for(P)) : 0u;
}
constexpr int LDS_EXP_BYTES = 2048;                                   // exptab, then Tables::hatab (Params::lds_tab)
inline int lds_tab_for(int blocks) { return LDS_EXP_BYTES + ht_bytes_for(blocks); }
inline bool hbm_plane(int cells_pad);
// LDS of a workgroup: [exptab][hatab][flag plane unless it stays in HBM][bitmaps if used][zombie table if Z > 64]
inline size_t lds_zl_offset(int cells_pad, int lds_tab) {
  return (size_t)lds_tab + (hbm_plane(cells_pad) ? 0u : (size_t)cells_pad) +
         (use_bitmaps(cells_pad) ? (size_t)BM_COUNT * 4u * (size_t)bm_words_for(cells_pad) : 0u);
}
inline size_t lds_bytes_for(int cells_pad, int lds_tab, int Z, int P) {
  return lds_zl_offset(cells_pad, lds_tab) + zl_bytes_for(Z, P);
}
// flag planes above this size stay in HBM (Core<.., HBM_PLANE>): staging them would leave < 12 wavefronts per CU
constexpr int LDS_PLANE_MAX = 12 * 1024;
inline bool hbm_plane(int cells_pad) { return cells_pad > LDS_PLANE_MAX; }

}  // namespace sf
