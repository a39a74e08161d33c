/*
 * strikeforce.h — C-ABI of the MI355X-native batched StrikeForce arena simulator.
 *
 * This is the drop-in boundary for the reference's per-tick gameplay path.  The reference
 * (bistoyek21-ric/StrikeForce, all paths relative to StrikeForce-client/) has no FFI: its
 * boundary is compile-time — `class Agent` (bots/bot-0.5/Agent.hpp:178,217,239,266;
 * minimal form bots/bot-0/Agent.hpp:27-37) plus three declared-only members of
 * `struct gameplay`: `char bot(Human&) const` (gameplay.hpp:477), `void prepare(Human&)`
 * (gameplay.hpp:481), `void view() const` (gameplay.hpp:485).  Each entry point below names
 * the reference code it replaces.  `include/sf_agent_adapter.hpp` rebuilds the reference's
 * Agent contract on top of these calls so that existing Agent.hpp bots drop in.
 *
 * Plain C, plain pointers and sizes, no torch / HIP types in any signature.
 * All integer game state is bit-identical to the reference CPU loop under the same
 * (tb, serial) seed; see DESIGN.md for the documented deviations (frame clock for Timer).
 */
#ifndef STRIKEFORCE_H
#define STRIKEFORCE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_ABI_VERSION 2

/* Observation geometry — bots/bot-0.5/Custom.hpp:137-159 (32 channels, 31x31 window). */
#define SF_OBS_CHANNELS 32
#define SF_OBS_WINDOW 31
#define SF_OBS_FLOATS (SF_OBS_CHANNELS * SF_OBS_WINDOW * SF_OBS_WINDOW) /* 30752 */

/* Slot caps supported by the device kernels.  Humans: one wavefront lane each; bullets: four registers per lane.
 * Zombies and portal exits: the reference's own pool size (gameplay.hpp:37 `Z = 9000`, :51-53 `portal[B]`); pools of
 * more than 64 slots live in the arena's LDS (12 B per zombie slot, 4 B per exit) and sf_create refuses a
 * configuration whose flag plane and tables do not fit the CU's 160 KiB.  A whole Timer game of the reference on its
 * shipped maps holds at most 22 humans and 25 bullets at once (level 10, 37 500 steps), but 564 zombies and ~290 exits. */
#define SF_MAX_HUMANS 64
#define SF_MAX_ZOMBIES 9000
#define SF_MAX_BULLETS 256
#define SF_MAX_PORTALS 9000
#define SF_MAX_AGENTS 16
#define SF_MAX_COORD 1024 /* random.hpp:62 — _rand() is 10 bits, so rand()%N never reaches 1024 */

/* Game modes — gameplay.hpp:1530,1567,1604,1642,1660 (`mode` strings). */
enum { SF_MODE_SOLO = 0, SF_MODE_TIMER = 1, SF_MODE_SQUAD = 2, SF_MODE_BATTLE = 3 };

/* Episode outcome codes written to sf_results()[..][7] — gameplay.hpp:1102-1229 (check_end). */
enum { SF_RUNNING = 0, SF_DIED = 1, SF_WON = 2, SF_TIME_LOST = 3, SF_TIME_WON = 4, SF_QUIT = 5 };

/* Status codes (API misuse only; gameplay "failures" stay silent no-ops as in the reference). */
enum {
  SF_OK = 0,
  SF_ERR_ARG = -1,      /* null pointer / out-of-range config */
  SF_ERR_DEVICE = -2,   /* no HIP device or a HIP runtime error: the product never falls back to CPU */
  SF_ERR_MEMORY = -3,
  SF_ERR_STATE = -4     /* call order (e.g. step before reset) */
};

/* Cell flag byte (one per cell per arena) — the persistent subset of `node::s`
 * (gameplay.hpp:237-241).  s[0] human / s[1] zombie / s[2] bullet are derived from the
 * entity tables; s[8] corpse and s[9] hit-flash are render-only. */
enum {
  SF_CELL_WALL = 0x01,    /* s[3]  '#' */
  SF_CELL_TEMP = 0x02,    /* s[10] destructible (player-built) */
  SF_CELL_PIN_UP = 0x04,  /* s[5]  '^' portal entrance */
  SF_CELL_PIN_DN = 0x08,  /* s[6]  'v' portal entrance */
  SF_CELL_POUT = 0x10,    /* s[7]  'O' portal exit */
  SF_CELL_CHEST = 0x20,   /* s[4]  '?' */
  SF_CELL_CONS_SHIFT = 6  /* bits 6-7: consumable type of the chest (gameplay.hpp:539) */
};

/* A character record in the reference's 33-token text format minus the name
 * (Character.hpp:650-709 `Human::build`; files character/human.txt, character/human_enemy.txt,
 * accounts/game/<user>/info, <user>.txt). */
typedef struct sf_profile {
  int32_t def_hp, mindamage_def, def_stamina;
  int32_t level_solo, level_timer, level_squad;
  int32_t money;
  int32_t rate_solo, rate_timer, rate_squad, rate;
  int32_t cons[4];          /* owned consumables */
  int32_t throw_lvl_cnt[4][2]; /* (level, count) per throwable */
  int32_t weapon_lvl[8];    /* 0 = not owned */
  int32_t backpack_lvl;
} sf_profile;

/* Item stat tables — Item.hpp:69-74,113-118,149-154; files Items/{cons,throw,w}*.txt. */
typedef struct sf_items {
  int32_t cons[4][3];   /* stamina, Hp, effect */
  int32_t thr[4][4];    /* stamina, damage, effect, range */
  int32_t weapon[8][4]; /* stamina, damage, effect, range (level 0; each owned level +50/-50) */
} sf_items;

typedef struct sf_config {
  int32_t abi_version;      /* SF_ABI_VERSION */
  int32_t arenas;           /* independent worlds on this device */
  int32_t floors, rows, cols; /* gameplay.hpp:37 F, N, M */
  int32_t cap_humans, cap_zombies, cap_bullets, cap_portals, cap_chests; /* gameplay.hpp:37 H, Z, B, (B), C */
  int32_t mode;             /* SF_MODE_* */
  int32_t level;            /* 1..10, gameplay.hpp:459 L */
  int32_t n_agents;         /* humans commanded through sf_step */
  int32_t ind;              /* which of them is the reference's `ind` (gameplay.hpp:39): the player whose death ends the
                               episode and whose team the kill/loot counters follow.  0 except in lock-step Battle matches,
                               where the match server assigns it (server.cpp:243-246) */
  int32_t agent_team[SF_MAX_AGENTS]; /* BATTLE mode teams (server.cpp:239-246); ignored otherwise */
  int32_t auto_reset;       /* re-seed (tb += reseed_stride) and restart an arena when its episode ends */
  int32_t reseed_stride;    /* 0 -> arenas; multi-GPU runs pass the global arena count so that shards never reuse a seed */
  int32_t timer_frames_per_level; /* frame clock replacing time(0) in Timer mode (gameplay.hpp:1145-1146); 0 -> 7500 */
  int32_t device;           /* HIP device ordinal */
  const char *map;          /* floors*rows*cols chars from {# . O ^ v}; map/floor*.txt, gameplay.hpp:1249-1274 */
  const int16_t *map_portal;/* per cell: exit number for '^'/'v' cells, ignored elsewhere */
  sf_profile player;        /* profile of every commanded human (Character::me) */
  sf_profile npc;           /* character/human_enemy.txt */
  sf_items items;
  /* ABI 2: one character record per commanded human, as the players of a lock-step match exchange them before the
   * first tick (give_info / get_info gameplay.hpp:120-151; blob = Human::log_file / scan_file Character.hpp:570-648).
   * 0: every commanded human is built from `player`; n_agents: commanded human i is built from agent_profile[i]
   * (and `player` is unused).  NPC humans always come from `npc`. */
  int32_t n_agent_profiles;
  sf_profile agent_profile[SF_MAX_AGENTS];
} sf_config;

typedef struct sf_env sf_env; /* opaque; one per GPU */

/* ---- state dump (parity tests; not on the hot path) ---------------------------------------- */
typedef struct sf_arena_hdr {
  int64_t frame, kills, teams_kills, loot, chests; /* gameplay.hpp:461 */
  int64_t jomle;            /* random.hpp:29 */
  int64_t tb, serial;       /* seed of the running episode */
  int64_t steps;            /* loop iterations completed in this episode */
  int64_t episodes;         /* episodes completed by this arena since sf_create */
  int32_t rng[18];          /* random.hpp:31 `random[]` */
  int32_t done, outcome;
} sf_arena_hdr;

typedef struct sf_human_rec {
  int32_t alive, remote, rnpc, profile; /* mh / remote bitsets gameplay.hpp:55; profile 0 player, 1 npc */
  int32_t f, r, c, way, team;
  int32_t hp, stamina, mindamage;
  int32_t kills, damage, effect;         /* Character.hpp:294 */
  int32_t vec, ind;                      /* Backpack selection Character.hpp:65 */
  int32_t cons[4], throw_cnt[4];
  int32_t blocks, portals, portal_ind;
} sf_human_rec;

typedef struct sf_zombie_rec {
  int32_t alive, f, r, c, hp, mindamage, super_;
} sf_zombie_rec;

typedef struct sf_bullet_rec {
  int32_t alive, f, r, c, way, traveled, damage, effect, range, owner /* human slot + 1, 0 = none */,
      ref /* 1 if the cell's bullet pointer designates this bullet (gameplay.hpp:814-816,1087-1089) */;
} sf_bullet_rec;

typedef struct sf_portal_rec {
  int32_t active, f, r, c;
} sf_portal_rec;

/* ---- lifecycle ----------------------------------------------------------------------------- */

/* Replaces: compile-time dims gameplay.hpp:37, download_items() Item.hpp:179-188, make_p()
 * random.hpp:33-40, Human::build Character.hpp:650-709.  Allocates all device state. */
int sf_create(const sf_config *cfg, sf_env **out);
int sf_destroy(sf_env *env);

/* Fills `cfg` with the reference's shipped tables (Items/ text files, character/human.txt as player,
 * character/human_enemy.txt as npc) and zeroes everything else. */
void sf_config_defaults(sf_config *cfg);

/* Replaces gameplay::setup() + load_data() (gameplay.hpp:1231-1277,1741-1925) for every arena:
 * reset slots, rebuild the map, Random::_srand(tb[a], serial[a]) (random.hpp:64-76; the libc
 * srand/rand derivation of `serial` at gameplay.hpp:1745-1746 is bypassed), place players per mode,
 * then run the first loop-top (spawns at frame 1, gameplay.hpp:1441-1450).
 * tb/serial: host arrays of `arenas` entries. */
int sf_reset(sf_env *env, const uint64_t *tb, const uint64_t *serial);

/* One iteration of gameplay::play()'s while(true) body, gameplay.hpp:1452-1471, followed by the next
 * iteration's loop-top (spawns + check_end, gameplay.hpp:1444-1450), for every arena.
 * cmd: host array [arenas][n_agents] of reference command chars (valid_commands gameplay.hpp:45, plus
 * '_' gameplay.hpp:696); replaces get_my_action()/client.recieve() (gameplay.hpp:939-963,170-193). */
int sf_step(sf_env *env, const uint8_t *cmd);

/* Same, `k` iterations in one launch, commands already resident in device memory
 * ([k][arenas][n_agents]).  This is the throughput path. */
int sf_step_device(sf_env *env, const uint8_t *d_cmd, int32_t k);

/* The same iteration in two calls, cut where the reference asks the agents of humans OTHER than `ind` for their command:
 * get_command(i) -> bot(hum[i]) runs inside human_action (gameplay.hpp:988-999), i.e. after zombie_action ... the first
 * update_bull (gameplay.hpp:1455-1463), whereas the player `ind` was asked at the loop top (gameplay.hpp:955-958).
 *   sf_step_begin          gameplay.hpp:1455-1463
 *   sf_observe...          what bot(hum[i]), i != ind, encodes (only built with USE_AGENT_IN_SQUAD_NPCS does the
 *                          reference have such agents: Squad mode, gameplay.hpp:1883-1885,1896-1898)
 *   sf_step_end(cmd)       gameplay.hpp:1464-1471 + the next loop top; cmd as for sf_step (the entry of `ind` being
 *                          the command chosen at the loop top)
 * sf_step == sf_step_begin + sf_step_end on the same commands, bit for bit.  Between the two calls only the observe /
 * dump / digest / agent_alive entry points may be used (SF_ERR_STATE otherwise). */
int sf_step_begin(sf_env *env);
int sf_step_end(sf_env *env, const uint8_t *cmd);
int sf_step_end_device(sf_env *env, const uint8_t *d_cmd);

/* Replaces Human::active_agent / deleteAgent (Character.hpp:291,333-338): out[arenas][n_agents], 1 while the commanded
 * human is alive and still driven through sf_step.  The reference deletes a dead human's Agent in hit_human
 * (gameplay.hpp:648-649) and never queries it again (gameplay.hpp:985,991: `mh[i]`, get_active_agent()); its slot may
 * be handed to a spawned NPC (h_ind, gameplay.hpp:216-221), which then runs on human_rnpc_bot: commands sent for a
 * dead agent are ignored and its observation is all zero.  (The player `ind` dying ends the episode: sf_done.) */
int sf_agent_alive(sf_env *env, uint8_t *out_host);
int sf_agent_alive_device(sf_env *env, uint8_t *d_out);

/* Replaces gameplay::bot() observation encoding, bots/bot-0.5/Custom.hpp:29-159, for every
 * (arena, agent): out[arenas][n_agents][32][31][31] float32. */
int sf_observe(sf_env *env, float *out_host);
int sf_observe_device(sf_env *env, float *d_out);
/* The same observation for a caller that keeps ONE device buffer per env and never writes to it: only the floats
 * that were non-zero after the previous call on this buffer, or are non-zero now, are written (an observation is
 * about 1 % non-zero, so this is a small fraction of the 123 KB per agent).  The first call on a buffer, a call with
 * another pointer, or one after a plain observe call on it writes everything.  The buffer ends up bit-identical
 * to what the plain call would have written. */
int sf_observe_device_delta(sf_env *env, float *d_out);
/* The same observation as the list of its non-zero floats, for a consumer on the device that can take it in that form
 * (sf_policy_forward_sparse: the bot network's first convolution works on the non-zeros anyway): nothing dense is
 * written at all.  Per (arena, agent) i: d_counts[i] entries, in the dense buffer's own order (channel, then row, then
 * column), d_keys[i * cap + e] = channel * 9 | row << 9 | column << 14 and d_vals[i * cap + e] the float the dense
 * call would have written there, bit for bit; d_pov[i * 160 ..] = the 5 x 32 floats around the window's centre that
 * AgentModel::forward reads directly (Modules.hpp:114-121: cells (-1,0) (0,-1) (0,0) (0,1) (1,0), channel fastest).
 * d_counts[i] > cap (the list was cut) or == 0xffffffff (a window more crowded than the kernel's record table) means:
 * take the dense call for this step.  An observation of the BASELINE configurations has ~250 non-zeros. */
#define SF_OBS_POV_FLOATS 160
int sf_observe_sparse_device(sf_env *env, uint32_t *d_keys, float *d_vals, uint32_t *d_counts, float *d_pov, int32_t cap);
/* The dense call for exactly the agents whose list did not fit (d_counts[i] > cap, or the 0xffffffff marker), to be
 * issued right behind sf_observe_sparse_device with the same d_counts / cap: their rows of d_dense
 * ([arenas][n_agents][32][31][31], the other rows are not touched) get the observation sf_observe_device would write,
 * and their rows of d_pov are rewritten from it.  Normally no agent qualifies and the two launches exit at once; no
 * host synchronisation either way.  sf_policy_forward_sparse_or_dense then evaluates those agents from d_dense. */
int sf_observe_overflow_device(sf_env *env, const uint32_t *d_counts, int32_t cap, float *d_dense, float *d_pov);

/* Per (arena, agent) 8 x int32: kills, teams_kills, loot, damage, effect, Hp, frames, outcome
 * (gameplay.hpp:461,588-593,625-629; Character.hpp:294).  Latched at episode end. */
int sf_results(sf_env *env, int32_t *out_host);
int sf_results_device(sf_env *env, int32_t *d_out);

/* ---- multi-GPU: the one exchange step ------------------------------------------
 * Arenas are sharded over one sf_env per GPU with no data-path collective.  The result records of all shards are
 * exchanged with an RCCL all-gather over xGMI.  The reference has no counterpart (its only inter-process traffic is the
 * TCP command relay, gameplay.hpp:113-118,170-193); what this replaces is every client simulating every arena.
 * RCCL is loaded on first use; the communicator is the library's own.  Bootstrap: rank 0 calls sf_comm_unique_id and
 * hands the SF_COMM_ID_BYTES bytes to the other ranks by any means (torch.distributed, MPI, a file); every rank then
 * calls sf_comm_init on its env. */
#define SF_COMM_ID_BYTES 128
int sf_comm_unique_id(uint8_t *id);
int sf_comm_init(sf_env *env, const uint8_t *id, int32_t rank, int32_t world);
/* ncclCommCount of the library's communicator: the number of ranks RCCL itself reports (bench.py prints it) */
int sf_comm_ranks(sf_env *env, int32_t *ranks);
/* Snapshot this env's result records on its stream and all-gather the snapshots of all ranks into d_out
 * ([world][arenas][n_agents][8] int32, device memory) on a side stream owned by the library: the call returns at
 * once and the gather runs beside the launches that follow.  d_out is complete after sf_comm_wait. */
int sf_results_allgather(sf_env *env, int32_t *d_out);
/* Make the env's stream (and, with host_too != 0, the calling thread) wait for every gather issued so far. */
int sf_comm_wait(sf_env *env, int32_t host_too);

/* Replaces the loop exit test `if(check_end()) break;` gameplay.hpp:1450.  Without auto_reset: 1 once the arena's
 * episode has ended (the arena then stands still).  With auto_reset: the number of episodes (saturating at 255) that
 * ended during the LAST sf_step / sf_step_device call, over all of its k iterations; the arena has already restarted.
 * The result record (sf_results) is the one of the last episode that ended; a caller that gathers records after
 * multi-step launches tells fresh from stale ones by this count or by sf_arena_hdr.episodes. */
int sf_done(sf_env *env, uint8_t *out_host);
/* The same flag on the device, one byte per (arena, agent) (every agent of an arena gets its arena's flag), in the
 * layout the policy library's reset_memory entry (strikeforce_policy.h) takes: with auto_reset it marks the agents whose game just restarted, i.e. where the
 * reference would have built a new Agent in prepare() (gameplay.hpp:481). */
int sf_done_device(sf_env *env, uint8_t *d_out);
/* The same flags where the environment keeps them: arena a's is the int32 at (*d_words)[a * *stride_words], and each of
 * its *agents_per_arena agents takes it (device memory owned by env, valid until sf_destroy, rewritten by every step) —
 * for a consumer on the same stream that can read them in place (sf_policy_predict_sparse) instead of a copy. */
int sf_done_view_device(sf_env *env, const int32_t **d_words, int32_t *stride_words, int32_t *agents_per_arena);

/* ---- parity / tooling ---------------------------------------------------------------------- */
int sf_state_digest(sf_env *env, uint64_t *out_host); /* one 64-bit digest per arena */
/* Generator draws (random.hpp:54-62 `_rand()`) of every arena's LAST iteration by phase, out[arenas][6]: zombie_action
 * (gameplay.hpp:654-693), the first update_bull (:1059-1100: always 1), human_action (:965-1012: the NPC commands + the
 * sweep direction), the second update_bull (1), the next loop top's spawns (:1444-1449), everything else (0:
 * portal_damage, update_tmp and the hits draw nothing).  The state digest pins the TOTAL (jomle); this pins the order in
 * which the phases draw (SURVEY.md §8c golden item 5).  After sf_step_begin: the first two entries are of the running
 * iteration, the others still of the one before. */
int sf_phase_draws(sf_env *env, int32_t *out_host);
int sf_dump_arena(sf_env *env, int32_t arena, sf_arena_hdr *hdr, sf_human_rec *humans,
                  sf_zombie_rec *zombies, sf_bullet_rec *bullets, sf_portal_rec *portals,
                  uint8_t *cell_flags, int32_t *cell_dmg, int32_t *cell_portal);

/* Stream control: the library launches on this hipStream_t (passed as void*); NULL = default stream. */
int sf_set_stream(sf_env *env, void *hip_stream);
int sf_synchronize(sf_env *env);

/* Device time of the step kernels launched since the last call, measured with HIP events on the
 * library's stream: *ms = sum of kernel durations, *launches = number of launches. */
int sf_kernel_time(sf_env *env, int32_t enable, float *ms, int32_t *launches);

const char *sf_last_error(void);
int sf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STRIKEFORCE_H */
