/*
 * sf_oracle.c — CPU restatement of the StrikeForce per-tick gameplay path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (strikeforce_amd/) never links, imports or calls it.
 *
 * PARITY PIN STATUS: the RNG (o_srand/o_rand) and compute_damage are pinned by known answers recorded
 * from the compiled reference (tests/golden/kat.json).  The tick functions are a
 * line-by-line restatement of the reference sources cited at each function and are otherwise
 * "parity unpinned": the reference has no tests or golden trajectories, and it cannot be
 * built in this image without stand-ins for SFML, which the build rules forbid.
 *
 * Citations: G = StrikeForce-client/gameplay.hpp, CH = Character.hpp, IT = Item.hpp,
 * RN = random.hpp, CU = bots/bot-0.5/Custom.hpp (all under /root/reference/StrikeForce-client).
 *
 * The structure deliberately follows the reference (array-of-structs slots, per-cell node with
 * occupant "pointers" as slot indices, the same loops in the same order) so that it can be audited
 * against the source; the device code in strikeforce_amd/csrc uses a different layout on purpose.
 */
#include "../include/strikeforce.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* node::s bit numbers, G:237-241 / G:321-346 */
enum { S_HUMAN = 0, S_ZOMBIE = 1, S_BULLET = 2, S_WALL = 3, S_CHEST = 4, S_PIN_UP = 5, S_PIN_DN = 6, S_POUT = 7, S_TEMP = 10 };
#define SB(n, k) (((n)->s >> (k)) & 1u)
#define SSET(n, k, v) ((n)->s = (uint16_t)(((n)->s & ~(1u << (k))) | ((unsigned)((v) != 0) << (k))))

static const int wdx[4] = {1, 0, -1, 0}, wdy[4] = {0, 1, 0, -1}; /* CH:47, G:459 */
enum { LIM_PORTAL = 1000, LIM_BLOCK = 1100 };                   /* G:37 */

/* branch-coverage counters (test aid: tests assert that the parity runs really exercise these paths) */
enum { EV_ZOMBIE_PUNCH, EV_ZOMBIE_MOVE, EV_ZOMBIE_BLOCKED, EV_HUMAN_HIT, EV_ZOMBIE_HIT, EV_HUMAN_KILLED, EV_ZOMBIE_KILLED,
       EV_BLOCK_PLACED, EV_BLOCK_BROKEN, EV_EXIT_PLACED, EV_ENTRANCE_PLACED, EV_PORTAL_BROKEN, EV_TELEPORT,
       EV_CHEST_SPAWN, EV_CHEST_CLAIM, EV_ZOMBIE_SPAWN, EV_NPC_SPAWN, EV_PUNCH, EV_SHOT, EV_THROW, EV_SHOT_BLOCKED,
       EV_USE, EV_BULLET_ORPHANED, EV_BULLET_STOPPED, EV_BULLET_EXPIRED, EV_BULLET_ABSORBED, EV_RADIATION,
       EV_NO_BULLET_SLOT, EV_EPISODE_END, EV_HUMAN_MOVE, EV_HUMAN_MOVE_BLOCKED, EV_COUNT };
static const char *const EV_NAMES[EV_COUNT] = {
    "zombie_punch", "zombie_move", "zombie_blocked", "human_hit", "zombie_hit", "human_killed", "zombie_killed",
    "block_placed", "block_broken", "exit_placed", "entrance_placed", "portal_broken", "teleport", "chest_spawn",
    "chest_claim", "zombie_spawn", "npc_spawn", "punch", "shot", "throw", "shot_blocked", "use", "bullet_orphaned",
    "bullet_stopped", "bullet_expired", "bullet_absorbed", "radiation", "no_bullet_slot", "episode_end", "human_move",
    "human_move_blocked"};
#define EV(a, k) (++(a)->ev[k])

typedef struct {
  uint16_t s;
  int dmg, portal_ind;
  int human, zombie, bullet, cons; /* slot indices standing in for the four pointers, G:239-241 */
} onode;

typedef struct {
  int alive, remote, rnpc, prof, ctrl; /* mh, remote (G:55), is_rnpc (CH:291), active_agent (CH:291) */
  int f, r, c, way, team;
  int hp, stamina, mindamage;
  int kills, damage, effect;
  int vec, ind;
  int cons[4], thr_cnt[4];
  int blocks, portals, portal_ind;
} ohuman;

typedef struct { int alive, f, r, c, hp, mindamage, super_; } ozombie;
typedef struct { int alive, f, r, c, way, df, dr, dc, damage, effect, range, owner; } obullet;
typedef struct { int active, f, r, c; } oportal;

/* What Human::build (CH:650-709) leaves in a freshly built human of a given profile. */
typedef struct {
  int hp, mindamage, stamina, mindamage_def;
  int blocks, portals;
  int cons[4], thr_cnt[4];
  int thr[4][4];    /* stamina, damage, effect, range after upgrades (CH:677-678) */
  int weapon[8][4]; /* stamina, damage, effect, range after upgrades (CH:683-684) */
  int weapon_lvl[8];
} oderived;

typedef struct {
  /* per-arena game state */
  onode *map, *map1; /* themap, themap1 G:475 */
  ohuman *hum;
  ozombie *zomb;
  obullet *bull;
  oportal *portal;
  int *temp, ntemp;  /* std::vector<node*> temp G:467, as cell indices */
  int *place;        /* G:469 */
  uint8_t *command;  /* G:43 */
  long long random[18], seed[18], us[18], jomle; /* RN:29-31 */
  long long loot, teams_kills, kills, chest, frame; /* G:461 */
  long long tb, serial, steps, episodes;
  int done, outcome, ended_last_step;
  int32_t results[SF_MAX_AGENTS][8];
  long long draws; /* number of _rand() calls since reset (test aid) */
  long long ev[EV_COUNT];
} oarena;

typedef struct sfo_env {
  sf_config cfg;
  char *map_chars;
  int16_t *map_portal;
  int F, N, M, H, Z, B, P, C, ind;
  oderived der[2];
  oarena *ar;
} sfo_env;

/* ------------------------------------------------------------------------------------------------ */
/* RN:27-77 */
static const long long MOD = (1 << 16) + 1;

static long long o_binpow(long long a, long long b) { /* RN:42-52 */
  long long res = 1;
  b %= MOD - 1;
  while (b) {
    if (b & 1) res = (res * a) % MOD;
    a = (a * a) % MOD;
    b >>= 1;
  }
  return res;
}

static long long o_smallpow(long long x, long long e) { /* the p[x][e] table of RN:33-40: x^e mod 65537, e in 0..10 */
  long long r = 1;
  for (long long j = 0; j < e; ++j) r = (x * r) % MOD;
  return r;
}

static int o_rand(oarena *a) { /* RN:54-62 */
  long long sum = 1;
  for (int i = 0; i < 18; ++i) sum = (sum + a->us[i] * o_smallpow(a->random[i], a->seed[i])) % MOD;
  a->random[0] = o_binpow(sum + (int)(sum == 0), ++a->jomle);
  for (int i = 0; i < 17; ++i) {
    long long t = a->random[i];
    a->random[i] = a->random[i + 1];
    a->random[i + 1] = t;
  }
  ++a->draws;
  return (int)(a->random[17] & 1023);
}

static void o_srand(oarena *a, long long tb, long long u_s) { /* RN:64-76 */
  for (int i = 0; i < 18; ++i) {
    a->us[i] = u_s % 10 + 1;
    a->seed[i] = tb % 10 + 1;
    u_s /= 10;
    tb /= 10;
    a->random[i] = 0;
  }
  a->jomle = 18;
  for (int i = 0; i < 1024; ++i) o_rand(a);
}

/* CH:29-45 (note the reference's indentation: only `tmp /= mid` is inside the for) */
static int o_compute_damage(int x, int y) {
  int l = 0, r = x + 1, z = 2;
  while (1 < y) {
    y >>= 1;
    ++z;
  }
  while (r - l > 1) {
    int mid = (l + r) >> 1, tmp = x;
    for (int i = 0; i < z && mid; ++i) tmp /= mid;
    if (tmp)
      l = mid;
    else
      r = mid;
  }
  return l;
}

/* ------------------------------------------------------------------------------------------------ */
static inline int cell_of(const sfo_env *e, int f, int r, int c) { return (f * e->N + r) * e->M + c; }
static inline int in_map(const sfo_env *e, int r, int c) { return r >= 0 && c >= 0 && r < e->N && c < e->M; }

static const onode ND_WALL = {1u << S_WALL, 0, -1, -1, -1, -1, -1};

/* The reference indexes themap without bounds checks in zombie_action/update_bull/portal_damage
 * (SURVEY App. E-3).  Out-of-range reads see an indestructible wall; no out-of-range write happens
 * because nothing enters such a cell. */
static inline const onode *node_at(const sfo_env *e, const oarena *a, int f, int r, int c) {
  if (!in_map(e, r, c) || f < 0 || f >= e->F) return &ND_WALL;
  return &a->map[cell_of(e, f, r, c)];
}

/* G:321-346.  Returns the same characters; humans map to their facing symbol G:43. */
static char o_showit(const oarena *a, const onode *n) {
  static const char hsym[4] = {'V', '>', 'A', '<'};
  if (SB(n, S_WALL)) return '#';
  if (SB(n, S_HUMAN)) return hsym[a->hum[n->human].way - 1];
  if (SB(n, S_ZOMBIE)) return a->zomb[n->zombie].super_ ? 'Z' : 'z';
  if (SB(n, S_PIN_UP)) return '^';
  if (SB(n, S_PIN_DN)) return 'v';
  if (SB(n, S_BULLET)) return '*';
  if (SB(n, S_CHEST)) return '?';
  /* s[8] 'X' is render-only: set G:575,586,612,623,643 and cleared by updmap G:489-495 before any reader */
  if (SB(n, S_POUT)) return 'O';
  return '.';
}

/* G:209-235 */
static int o_p_ind(const sfo_env *e, const oarena *a) {
  for (int i = 0; i < e->P; ++i)
    if (!a->portal[i].active) return i;
  return -1;
}
static int o_h_ind(const sfo_env *e, const oarena *a) {
  for (int i = 0; i < e->H; ++i)
    if (i != e->ind && !a->hum[i].alive && !a->hum[i].remote) return i;
  return -1;
}
static int o_z_ind(const sfo_env *e, const oarena *a) {
  for (int i = 0; i < e->Z; ++i)
    if (!a->zomb[i].alive) return i;
  return -1;
}
static int o_b_ind(const sfo_env *e, const oarena *a) {
  for (int i = 0; i < e->B; ++i)
    if (!a->bull[i].alive) return i;
  return -1;
}

/* IT:156-163 Bullet::shot */
static void o_shot(obullet *b, int f, int r, int c, int way, int damage, int effect, int range, int owner) {
  b->f = b->df = f;
  b->r = b->dr = r;
  b->c = b->dc = c;
  b->way = way;
  b->damage = damage;
  b->effect = effect;
  b->range = range;
  b->owner = owner;
}

/* ------------------------------------------------------------------------------------------------ */
/* Human::build CH:650-709 on a profile record */
static void o_derive(const sf_config *cfg, const sf_profile *p, oderived *d) {
  int def_blocks = 8, def_portals = 1; /* CH:78-79 */
  int def_hp = p->def_hp, md = p->mindamage_def, def_st = p->def_stamina;
  d->hp = def_hp, d->mindamage = md, d->stamina = def_st; /* CH:667, before the level loops */
  for (int i = 0; i < 4; ++i) d->cons[i] = p->cons[i];
  for (int i = 0; i < 4; ++i) {
    int lvl = p->throw_lvl_cnt[i][0];
    int up = lvl - 1 > 0 ? lvl - 1 : 0; /* for(j = 0; j + 1 < lvl; ++j) upgrade()  CH:674-675 */
    d->thr_cnt[i] = p->throw_lvl_cnt[i][1];
    d->thr[i][0] = cfg->items.thr[i][0];
    d->thr[i][1] = cfg->items.thr[i][1] + 50 * up; /* IT:105-111 */
    d->thr[i][2] = cfg->items.thr[i][2] - 50 * up;
    d->thr[i][3] = cfg->items.thr[i][3];
  }
  for (int i = 0; i < 8; ++i) {
    int lvl = p->weapon_lvl[i] > 0 ? p->weapon_lvl[i] : 0; /* for(j = 0; j < lvl; ++j) upgrade()  CH:680-681 */
    d->weapon_lvl[i] = p->weapon_lvl[i];
    d->weapon[i][0] = cfg->items.weapon[i][0];
    d->weapon[i][1] = cfg->items.weapon[i][1] + 50 * lvl;
    d->weapon[i][2] = cfg->items.weapon[i][2] - 50 * lvl;
    d->weapon[i][3] = cfg->items.weapon[i][3];
  }
  const int lv[3] = {p->level_solo, p->level_timer, p->level_squad}; /* CH:689-706 */
  for (int m = 0; m < 3; ++m)
    for (int level = 2; level <= lv[m]; ++level) { /* level_*_up CH:765-807 */
      md += 5, def_hp += 50, def_st += 50;
      if (level % 2 == 1) ++def_blocks, ++def_portals;
    }
  d->mindamage_def = md;
  d->blocks = def_blocks, d->portals = def_portals; /* back_tmp CH:139-144 */
  (void)def_hp;
  (void)def_st;
}

static void o_make_human(const sfo_env *e, ohuman *h, int prof) {
  const oderived *d = &e->der[prof];
  memset(h, 0, sizeof *h);
  h->prof = prof;
  h->hp = d->hp, h->stamina = d->stamina, h->mindamage = d->mindamage;
  for (int i = 0; i < 4; ++i) h->cons[i] = d->cons[i], h->thr_cnt[i] = d->thr_cnt[i];
  h->blocks = d->blocks, h->portals = d->portals, h->portal_ind = -1;
  h->vec = h->ind = -1; /* CH:83 */
  h->way = 1;
}

/* CH:866-888 gen_human: profile 1 (human_enemy.txt) with (level-1) x 3 level-ups already folded into der[1] */
static void o_gen_human(const sfo_env *e, ohuman *h, int rnpc, int f, int r, int c) {
  o_make_human(e, h, 1);
  h->f = f, h->r = r, h->c = c;
  h->way = 1;
  h->rnpc = rnpc;
  h->team = 0;
  h->kills = h->damage = h->effect = 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* G:507-515 */
static void o_claim_chest(sfo_env *e, oarena *a, ohuman *pl) {
  onode *n = &a->map[cell_of(e, pl->f, pl->r, pl->c)];
  if (SB(n, S_CHEST)) {
    const int32_t *c = e->cfg.items.cons[n->cons]; /* Human::claim_chest CH:372-377 */
    pl->stamina += c[0];
    pl->hp += c[1];
    pl->mindamage += c[2];
    SSET(n, S_CHEST, 0);
    --a->chest;
    EV(a, EV_CHEST_CLAIM);
  }
}

/* G:517-530 */
static void o_teleport(sfo_env *e, oarena *a, int hi) {
  ohuman *pl = &a->hum[hi];
  onode *n = &a->map[cell_of(e, pl->f, pl->r, pl->c)];
  int index = n->portal_ind;
  if (index == -1) return;
  const oportal *p = &a->portal[index];
  onode *t = &a->map[cell_of(e, p->f, p->r, p->c)];
  if (o_showit(a, t) != 'O') return;
  SSET(t, S_HUMAN, 1);
  t->human = hi;
  SSET(n, S_HUMAN, 0);
  pl->f = p->f, pl->r = p->r, pl->c = p->c;
  EV(a, EV_TELEPORT);
}

/* G:532-543 */
static void o_spawn_chest(sfo_env *e, oarena *a) {
  if (e->C <= a->chest) return;
  int i = o_rand(a) % e->F, j = o_rand(a) % e->N, k = o_rand(a) % e->M;
  onode *n = &a->map[cell_of(e, i, j, k)];
  if (o_showit(a, n) != '.') return;
  n->cons = o_rand(a) % 4;
  SSET(n, S_CHEST, 1);
  ++a->chest;
  EV(a, EV_CHEST_SPAWN);
}

/* G:545-558 */
static void o_spawn_zombie_npc(sfo_env *e, oarena *a) {
  int i = o_rand(a) % e->F, j = o_rand(a) % e->N, k = o_rand(a) % e->M;
  onode *n = &a->map[cell_of(e, i, j, k)];
  if (o_showit(a, n) != '.') return;
  int index = o_z_ind(e, a);
  if (index == -1) return;
  int super_ = (o_rand(a) % 4 == 0);
  ozombie *z = &a->zomb[index]; /* gen_zombie CH:866-871, Zombie::gen_npc CH:850-857 */
  z->f = i, z->r = j, z->c = k;
  z->super_ = super_;
  z->mindamage = (super_ + 1) * 100;
  z->hp = (super_ + 1) * 400;
  n->zombie = index;
  SSET(n, S_ZOMBIE, 1);
  z->alive = 1;
  EV(a, EV_ZOMBIE_SPAWN);
}

/* G:560-572 */
static void o_spawn_human_npc(sfo_env *e, oarena *a) {
  int i = o_rand(a) % e->F, j = o_rand(a) % e->N, k = o_rand(a) % e->M;
  onode *n = &a->map[cell_of(e, i, j, k)];
  if (o_showit(a, n) != '.') return;
  int index = o_h_ind(e, a);
  if (index == -1) return;
  o_gen_human(e, &a->hum[index], 1, i, j, k);
  n->human = index;
  SSET(n, S_HUMAN, 1);
  a->hum[index].remote = 0;
  a->hum[index].alive = 1;
  EV(a, EV_NPC_SPAWN);
}

/* G:574-598 */
static void o_zombie_damage(sfo_env *e, oarena *a, onode *pix) {
  obullet *b = &a->bull[pix->bullet];
  ozombie *z = &a->zomb[pix->zombie];
  z->hp -= b->damage; /* Character::hit CH:242-246 */
  z->mindamage += b->effect;
  SSET(pix, S_BULLET, 0);
  ohuman *owner = b->owner ? &a->hum[b->owner - 1] : NULL;
  if (owner) {
    owner->damage += b->damage;
    owner->effect += b->effect;
  }
  b->alive = 0;
  EV(a, EV_ZOMBIE_HIT);
  if (z->hp <= 0) {
    z->alive = 0;
    EV(a, EV_ZOMBIE_KILLED);
    SSET(pix, S_ZOMBIE, 0);
    if (owner && owner->team == a->hum[e->ind].team) {
      int pts = 500 + 250 * z->super_;
      ++a->teams_kills, a->loot += pts / 10;
      if (owner == &a->hum[e->ind]) a->loot += pts * 9 / 10, ++a->kills;
    }
    if (owner) ++owner->kills;
  }
}

/* G:600-609 */
static void o_hit_zombie(sfo_env *e, oarena *a) {
  for (int i = 0; i < e->Z; ++i)
    if (a->zomb[i].alive) {
      onode *pix = &a->map[cell_of(e, a->zomb[i].f, a->zomb[i].r, a->zomb[i].c)];
      if (SB(pix, S_BULLET)) o_zombie_damage(e, a, pix);
    }
}

/* G:611-634 */
static void o_human_damage(sfo_env *e, oarena *a, onode *pix) {
  obullet *b = &a->bull[pix->bullet];
  ohuman *h = &a->hum[pix->human];
  h->hp -= b->damage;
  h->mindamage += b->effect;
  SSET(pix, S_BULLET, 0);
  ohuman *owner = b->owner ? &a->hum[b->owner - 1] : NULL;
  ohuman *me = &a->hum[e->ind];
  if (owner && h->team != owner->team) {
    owner->damage += b->damage;
    owner->effect += b->effect;
  }
  b->alive = 0;
  EV(a, EV_HUMAN_HIT);
  if (h->hp <= 0) {
    h->alive = 0;
    EV(a, EV_HUMAN_KILLED);
    SSET(pix, S_HUMAN, me == h);
    if (owner && owner->team == me->team && h->team != me->team) {
      ++a->teams_kills, a->loot += 100;
      if (owner == me) a->loot += 900, ++a->kills;
    }
    if (owner && h->team != owner->team) ++owner->kills;
  }
}

/* G:636-652 */
static void o_hit_human(sfo_env *e, oarena *a) {
  for (int i = 0; i < e->H; ++i)
    if (a->hum[i].alive) {
      onode *pix = &a->map[cell_of(e, a->hum[i].f, a->hum[i].r, a->hum[i].c)];
      if (a->hum[i].hp <= 0) {
        a->hum[i].alive = 0;
        SSET(pix, S_HUMAN, pix->human == e->ind);
      } else if (SB(pix, S_BULLET))
        o_human_damage(e, a, pix);
      if (a->hum[i].hp <= 0 && i != e->ind) a->hum[i].ctrl = 0;
    }
}

/* G:654-693 */
static void o_zombie_action(sfo_env *e, oarena *a) {
  for (int _ = 0; _ < e->Z; ++_)
    if (a->zomb[_].alive) {
      ozombie *z = &a->zomb[_];
      int i = z->f, j = z->r, k = z->c;
      onode *own = &a->map[cell_of(e, i, j, k)];
      if (SB(own, S_BULLET)) continue;
      int b = 0;
      for (int i1 = 0; i1 < 4; ++i1) {
        const onode *cn = node_at(e, a, i, wdx[i1] + j, wdy[i1] + k);
        if (SB(cn, S_HUMAN)) {
          onode *pix = (onode *)cn; /* a cell with a human is always inside the map */
          int index = o_b_ind(e, a);
          if (!SB(pix, S_BULLET) && index != -1) {
            /* Zombie::punch CH:838-844 */
            int dmg = z->mindamage > 0 ? z->mindamage : 0;
            o_shot(&a->bull[index], i, j + wdx[i1], k + wdy[i1], i1 + 1, dmg, 0, 1, 0);
            pix->bullet = index;
            SSET(pix, S_BULLET, 1);
            a->bull[index].alive = 1;
            EV(a, EV_ZOMBIE_PUNCH);
          } else if (index == -1)
            EV(a, EV_NO_BULLET_SLOT);
          b = 1;
        }
      }
      if (!b) {
        if (o_rand(a) % 5 < 2) continue;
        for (int i1 = 0; i1 < 2; ++i1) {
          int i2 = o_rand(a) % 4;
          const onode *cn = node_at(e, a, i, wdx[i2] + j, wdy[i2] + k);
          if (o_showit(a, cn) == '.') {
            onode *t = (onode *)cn;
            SSET(t, S_ZOMBIE, 1);
            t->zombie = _;
            SSET(own, S_ZOMBIE, 0);
            z->r = wdx[i2] + j, z->c = wdy[i2] + k;
            EV(a, EV_ZOMBIE_MOVE);
            break;
          } else
            EV(a, EV_ZOMBIE_BLOCKED);
        }
      }
    }
}

/* Human::punch / shot_it / throw_it, CH:391-427 */
static int o_punch(sfo_env *e, oarena *a, int hi, obullet *b) {
  ohuman *h = &a->hum[hi];
  int cd = o_compute_damage(e->der[h->prof].mindamage_def, 1);
  int dmg = cd > h->mindamage ? cd : h->mindamage;
  o_shot(b, h->f, h->r + wdx[h->way - 1], h->c + wdy[h->way - 1], h->way, dmg, 0, 1, hi + 1);
  return 1;
}
static int o_shot_it(sfo_env *e, oarena *a, int hi, obullet *b) {
  ohuman *h = &a->hum[hi];
  const int *w = e->der[h->prof].weapon[h->ind];
  if (h->stamina + w[0] < 0) return 0;
  h->stamina += w[0];
  int cd = o_compute_damage(w[1], w[3]);
  int dmg = cd > w[1] + h->mindamage ? cd : w[1] + h->mindamage;
  o_shot(b, h->f, h->r + wdx[h->way - 1], h->c + wdy[h->way - 1], h->way, dmg, w[2], w[3], hi + 1);
  return 1;
}
static int o_throw_it(sfo_env *e, oarena *a, int hi, obullet *b) {
  ohuman *h = &a->hum[hi];
  const int *t = e->der[h->prof].thr[h->ind];
  int dmg = t[1] > t[1] + h->mindamage ? t[1] : t[1] + h->mindamage;
  if (h->stamina + t[0] < 0) return 0;
  if (h->thr_cnt[h->ind] < 1) {
    h->vec = -1;
    return 0;
  }
  h->stamina += t[0];
  --h->thr_cnt[h->ind];
  if (h->thr_cnt[h->ind] < 1) h->vec = -1;
  o_shot(b, h->f, h->r + wdx[h->way - 1], h->c + wdy[h->way - 1], h->way, dmg, t[2], t[3], hi + 1);
  return 1;
}

/* Human::use CH:379-389 */
static void o_use(sfo_env *e, ohuman *h) {
  if (h->vec != 0 || h->cons[h->ind] < 1) return; /* vec != 0 short-circuits before ind is used (App. E-8) */
  const int32_t *c = e->cfg.items.cons[h->ind];
  h->stamina += c[0];
  h->hp += c[1];
  h->mindamage += c[2];
  if (--h->cons[h->ind] < 1) h->vec = -1;
}

static int idx_in(const char *set, int n, char c) {
  for (int i = 0; i < n; ++i)
    if (set[i] == c) return i;
  return -1;
}

/* G:695-821 */
static void o_obey(sfo_env *e, oarena *a, char c, int hi) {
  ohuman *pl = &a->hum[hi];
  if (c == '_') {
    pl->hp = 0;
    return;
  }
  if (c == '[' || c == ']') {
    int d = pl->way - 1;
    int r = pl->r + wdx[d], cc = pl->c + wdy[d];
    if (r >= e->N || 0 > r || cc >= e->M || 0 > cc) return;
    int ci = cell_of(e, pl->f, r, cc);
    onode *n = &a->map[ci];
    if (o_showit(a, n) != '.') return;
    if (c == '[') {
      if (pl->blocks) {
        SSET(n, S_TEMP, 1);
        SSET(n, S_WALL, 1);
        --pl->blocks;
        a->temp[a->ntemp++] = ci;
        EV(a, EV_BLOCK_PLACED);
      }
      return;
    } else {
      if (~pl->portal_ind) {
        SSET(n, S_TEMP, 1);
        SSET(n, S_PIN_UP, 1);
        n->portal_ind = pl->portal_ind;
        pl->portal_ind = -1;
        a->temp[a->ntemp++] = ci;
        EV(a, EV_ENTRANCE_PLACED);
      } else if (pl->portals) {
        int index = o_p_ind(e, a);
        if (index == -1) return;
        SSET(n, S_TEMP, 1);
        SSET(n, S_POUT, 1);
        --pl->portals;
        pl->portal_ind = index;
        a->portal[index].f = pl->f, a->portal[index].r = r, a->portal[index].c = cc;
        a->portal[index].active = 1;
        a->temp[a->ntemp++] = ci;
        EV(a, EV_EXIT_PLACED);
      }
      return;
    }
  }
  if (c == 'q' || c == 'e') {
    if (c == 'e') /* turn_r CH:745-751 */
      pl->way = pl->way == 1 ? 4 : pl->way - 1;
    else /* turn_l CH:753-759 */
      pl->way = pl->way == 4 ? 1 : pl->way + 1;
    return;
  }
  int i;
  if ((i = idx_in("sdwa", 4, c)) >= 0) {
    int r = pl->r + wdx[i], cc = pl->c + wdy[i];
    if (r >= e->N || 0 > r || cc >= e->M || 0 > cc) return;
    onode *t = &a->map[cell_of(e, pl->f, r, cc)];
    char sit = o_showit(a, t);
    if (sit == '?' || sit == '^' || sit == 'v' || sit == '.' || sit == 'X' || sit == '*') {
      SSET(t, S_HUMAN, 1);
      t->human = hi;
      SSET(&a->map[cell_of(e, pl->f, pl->r, pl->c)], S_HUMAN, 0);
      pl->r = r, pl->c = cc;
      EV(a, EV_HUMAN_MOVE);
    } else
      EV(a, EV_HUMAN_MOVE_BLOCKED);
    return;
  }
  if ((i = idx_in("fghj", 4, c)) >= 0) {
    if (!pl->cons[i]) return;
    pl->vec = 0, pl->ind = i;
    return;
  }
  if ((i = idx_in("kl;'", 4, c)) >= 0) {
    if (!pl->thr_cnt[i]) return;
    pl->vec = 1, pl->ind = i;
    return;
  }
  if ((i = idx_in("cvbnm,./", 8, c)) >= 0) {
    if (!e->der[pl->prof].weapon_lvl[i]) return;
    pl->vec = 2, pl->ind = i;
    return;
  }
  if (c == 'u') {
    if (pl->vec == 0 && pl->cons[pl->ind] >= 1) EV(a, EV_USE);
    o_use(e, pl);
    return;
  }
  if (c == 'z' || c == 'x') {
    int bway = pl->way - 1;
    int r = pl->r + wdx[bway], cc = pl->c + wdy[bway];
    int index = o_b_ind(e, a);
    if (index == -1) EV(a, EV_NO_BULLET_SLOT);
    if (index == -1 || r >= e->N || 0 > r || cc >= e->M || 0 > cc) return;
    int can;
    const int kind = c == 'z' ? EV_PUNCH : (pl->vec == 1 ? EV_THROW : EV_SHOT);
    if (c == 'z')
      can = o_punch(e, a, hi, &a->bull[index]);
    else if (pl->vec == 1)
      can = o_throw_it(e, a, hi, &a->bull[index]);
    else if (pl->vec == 2)
      can = o_shot_it(e, a, hi, &a->bull[index]);
    else
      return;
    onode *t = &a->map[cell_of(e, pl->f, r, cc)];
    char sit = o_showit(a, t);
    if (can && ((sit != '#' && sit != 'v' && sit != '^') || SB(t, S_TEMP))) {
      if (SB(t, S_BULLET)) EV(a, EV_BULLET_ORPHANED);
      t->bullet = index;
      SSET(t, S_BULLET, 1);
      a->bull[index].alive = 1;
      EV(a, kind);
    } else if (can)
      EV(a, EV_SHOT_BLOCKED);
    return;
  }
}

/* G:1927-1940 */
static char o_human_rnpc_bot(oarena *a) {
  if (a->frame % 50 <= 1) {
    static const char c[8] = {'c', 'v', 'b', 'n', 'm', ',', '.', '/'};
    return c[o_rand(a) % 8];
  } else if (o_rand(a) % 5 < 3)
    return 'x';
  else if (o_rand(a) % 5 < 3) {
    static const char c[7] = {'1', '2', 'a', 'w', 's', 'd', 'p'};
    return c[o_rand(a) % 7];
  }
  static const char c[8] = {'+', 'u', 'f', 'g', 'h', 'j', '[', ']'};
  return c[o_rand(a) % 8];
}

/* G:965-1012.  ext = this step's external commands, one per configured agent slot. */
static void o_human_action(sfo_env *e, oarena *a, const uint8_t *ext) {
  a->command[e->ind] = ext[e->ind]; /* get_my_action G:939-963: command[ind] = bot(hum[ind]) */
  for (int i = 0; i < e->H; ++i)
    if (i != e->ind && a->hum[i].alive) {
      if (a->hum[i].remote) {
        a->command[i] = ext[i]; /* client.recieve() G:170-193 */
      } else {                  /* get_command G:929-937 */
        if (a->hum[i].rnpc)
          a->command[i] = (uint8_t)o_human_rnpc_bot(a);
        else if (a->hum[i].ctrl)
          a->command[i] = ext[i]; /* bot(hum[i]) with an active agent */
        else
          a->command[i] = '+'; /* bot() without an agent: CU:138-139 */
      }
    }
  int r = o_rand(a) & 1, st = (1 - r) * (e->H - 1), dif = 2 * r - 1;
  for (int i = st; i < e->H && (~i); i += dif)
    if (a->hum[i].alive) {
      o_obey(e, a, (char)a->command[i], i);
      o_teleport(e, a, i);
      o_claim_chest(e, a, &a->hum[i]);
      a->command[i] = '+';
    }
}

/* G:1059-1100 */
static void o_update_bull(sfo_env *e, oarena *a) {
  int cnt = 0;
  for (int _ = 0; _ < e->B; ++_)
    if (a->bull[_].alive) {
      obullet *b = &a->bull[_];
      int i = b->f, j = b->r, k = b->c, d = b->way - 1;
      int ci = cell_of(e, i, j, k);
      a->map1[ci].s = a->map[ci].s;
      SSET(&a->map1[ci], S_BULLET, 0);
      a->place[cnt++] = ci;
      if (in_map(e, j + wdx[d], k + wdy[d])) { /* App. E-3: the reference does not check */
        int di = cell_of(e, i, j + wdx[d], k + wdy[d]);
        a->map1[di].s = a->map[di].s;
        SSET(&a->map1[di], S_BULLET, 0);
        a->place[cnt++] = di;
      }
    }
  int r = o_rand(a) & 1, st = (1 - r) * (e->B - 1), dif = 2 * r - 1;
  /* `if(r) reverse(place, place + cnt)` G:1075: the write-back below is order independent */
  for (int _ = st; _ < e->B && (~_); _ += dif)
    if (a->bull[_].alive) {
      obullet *b = &a->bull[_];
      int i = b->f, j = b->r, k = b->c;
      int dist = abs(b->f - b->df) + abs(b->r - b->dr) + abs(b->c - b->dc); /* Bullet::expire IT:165-168 */
      if (dist + 1 >= b->range) {
        b->alive = 0;
        EV(a, EV_BULLET_EXPIRED);
        continue;
      }
      int d = b->way - 1;
      const onode *dn = node_at(e, a, i, j + wdx[d], k + wdy[d]);
      char sit = o_showit(a, dn);
      if ((sit != '#' && sit != 'v' && sit != '^') || SB(dn, S_TEMP)) {
        onode *d1 = &a->map1[cell_of(e, i, j + wdx[d], k + wdy[d])];
        if (SB(d1, S_BULLET)) EV(a, EV_BULLET_ORPHANED);
        d1->bullet = _;
        b->r = j + wdx[d], b->c = k + wdy[d];
        SSET(d1, S_BULLET, 1);
      } else {
        b->alive = 0;
        EV(a, EV_BULLET_STOPPED);
      }
    }
  for (int _ = 0; _ < cnt; ++_) {
    int ci = a->place[_];
    SSET(&a->map[ci], S_BULLET, SB(&a->map1[ci], S_BULLET));
    a->map[ci].bullet = a->map1[ci].bullet;
  }
}

/* G:1279-1297 */
static void o_portal_damage(sfo_env *e, oarena *a) {
  for (int i = 0; i < e->P; ++i) {
    if (!a->portal[i].active) continue;
    const oportal *p = &a->portal[i];
    onode *n = &a->map[cell_of(e, p->f, p->r, p->c)];
    if (o_showit(a, n) != 'O') {
      int index = o_b_ind(e, a);
      if (index == -1) return;
      o_shot(&a->bull[index], p->f, p->r, p->c, 3, 20, -10, 1, 0);
      if (SB(n, S_BULLET)) EV(a, EV_BULLET_ORPHANED);
      EV(a, EV_RADIATION);
      n->bullet = index;
      SSET(n, S_BULLET, 1);
      a->bull[index].alive = 1;
    }
  }
}

/* G:1343-1381 */
static void o_update_tmp(sfo_env *e, oarena *a) {
  for (int _ = 0; _ < e->B; ++_) {
    if (!a->bull[_].alive) continue;
    obullet *b = &a->bull[_];
    onode *n = &a->map[cell_of(e, b->f, b->r, b->c)];
    char sit = o_showit(a, n);
    if ((sit == '^' || sit == '#') && SB(n, S_TEMP)) {
      n->dmg += b->damage;
      SSET(n, S_BULLET, 0);
      b->alive = 0;
      EV(a, EV_BULLET_ABSORBED);
    }
  }
  for (int t = 0; t < a->ntemp; ++t) {
    onode *en = &a->map[a->temp[t]];
    char c = o_showit(a, en);
    int dmg = en->dmg;
    if (c == '^' && dmg >= LIM_PORTAL) {
      int i = en->portal_ind;
      onode *e1 = &a->map[cell_of(e, a->portal[i].f, a->portal[i].r, a->portal[i].c)];
      SSET(e1, S_POUT, 0);
      SSET(e1, S_TEMP, 0);
      SSET(en, S_PIN_UP, 0);
      SSET(en, S_TEMP, 0);
      en->portal_ind = -1;
      en->dmg = 0;
      a->portal[i].active = 0;
      EV(a, EV_PORTAL_BROKEN);
    } else if (c == '#' && dmg >= LIM_BLOCK) {
      SSET(en, S_WALL, 0);
      SSET(en, S_TEMP, 0);
      en->dmg = 0;
      EV(a, EV_BLOCK_BROKEN);
    }
  }
  for (int i = 0; i < a->ntemp; ++i)
    if (!SB(&a->map[a->temp[i]], S_TEMP)) {
      int t = a->temp[i];
      a->temp[i] = a->temp[a->ntemp - 1];
      a->temp[a->ntemp - 1] = t;
      --a->ntemp;
      --i;
    }
}

/* G:497-505 */
static int o_rivals_are_dead(sfo_env *e, oarena *a) {
  for (int i = 0; i < e->H; ++i)
    if (a->hum[i].alive) {
      int team = a->hum[i].team;
      if (team && team != a->hum[e->ind].team) return 0;
    }
  return 1;
}

/* G:1102-1229, logic only.  Returns an SF_* outcome (SF_RUNNING = keep playing). */
static int o_check_end(sfo_env *e, oarena *a) {
  const int mode = e->cfg.mode, level = e->cfg.level;
  if (mode == SF_MODE_BATTLE && o_rivals_are_dead(e, a)) return SF_WON;
  if (a->hum[e->ind].hp <= 0) return SF_DIED;
  if (mode == SF_MODE_TIMER) {
    /* G:1145-1146 `time(0) - tb >= level * 60 * 5` on wall-clock; replaced by the frame clock
     * (40 ms per frame G:1942 -> 7500 frames per level), a documented deviation. */
    long long lim = (long long)level * (e->cfg.timer_frames_per_level > 0 ? e->cfg.timer_frames_per_level : 7500);
    if (a->frame - 1 >= lim) return a->kills < level * 5 ? SF_TIME_LOST : SF_TIME_WON;
    return SF_RUNNING;
  }
  if (level * 5 <= a->kills && mode == SF_MODE_SOLO) return SF_WON;
  if (level * 10 <= a->teams_kills && o_rivals_are_dead(e, a) && mode == SF_MODE_SQUAD) return SF_WON;
  return SF_RUNNING;
}

/* ------------------------------------------------------------------------------------------------ */
static void o_place_human(sfo_env *e, oarena *a, int i, int f, int r, int c) {
  onode *n = &a->map[cell_of(e, f, r, c)];
  n->human = i;
  SSET(n, S_HUMAN, 1);
  a->hum[i].f = f, a->hum[i].r = r, a->hum[i].c = c;
}

/* setup() G:1231-1277 + load_data() G:1741-1925 */
static void o_setup(sfo_env *e, oarena *a, long long tb, long long serial) {
  const int cells = e->F * e->N * e->M;
  a->tb = tb, a->serial = serial;
  a->loot = a->teams_kills = a->kills = a->frame = 0;
  a->chest = 0; /* the reference never resets `chest` (App. E-2); first-game semantics */
  a->ntemp = 0;
  a->steps = 0;
  a->done = 0, a->outcome = SF_RUNNING;
  a->draws = 0;
  for (int i = 0; i < e->B; ++i) a->bull[i].alive = 0;
  for (int i = 0; i < e->P; ++i) a->portal[i].active = 0;
  for (int i = 0; i < e->Z; ++i) a->zomb[i].alive = 0;
  for (int i = 0; i < e->H; ++i) {
    memset(&a->hum[i], 0, sizeof(ohuman));
    a->command[i] = '+';
  }
  for (int ci = 0; ci < cells; ++ci) {
    onode *n = &a->map[ci];
    n->s = 0, n->dmg = 0, n->portal_ind = -1, n->human = n->zombie = n->bullet = n->cons = -1;
    a->map1[ci] = *n;
    char c = e->map_chars[ci];
    if (c == '#')
      SSET(n, S_WALL, 1);
    else if (c == '^') {
      SSET(n, S_PIN_UP, 1);
      n->portal_ind = e->map_portal[ci];
    } else if (c == 'v') {
      SSET(n, S_PIN_DN, 1);
      n->portal_ind = e->map_portal[ci];
    } else if (c == 'O') {
      SSET(n, S_POUT, 1);
      int index = o_p_ind(e, a);
      if (index >= 0) {
        a->portal[index].f = ci / (e->N * e->M);
        a->portal[index].r = (ci / e->M) % e->N;
        a->portal[index].c = ci % e->M;
        a->portal[index].active = 1;
      }
    }
  }
  /* load_data: G:1745-1747 derive `serial` from libc rand(); here (tb, serial) are explicit inputs */
  o_srand(a, tb, serial);
  const int mode = e->cfg.mode;
  if (mode == SF_MODE_BATTLE) { /* G:1846-1859 */
    int players = e->cfg.n_agents;
    for (int i = 0; i < players; ++i) { /* Client::start / get_info G:104-151 */
      o_make_human(e, &a->hum[i], 0);
      a->hum[i].team = e->cfg.agent_team[i];
      a->hum[i].alive = 1;
      a->hum[i].remote = (i != e->ind);
      a->hum[i].ctrl = 1;
    }
    for (int i = 0; i < players; ++i) {
      a->hum[i].way = o_rand(a) % 4 + 1;
      while (1) {
        int f = o_rand(a) % e->F, r = o_rand(a) % e->N, c = o_rand(a) % e->M;
        if (o_showit(a, &a->map[cell_of(e, f, r, c)]) == '.') {
          o_place_human(e, a, i, f, r, c);
          break;
        }
      }
    }
  } else if (mode == SF_MODE_SQUAD) { /* G:1861-1903 */
    const int of = e->F > 2 ? 2 : e->F - 1; /* opponents start on floor index 2; clamped for maps with fewer floors */
    o_make_human(e, &a->hum[0], 0);
    a->hum[0].alive = 1, a->hum[0].ctrl = 1;
    o_place_human(e, a, 0, 0, 3, 1);
    a->hum[0].way = 1, a->hum[0].team = 1;
    for (int i = 1; i < 5; ++i) {
      o_gen_human(e, &a->hum[i], 0, 0, 1, i + 1);
      a->hum[i].alive = 1;
      o_place_human(e, a, i, 0, 1, i + 1);
      a->hum[i].team = 1;
      a->hum[i].ctrl = i < e->cfg.n_agents; /* USE_AGENT_IN_SQUAD_NPCS G:1883-1885 */
    }
    for (int i = 5; i < 10; ++i) {
      o_gen_human(e, &a->hum[i], 0, of, 1, i + 1);
      a->hum[i].alive = 1;
      o_place_human(e, a, i, of, 1, i + 1);
      a->hum[i].team = 2;
      a->hum[i].ctrl = i < e->cfg.n_agents;
    }
  } else { /* Solo / Timer G:1905-1920 */
    o_make_human(e, &a->hum[0], 0);
    a->hum[0].alive = 1, a->hum[0].ctrl = 1;
    o_place_human(e, a, 0, 0, 1, 1);
    a->hum[0].way = 1, a->hum[0].team = 1;
  }
}

static void o_latch_results(sfo_env *e, oarena *a) {
  for (int g = 0; g < e->cfg.n_agents; ++g) {
    const ohuman *h = &a->hum[g];
    int32_t *r = a->results[g];
    r[0] = h->kills, r[1] = (int32_t)a->teams_kills, r[2] = (int32_t)a->loot;
    r[3] = h->damage, r[4] = h->effect, r[5] = h->hp, r[6] = (int32_t)a->frame, r[7] = a->outcome;
  }
}

/* The top of play()'s while(true): G:1444-1450 */
static void o_loop_top(sfo_env *e, oarena *a) {
  if (a->frame % 30 <= 1) o_spawn_chest(e, a);
  if (a->frame % 40 <= 1) o_spawn_zombie_npc(e, a);
  if (a->frame % 50 <= 1) o_spawn_human_npc(e, a);
  int out = o_check_end(e, a);
  if (out != SF_RUNNING) {
    a->done = 1;
    a->outcome = out;
    o_latch_results(e, a);
  }
}

static void o_reset_arena(sfo_env *e, oarena *a, long long tb, long long serial) {
  o_setup(e, a, tb, serial);
  ++a->frame; /* G:1441 */
  o_loop_top(e, a);
}

/* One iteration of the loop body G:1452-1471 followed by the next loop top. */
static void o_step_arena(sfo_env *e, oarena *a, const uint8_t *ext) {
  a->ended_last_step = 0;
  if (a->done) return;
  o_zombie_action(e, a);
  o_portal_damage(e, a);
  o_update_tmp(e, a);
  o_hit_human(e, a), o_hit_zombie(e, a);
  ++a->frame; /* updmap G:489-495 only clears render bits */
  o_update_bull(e, a);
  o_human_action(e, a, ext);
  o_update_tmp(e, a);
  o_hit_human(e, a), o_hit_zombie(e, a);
  ++a->frame;
  o_update_bull(e, a);
  ++a->steps;
  o_loop_top(e, a);
  if (a->done) {
    a->ended_last_step = 1;
    ++a->episodes;
    EV(a, EV_EPISODE_END);
    if (e->cfg.auto_reset)
      o_reset_arena(e, a, a->tb + (e->cfg.reseed_stride > 0 ? e->cfg.reseed_stride : e->cfg.arenas), a->serial);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* describe() CU:29-135.  `cell` may be NULL for the out-of-map zero node `nd` (CU:146-147). */
static void o_describe(const sfo_env *e, const oarena *a, const onode *cell, const ohuman *player, float *res) {
  static const onode ND = {0, 0, -1, -1, -1, -1, -1};
  if (!cell) cell = &ND;
  int n = 0;
  const int s0 = SB(cell, S_HUMAN), s1 = SB(cell, S_ZOMBIE), s2 = SB(cell, S_BULLET), s3 = SB(cell, S_WALL),
            s4 = SB(cell, S_CHEST), s5 = SB(cell, S_PIN_UP), s6 = SB(cell, S_PIN_DN), s7 = SB(cell, S_POUT),
            s10 = SB(cell, S_TEMP);
  const ohuman *hm = s0 ? &a->hum[cell->human] : NULL;
  const ozombie *zm = s1 ? &a->zomb[cell->zombie] : NULL;
  res[n++] = (float)(s0 || s1);
  res[n++] = (float)s2, res[n++] = (float)s3, res[n++] = (float)s4;
  res[n++] = (float)(s5 || s6);
  res[n++] = (float)s7, res[n++] = (float)s10;
  float sit[4] = {0, 0, 0, 0};
  if (s0) {
    int t = hm->team;
    if (!t)
      sit[2] = 1;
    else if (t == player->team)
      sit[0] = 1;
    else
      sit[1] = 1;
  }
  if (s1) sit[3] = 1;
  for (int i = 0; i < 4; ++i) res[n++] = sit[i];
  if (s0) {
    res[n++] = (float)hm->kills;
    res[n++] = (float)hm->blocks;
    res[n++] = (float)hm->portals;
    res[n++] = (float)(hm->portal_ind != -1);
  } else
    for (int i = 0; i < 4; ++i) res[n++] = 0;
  sit[0] = sit[1] = sit[2] = 0;
  float hp = 0;
  if (s3 || s5 || s6 || s0 || s1) {
    sit[0] = sit[1] = 1;
    sit[2] = (float)(s10 || s0 || s1);
    if (s0)
      hp = (float)(hm->hp / 1000.0);
    else if (s1)
      hp = (float)(zm->hp / 1000.0);
    else if (s10) {
      if (s3)
        hp = (float)((LIM_BLOCK - cell->dmg) / 1000.0);
      else
        hp = (float)((LIM_PORTAL - cell->dmg) / 1000.0);
    }
  } else if (s7) {
    sit[0] = 1;
    sit[1] = sit[2] = 0;
  }
  for (int i = 0; i < 3; ++i) res[n++] = sit[i];
  res[n++] = hp;
  sit[0] = sit[1] = sit[2] = sit[3] = 0;
  float damage = 0, effect = 0, is_bull = 0, estamina = 0;
  if (s0) {
    sit[hm->way - 1] = 1;
    /* Human::get_damage_effect CH:429-443 */
    const oderived *d = &e->der[hm->prof];
    int v0, v1 = 0;
    int dmg = o_compute_damage(d->mindamage_def, 1);
    if (hm->mindamage > dmg) dmg = hm->mindamage;
    v0 = dmg;
    int decided = 0;
    if (hm->vec == 1) {
      const int *b = d->thr[hm->ind];
      if (0 <= hm->stamina + b[0]) {
        int m = b[1];
        if (b[1] + hm->mindamage > m) m = b[1] + hm->mindamage;
        if (dmg > m) m = dmg;
        v0 = m, v1 = b[2], decided = 1;
      }
    }
    if (!decided && hm->vec == 2) {
      const int *w = d->weapon[hm->ind];
      if (0 <= hm->stamina + w[0]) {
        int m = o_compute_damage(w[1], w[3]);
        if (w[1] + hm->mindamage > m) m = w[1] + hm->mindamage;
        if (dmg > m) m = dmg;
        v0 = m, v1 = w[2];
      }
    }
    damage = (float)(v0 / 1000.0);
    effect = (float)(-v1 / 1000.0);
    estamina = (float)(hm->stamina / 1000.0);
  } else if (s1) {
    sit[0] = sit[1] = sit[2] = sit[3] = (float)0.01;
    damage = (float)(zm->mindamage / 1000.0);
  } else if (s2) {
    const obullet *b = &a->bull[cell->bullet];
    is_bull = 1;
    int dist_traveled = abs(b->r - b->dr) + abs(b->c - b->dc);
    sit[b->way - 1] = (float)((b->range - dist_traveled) / 100.0);
    damage = (float)(b->damage / 1000.0);
    effect = (float)(-b->effect / 1000.0);
  } else if (s7) {
    damage = (float)(20 / 1000.0);
    effect = (float)(10 / 1000.0);
  }
  res[n++] = is_bull;
  for (int i = 0; i < 4; ++i) res[n++] = sit[i];
  res[n++] = damage, res[n++] = effect, res[n++] = estamina;
  sit[0] = sit[1] = sit[2] = 0;
  if (s4) {
    const int32_t *c = e->cfg.items.cons[cell->cons];
    sit[0] = (float)(c[0] / 1000.0);
    sit[1] = (float)(c[2] / 1000.0);
    sit[2] = (float)(c[1] / 1000.0);
  }
  for (int i = 0; i < 3; ++i) res[n++] = sit[i];
  if (s0) {
    res[n++] = (float)(hm->damage / 1000.0);
    res[n++] = (float)(-hm->effect / 1000.0);
  } else {
    res[n++] = 0;
    res[n++] = 0;
  }
}

/* gameplay::bot CU:137-159 (observation part) */
static void o_observe_agent(const sfo_env *e, const oarena *a, int hi, float *obs) {
  const ohuman *pl = &a->hum[hi];
  const int W = SF_OBS_WINDOW, r = W / 2;
  float vec[SF_OBS_CHANNELS];
  for (int i = pl->r - r, ii = 0; i <= pl->r + r; ++i, ++ii)
    for (int j = pl->c - r, jj = 0; j <= pl->c + r; ++j, ++jj) {
      if (i < 0 || j < 0 || e->N <= i || e->M <= j)
        o_describe(e, a, NULL, pl, vec);
      else
        o_describe(e, a, &a->map[cell_of(e, pl->f, i, j)], pl, vec);
      for (int k = 0; k < SF_OBS_CHANNELS; ++k) {
        float x = fabsf(vec[k]) / 10; /* float / int -> float, CU:157 */
        obs[k * W * W + ii * W + jj] = (float)pow((double)x, 0.2);
      }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Canonical state dump + digest (shared definition with the device library). */
static inline uint64_t mix64(uint64_t x) { /* splitmix64 finalizer */
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
static inline uint64_t dg(uint64_t tag, uint64_t idx, int64_t v) {
  return mix64((tag << 56) ^ (idx << 32) ^ (uint64_t)(uint32_t)v ^ ((uint64_t)(v >> 32) << 40)) ;
}

static void o_dump(const sfo_env *e, const oarena *a, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs,
                   sf_bullet_rec *bs, sf_portal_rec *ps, uint8_t *cf, int32_t *cd, int32_t *cp) {
  const int cells = e->F * e->N * e->M;
  if (hdr) {
    memset(hdr, 0, sizeof *hdr);
    hdr->frame = a->frame, hdr->kills = a->kills, hdr->teams_kills = a->teams_kills, hdr->loot = a->loot;
    hdr->chests = a->chest, hdr->jomle = a->jomle, hdr->tb = a->tb, hdr->serial = a->serial;
    hdr->steps = a->steps, hdr->episodes = a->episodes;
    for (int i = 0; i < 18; ++i) hdr->rng[i] = (int32_t)a->random[i];
    hdr->done = a->done, hdr->outcome = a->outcome;
  }
  if (hs)
    for (int i = 0; i < e->H; ++i) {
      const ohuman *h = &a->hum[i];
      sf_human_rec *o = &hs[i];
      memset(o, 0, sizeof *o);
      o->alive = h->alive, o->remote = h->remote, o->rnpc = h->rnpc, o->profile = h->prof;
      o->f = h->f, o->r = h->r, o->c = h->c, o->way = h->way, o->team = h->team;
      o->hp = h->hp, o->stamina = h->stamina, o->mindamage = h->mindamage;
      o->kills = h->kills, o->damage = h->damage, o->effect = h->effect;
      o->vec = h->vec, o->ind = h->ind;
      for (int k = 0; k < 4; ++k) o->cons[k] = h->cons[k], o->throw_cnt[k] = h->thr_cnt[k];
      o->blocks = h->blocks, o->portals = h->portals, o->portal_ind = h->portal_ind;
    }
  if (zs)
    for (int i = 0; i < e->Z; ++i) {
      const ozombie *z = &a->zomb[i];
      memset(&zs[i], 0, sizeof zs[i]);
      if (!z->alive) continue; /* dead slots are canonicalised to zero */
      zs[i].alive = 1, zs[i].f = z->f, zs[i].r = z->r, zs[i].c = z->c;
      zs[i].hp = z->hp, zs[i].mindamage = z->mindamage, zs[i].super_ = z->super_;
    }
  if (bs)
    for (int i = 0; i < e->B; ++i) {
      const obullet *b = &a->bull[i];
      memset(&bs[i], 0, sizeof bs[i]);
      if (!b->alive) continue;
      const onode *n = &a->map[cell_of(e, b->f, b->r, b->c)];
      bs[i].alive = 1, bs[i].f = b->f, bs[i].r = b->r, bs[i].c = b->c, bs[i].way = b->way;
      bs[i].traveled = abs(b->f - b->df) + abs(b->r - b->dr) + abs(b->c - b->dc);
      bs[i].damage = b->damage, bs[i].effect = b->effect, bs[i].range = b->range, bs[i].owner = b->owner;
      bs[i].ref = SB(n, S_BULLET) && n->bullet == i;
    }
  if (ps)
    for (int i = 0; i < e->P; ++i) {
      memset(&ps[i], 0, sizeof ps[i]);
      if (!a->portal[i].active) continue;
      ps[i].active = 1, ps[i].f = a->portal[i].f, ps[i].r = a->portal[i].r, ps[i].c = a->portal[i].c;
    }
  for (int ci = 0; ci < cells; ++ci) {
    const onode *n = &a->map[ci];
    uint8_t f = 0;
    if (SB(n, S_WALL)) f |= SF_CELL_WALL;
    if (SB(n, S_TEMP)) f |= SF_CELL_TEMP;
    if (SB(n, S_PIN_UP)) f |= SF_CELL_PIN_UP;
    if (SB(n, S_PIN_DN)) f |= SF_CELL_PIN_DN;
    if (SB(n, S_POUT)) f |= SF_CELL_POUT;
    if (SB(n, S_CHEST)) f |= SF_CELL_CHEST | (uint8_t)(n->cons << SF_CELL_CONS_SHIFT);
    if (cf) cf[ci] = f;
    if (cd) cd[ci] = SB(n, S_TEMP) ? n->dmg : 0;
    if (cp) cp[ci] = (SB(n, S_PIN_UP) || SB(n, S_PIN_DN)) ? n->portal_ind : -1;
  }
}

/* Order-independent digest: a sum of per-item hashes keyed by (table tag, slot, field). */
uint64_t sf_digest_from_dump(int H, int Z, int B, int P, int cells, const sf_arena_hdr *hdr, const sf_human_rec *hs,
                             const sf_zombie_rec *zs, const sf_bullet_rec *bs, const sf_portal_rec *ps,
                             const uint8_t *cf, const int32_t *cd, const int32_t *cp) {
  uint64_t d = 0;
  const int64_t hv[9] = {hdr->frame, hdr->kills, hdr->teams_kills, hdr->loot, hdr->chests, hdr->jomle, hdr->steps,
                         hdr->done, hdr->outcome};
  for (int i = 0; i < 9; ++i) d += dg(1, (uint64_t)i, hv[i]);
  for (int i = 0; i < 18; ++i) d += dg(2, (uint64_t)i, hdr->rng[i]);
  for (int i = 0; i < H; ++i) {
    const int32_t *w = (const int32_t *)&hs[i];
    for (size_t k = 0; k < sizeof(sf_human_rec) / 4; ++k) d += dg(3, (uint64_t)(i * 32 + (int)k), w[k]);
  }
  for (int i = 0; i < Z; ++i) {
    const int32_t *w = (const int32_t *)&zs[i];
    for (size_t k = 0; k < sizeof(sf_zombie_rec) / 4; ++k) d += dg(4, (uint64_t)(i * 8 + (int)k), w[k]);
  }
  for (int i = 0; i < B; ++i) {
    const int32_t *w = (const int32_t *)&bs[i];
    for (size_t k = 0; k < sizeof(sf_bullet_rec) / 4; ++k) d += dg(5, (uint64_t)(i * 16 + (int)k), w[k]);
  }
  for (int i = 0; i < P; ++i) {
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

/* ------------------------------------------------------------------------------------------------ */
/* Exported test API (mirrors include/strikeforce.h with an sfo_ prefix). */

static int o_validate(const sf_config *c) {
  if (!c || !c->map) return 0;
  if (c->arenas < 1 || c->floors < 1 || c->rows < 3 || c->cols < 3) return 0;
  if (c->rows > SF_MAX_COORD || c->cols > SF_MAX_COORD) return 0;
  if (c->cap_humans < 1 || c->cap_zombies < 1 || c->cap_bullets < 1 || c->cap_portals < 1) return 0;
  if (c->n_agents < 1 || c->n_agents > SF_MAX_AGENTS || c->n_agents > c->cap_humans) return 0;
  if (c->level < 1 || c->level > 10) return 0;
  if (c->mode == SF_MODE_SQUAD && c->cap_humans < 10) return 0;
  if (c->ind < 0 || c->ind >= c->n_agents || (c->ind != 0 && c->mode != SF_MODE_BATTLE)) return 0;
  return 1;
}

sfo_env *sfo_create(const sf_config *cfg) {
  if (!o_validate(cfg)) return NULL;
  sfo_env *e = (sfo_env *)calloc(1, sizeof *e);
  e->cfg = *cfg;
  e->F = cfg->floors, e->N = cfg->rows, e->M = cfg->cols;
  e->H = cfg->cap_humans, e->Z = cfg->cap_zombies, e->B = cfg->cap_bullets, e->P = cfg->cap_portals;
  e->C = cfg->cap_chests;
  e->ind = cfg->ind;
  const int cells = e->F * e->N * e->M;
  e->map_chars = (char *)malloc((size_t)cells);
  memcpy(e->map_chars, cfg->map, (size_t)cells);
  e->map_portal = (int16_t *)malloc(sizeof(int16_t) * (size_t)cells);
  for (int i = 0; i < cells; ++i) e->map_portal[i] = cfg->map_portal ? cfg->map_portal[i] : -1;
  e->cfg.map = e->map_chars, e->cfg.map_portal = e->map_portal;
  o_derive(cfg, &cfg->player, &e->der[0]);
  o_derive(cfg, &cfg->npc, &e->der[1]);
  e->der[1].mindamage_def += 15 * (cfg->level - 1); /* gen_human CH:882-886: 3 level-ups per level */
  e->ar = (oarena *)calloc((size_t)cfg->arenas, sizeof(oarena));
  for (int i = 0; i < cfg->arenas; ++i) {
    oarena *a = &e->ar[i];
    a->map = (onode *)calloc((size_t)cells, sizeof(onode));
    a->map1 = (onode *)calloc((size_t)cells, sizeof(onode));
    a->hum = (ohuman *)calloc((size_t)e->H, sizeof(ohuman));
    a->zomb = (ozombie *)calloc((size_t)e->Z, sizeof(ozombie));
    a->bull = (obullet *)calloc((size_t)e->B, sizeof(obullet));
    a->portal = (oportal *)calloc((size_t)e->P, sizeof(oportal));
    a->temp = (int *)calloc((size_t)cells + 1, sizeof(int));
    a->place = (int *)calloc((size_t)(2 * e->B) + 2, sizeof(int));
    a->command = (uint8_t *)calloc((size_t)(e->H > SF_MAX_AGENTS ? e->H : SF_MAX_AGENTS), 1);
    a->done = 1; /* until reset */
  }
  return e;
}

void sfo_destroy(sfo_env *e) {
  if (!e) return;
  for (int i = 0; i < e->cfg.arenas; ++i) {
    oarena *a = &e->ar[i];
    free(a->map), free(a->map1), free(a->hum), free(a->zomb), free(a->bull), free(a->portal);
    free(a->temp), free(a->place), free(a->command);
  }
  free(e->ar), free(e->map_chars), free(e->map_portal), free(e);
}

int sfo_reset(sfo_env *e, const uint64_t *tb, const uint64_t *serial) {
  for (int i = 0; i < e->cfg.arenas; ++i) {
    e->ar[i].episodes = 0;
    e->ar[i].ended_last_step = 0;
    memset(e->ar[i].results, 0, sizeof e->ar[i].results);
    o_reset_arena(e, &e->ar[i], (long long)tb[i], (long long)serial[i]);
  }
  return SF_OK;
}

int sfo_step(sfo_env *e, const uint8_t *cmd) {
  uint8_t ext[SF_MAX_HUMANS];
  for (int i = 0; i < e->cfg.arenas; ++i) {
    memset(ext, '+', sizeof ext);
    memcpy(ext, cmd + (size_t)i * e->cfg.n_agents, (size_t)e->cfg.n_agents);
    o_step_arena(e, &e->ar[i], ext);
  }
  return SF_OK;
}

/* k steps, cmd[k][arenas][n_agents] */
int sfo_step_many(sfo_env *e, const uint8_t *cmd, int32_t k) {
  const size_t stride = (size_t)e->cfg.arenas * e->cfg.n_agents;
  for (int s = 0; s < k; ++s) sfo_step(e, cmd + stride * (size_t)s);
  return SF_OK;
}

int sfo_observe(sfo_env *e, float *out) {
  for (int i = 0; i < e->cfg.arenas; ++i)
    for (int g = 0; g < e->cfg.n_agents; ++g) {
      float *o = out + ((size_t)i * e->cfg.n_agents + g) * SF_OBS_FLOATS;
      const ohuman *h = &e->ar[i].hum[g];
      if (h->alive && h->ctrl)
        o_observe_agent(e, &e->ar[i], g, o);
      else
        memset(o, 0, sizeof(float) * SF_OBS_FLOATS);
    }
  return SF_OK;
}

int sfo_results(sfo_env *e, int32_t *out) {
  for (int i = 0; i < e->cfg.arenas; ++i)
    memcpy(out + (size_t)i * e->cfg.n_agents * 8, e->ar[i].results, sizeof(int32_t) * 8 * (size_t)e->cfg.n_agents);
  return SF_OK;
}

int sfo_done(sfo_env *e, uint8_t *out) {
  for (int i = 0; i < e->cfg.arenas; ++i)
    out[i] = (uint8_t)(e->cfg.auto_reset ? e->ar[i].ended_last_step : e->ar[i].done);
  return SF_OK;
}

int sfo_dump_arena(sfo_env *e, int32_t arena, sf_arena_hdr *hdr, sf_human_rec *hs, sf_zombie_rec *zs,
                   sf_bullet_rec *bs, sf_portal_rec *ps, uint8_t *cf, int32_t *cd, int32_t *cp) {
  if (arena < 0 || arena >= e->cfg.arenas) return SF_ERR_ARG;
  o_dump(e, &e->ar[arena], hdr, hs, zs, bs, ps, cf, cd, cp);
  return SF_OK;
}

int sfo_state_digest(sfo_env *e, uint64_t *out) {
  const int cells = e->F * e->N * e->M;
  sf_arena_hdr hdr;
  sf_human_rec *hs = (sf_human_rec *)malloc(sizeof(sf_human_rec) * (size_t)e->H);
  sf_zombie_rec *zs = (sf_zombie_rec *)malloc(sizeof(sf_zombie_rec) * (size_t)e->Z);
  sf_bullet_rec *bs = (sf_bullet_rec *)malloc(sizeof(sf_bullet_rec) * (size_t)e->B);
  sf_portal_rec *ps = (sf_portal_rec *)malloc(sizeof(sf_portal_rec) * (size_t)e->P);
  uint8_t *cf = (uint8_t *)malloc((size_t)cells);
  int32_t *cd = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells);
  int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells);
  for (int i = 0; i < e->cfg.arenas; ++i) {
    o_dump(e, &e->ar[i], &hdr, hs, zs, bs, ps, cf, cd, cp);
    out[i] = sf_digest_from_dump(e->H, e->Z, e->B, e->P, cells, &hdr, hs, zs, bs, ps, cf, cd, cp);
  }
  free(hs), free(zs), free(bs), free(ps), free(cf), free(cd), free(cp);
  return SF_OK;
}

/* ---- known-answer hooks -------------------------------------------------------------------------- */
void sfo_kat_rand(uint64_t tb, uint64_t serial, int32_t n, int32_t *out) {
  oarena a;
  memset(&a, 0, sizeof a);
  o_srand(&a, (long long)tb, (long long)serial);
  for (int i = 0; i < n; ++i) out[i] = o_rand(&a);
}
int32_t sfo_kat_compute_damage(int32_t x, int32_t y) { return o_compute_damage(x, y); }
int64_t sfo_draws(sfo_env *e, int32_t arena) { return e->ar[arena].draws; }
int32_t sfo_event_count(void) { return EV_COUNT; }
const char *sfo_event_name(int32_t k) { return k >= 0 && k < EV_COUNT ? EV_NAMES[k] : ""; }
/* sum over all arenas, never reset */
void sfo_events(sfo_env *e, int64_t *out) {
  for (int k = 0; k < EV_COUNT; ++k) out[k] = 0;
  for (int i = 0; i < e->cfg.arenas; ++i)
    for (int k = 0; k < EV_COUNT; ++k) out[k] += e->ar[i].ev[k];
}

/* cpu_baseline leg of bench.py: run k steps over all arenas with the bench's LCG random agent
 * (x <- 1664525 x + 1013904223, (x >> 16) % 28 over the command set minus '3' and '_'). */
static const char BENCH_CMDS[28] = {'+', 'q', 'e', 'u', 'z', 'x', 'a', 'w', 's', 'd', 'f', 'g', 'h', 'j',
                                    'k', 'l', ';', '\'', 'c', 'v', 'b', 'n', 'm', ',', '.', '/', '[', ']'};
int64_t sfo_bench_run(sfo_env *e, int32_t k, uint32_t *lcg /* [arenas][n_agents] state, updated */) {
  uint8_t ext[SF_MAX_HUMANS];
  int64_t steps = 0;
  for (int s = 0; s < k; ++s)
    for (int i = 0; i < e->cfg.arenas; ++i) {
      memset(ext, '+', sizeof ext);
      for (int g = 0; g < e->cfg.n_agents; ++g) {
        uint32_t *x = &lcg[(size_t)i * e->cfg.n_agents + g];
        *x = *x * 1664525u + 1013904223u;
        ext[g] = (uint8_t)BENCH_CMDS[(*x >> 16) % 28];
      }
      o_step_arena(e, &e->ar[i], ext);
      ++steps;
    }
  return steps;
}
