"""Hand-derived known answers for the ORDER-SENSITIVE rules of the tick path: who comes last when two humans aim at one
cell (gameplay.hpp:1002-1011, 814-816), which of two bullets entering one cell in the same update_bull stays the cell's
bullet and what becomes of the other (gameplay.hpp:1073-1098), a blocked shot still costing stamina and
ammunition (gameplay.hpp:803-817, E-5), a bullet on a player-built '^' being absorbed — or not, when somebody stands
there (gameplay.hpp:1343-1353, showit's priority :321-341), the Battle end condition (gameplay.hpp:1103).

Each expectation is written out from the cited reference lines; the sweep directions, which the generator decides, are
taken from the generator's own known-answer function (the draw's index is derived by hand and asserted through the
draw counter).  Run on the oracle, on the emulated device core, (-m gpu) on the device — and, where the reference's
command set allows it (the keyboard player has all 30 commands; its Squad agents the nine of Custom.hpp:162), on the
reference itself (oracle/_ref/sf_ref_tick*, oracle/ref_tick.py).  Seeds were chosen by search; nothing else was."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import reftick
from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import abi, config

ROWS, COLS = 30, 100  # the reference's own dimensions (gameplay.hpp:37), so that it can run the same scenario
SERIAL = 123456789
# a player record that is told apart from the NPC record by its punch and its shot: mindamage 300, level 1 everywhere
# (no level-ups in Human::build, Character.hpp:650-709), one of each item, every weapon at level 1
P300 = [1000, 300, 1000000, 1, 1, 1, 1000, 0, 0, 0, 0] + [1] * 4 + [1, 1] * 4 + [1] * 8 + [1]


def kat(tb, n):
    """The first n outputs of _rand() after _srand(tb, SERIAL): random.hpp:54-76 (pinned on the reference, test_ref_slices)."""
    out = (C.c_int32 * n)()
    oracle_lib.lib().sfo_kat_rand(tb, SERIAL, n, out)
    return list(out)


def world(open_cells, mode, n_agents, level=1, player=P300, H=12, Z=4, B=8, P=4, teams=None):
    """3 x 30 x 100, solid except `open_cells` [(f, r, c)]: the random spawns (gameplay.hpp:532-572) need a '.' cell and
    almost never find one; every scenario asserts that none did."""
    grid = [["#"] * (ROWS * COLS) for _ in range(3)]
    for f, r, c in open_cells:
        grid[f][r * COLS + c] = "."
    m = "".join("".join(g) for g in grid).encode()
    cfg = config.make_config(1, ROWS, COLS, floors=3, H=H, Z=Z, B=B, P=P, chests=4, mode=mode, level=level, n_agents=n_agents,
                             player_tokens=player, auto_reset=0, teams=teams, timer_frames=1 << 20)
    return config.Workload("scenario", cfg, m, [-1] * (3 * ROWS * COLS))


SQUAD_START = [(0, 3, 1)] + [(0, 1, i + 1) for i in range(1, 5)] + [(2, 1, i + 1) for i in range(5, 10)]  # gameplay.hpp:1861-1903


class Sim:
    """One implementation stepping a scenario: commands are a string, one char per commanded human."""

    def __init__(self, impl, w, tb):
        self.w, self.tb = w, tb
        self.sim = impl(w)
        self.sim.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(SERIAL))
        self.n = w.cfg.n_agents

    def step(self, chars):
        chars = chars + "+" * (self.n - len(chars))
        self.sim.step(np.frombuffer(chars.encode(), dtype=np.uint8))
        return self.dump()

    def dump(self):
        return self.sim.dump(0)


class RefSim:
    """The reference itself (Squad: built with USE_AGENT_IN_SQUAD_NPCS, keyboard player + scripted agents)."""

    def __init__(self, w, tb, player, squad, native_caps=True):
        self.like = Oracle(w)  # only for the dump fields the reference has no counterpart of (reftick.RefTick.dump)
        self.r = reftick.RefTick(w, player, agents=False, squad_agents=squad, native_caps=native_caps)
        self.r.reset(tb, SERIAL)
        self.w = w

    def step(self, chars):
        self.r.step(chars + "+" * (10 - len(chars)) if self.w.cfg.mode == abi.MODE_SQUAD else chars[:1])
        return self.dump()

    def dump(self):
        return as_dump(self.r.dump(), self.w.cfg)


class Rec:
    def __init__(self, names, row):
        for n, v in zip(names, row):
            setattr(self, n, int(v))


def as_dump(arrs, cfg):
    """reftick's arrays in the attribute form of oracle_lib.ArenaDump (what the expectations below read)."""
    hn = [n for n, _ in abi.HumanRec._fields_[:17]]
    d = type("D", (), {})()
    d.humans = []
    for row in arrs["humans"]:
        h = Rec(hn, row[:17])
        h.cons, h.throw_cnt = [int(x) for x in row[17:21]], [int(x) for x in row[21:25]]
        h.blocks, h.portals, h.portal_ind = int(row[25]), int(row[26]), int(row[27])
        d.humans.append(h)
    d.zombies = [Rec([n for n, _ in abi.ZombieRec._fields_], r) for r in arrs["zombies"]]
    d.bullets = [Rec([n for n, _ in abi.BulletRec._fields_], r) for r in arrs["bullets"]]
    d.portals = [Rec([n for n, _ in abi.PortalRec._fields_], r) for r in arrs["portals"]]
    d.flags, d.dmg, d.pidx = arrs["flags"], arrs["dmg"], arrs["pidx"]
    d.hdr = Rec(reftick.HDR_NAMES[:7], arrs["hdr"][:7])
    return d


def Device(w):
    from strikeforce_amd import env
    return env.ArenaBatch(w)


needs_ref = pytest.mark.skipif(not reftick.available(), reason="no reference build on this machine")
IMPLS = [pytest.param(Oracle, id="oracle"), pytest.param(Emu, id="emu"), pytest.param(Device, id="device", marks=pytest.mark.gpu)]
IMPLS_REF = IMPLS + [pytest.param("reference", id="reference", marks=needs_ref)]


def ref_player():
    """the level-10 account (15000 Hp): survives whatever the yard throws at it for 80 steps"""
    import ref_cases
    return list(ref_cases.RICH)


def phase_draws(s):
    """draws of the last step by phase [zombie_action, update_bull, human_action, update_bull, spawns, rest]: from the
    oracle, the reference driver, the emulator and the device (sf_phase_draws)"""
    if isinstance(s, RefSim):
        return s.r.phase_draws
    if hasattr(s.sim, "phase_draws"):
        return s.sim.phase_draws(0)
    pytest.skip("no per-phase draw counter on this implementation")


def close(s):
    if isinstance(s, RefSim):
        s.r.close()


def make(impl, w, tb, player=P300, squad=False, native_caps=True):
    return RefSim(w, tb, player, squad, native_caps) if impl == "reference" else Sim(impl, w, tb)


def quiet(d, humans):
    """no random spawn found a '.' cell: nobody but the scenario's own humans, no zombie, no chest"""
    return sum(h.alive for h in d.humans) <= humans and sum(z.alive for z in d.zombies) == 0 and d.hdr.chests == 0


def draws_ok(d, steps):
    """9 draws at the first loop top (three spawn attempts of three coordinates, frame 1: gameplay.hpp:1444-1449), then
    3 per quiet step: update_bull, human_action's sweep, update_bull (gameplay.hpp:1073,1002) — before frame 31, the
    next spawn.  jomle = 18 + 1024 warm-up draws + those (random.hpp:29,72-74)."""
    return d.hdr.jomle == 18 + 1024 + 9 + 3 * steps


def seed_with(bits):
    """the first tb >= 1 700 000 000 whose draws have the wanted low bits: {draw index: bit}"""
    for tb in range(1700000000, 1700000400):
        k = kat(tb, max(bits) + 1)
        if all((k[i] & 1) == b for i, b in bits.items()):
            return tb
    raise AssertionError("no seed")


# ---- two humans aim at one cell ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("r", [1, 0], ids=["ascending", "descending"])
@pytest.mark.parametrize("impl", IMPLS_REF)
def test_two_punches_on_one_cell_the_last_in_sweep_order_lands(impl, r):
    """Squad.  The player (slot 0, punch 300 = max(compute_damage(300, 1) = 17, mindamage 300), Character.hpp:391-397)
    walks to (2,3) and faces up; team mate 1 (slot 1, NPC record at level 1: punch 100) at (1,2) faces right; both punch
    the cell (1,3), on which team mate 2 (slot 2) stands, in the same step (step 5).  obey puts each punch on the target
    cell and makes it the cell's bullet (gameplay.hpp:796-819): the second one overwrites the pointer and the first is
    orphaned.  hit_human then applies the cell's bullet only (gameplay.hpp:611-620).  The sweep runs ascending when the
    draw `rand() & 1` is 1, descending when 0 (gameplay.hpp:1002-1004): ascending -> the mate's 100 lands, descending ->
    the player's 300.  Both bullets are gone by the end of the step (range 1: expire(), Item.hpp:165-168).  Friendly
    fire is applied but not credited (gameplay.hpp:613-619)."""
    tb = seed_with({9 + 3 * 5 + 1: r})
    w = world(SQUAD_START + [(0, 3, 2), (0, 3, 3), (0, 2, 3)], abi.MODE_SQUAD, 5)
    s = make(impl, w, tb, squad=True)
    assert quiet(s.dump(), 10)
    for chars in ("dq", "d", "w", "q", "q"):  # the player: (3,1) -> (3,3) -> (2,3), way 1 -> 2 -> 3 (turn_l, Character.hpp:752-758)
        d = s.step(chars)
    me, m1, m2 = d.humans[0], d.humans[1], d.humans[2]
    assert (me.r, me.c, me.way) == (2, 3, 3) and (m1.r, m1.c, m1.way) == (1, 2, 2) and (m2.r, m2.c, m2.hp) == (1, 3, 1000)
    d = s.step("zz")
    assert draws_ok(d, 6) and quiet(d, 10)
    assert d.humans[2].hp == (1000 - 100 if r == 1 else 1000 - 300)
    assert d.humans[2].mindamage == 100 and d.humans[2].alive == 1
    assert sum(b.alive for b in d.bullets) == 0
    assert (d.humans[0].damage, d.humans[1].damage, d.humans[0].kills) == (0, 0, 0)


# ---- two bullets enter one cell -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("r_h,r_b", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("impl", IMPLS)
def test_two_bullets_enter_one_cell_the_last_entrant_is_the_cells_bullet(impl, r_h, r_b):
    """Squad, three commanded humans with guns (this repo's command surface: any of the 30 codes per commanded human;
    the reference's Squad agents have nine).  X = (4,2).  Team mate 1 walks down to X.  Team mate 2 steps to (1,2) and
    faces down: its AK_47 shot (200 + mindamage 100 = 300, effect -105, Character.hpp:399-408) is put on (2,2) and flies
    (3,2), X.  The player walks to (4,5), faces left: its shot (200 + 300 = 500) is put on (4,4) and flies (4,3), X.
    Fired in the same step t = 11, both bullets enter X in the first update_bull of step 12.
      * bullet slots: b_ind hands out the lowest free slot (gameplay.hpp:230-235) in sweep order: r_h = 1 ascending ->
        the player's bullet is slot 0, the mate's slot 1; r_h = 0 the other way round.
      * update_bull walks the slots ascending when r_b = 1, descending when 0; every entrant overwrites the cell's
        pointer (gameplay.hpp:1087-1089): the cell's bullet is the LAST one walked.
      * hit_human of that step's second half hits the man on X with the cell's bullet only, and consumes it
        (gameplay.hpp:611-634): Hp 1000 - 500 or - 300.
      * the other bullet is still alive but no cell points at it (App. E-4).  The second update_bull moves it on:
        the player's, flying left, enters (4,1) alone and is that cell's bullet again; the mate's, flying down, meets
        the wall at (5,2) and dies (gameplay.hpp:1085-1093)."""
    t = 11
    tb = seed_with({9 + 3 * t + 1: r_h, 9 + 3 * (t + 1): r_b})
    opened = SQUAD_START + [(0, 2, 2), (0, 3, 2), (0, 4, 2), (0, 4, 1), (0, 4, 3), (0, 4, 4), (0, 4, 5)]
    w = world(opened, abi.MODE_SQUAD, 3)
    s = make(impl, w, tb)
    # moves: 's' down, 'd' right, 'a' left (gameplay.hpp:742-758).  The player: (3,1) -> (4,1) -> ... -> (4,5); mate 1 waits
    # on (3,2) until the player has passed X (two humans moving through one cell in one sweep is itself order-dependent)
    script = ["ss", "ds", "d+", "ds", "d+a", "e", "m+m", "+", "+", "+", "+"]  # steps 0 .. 10
    for chars in script:
        d = s.step(chars)
    me, v, a = d.humans[0], d.humans[1], d.humans[2]
    assert (me.r, me.c, me.way, me.vec, me.ind) == (4, 5, 4, 2, 4)      # turn_r: 1 -> 4 (Character.hpp:745-751); AK_47 = weapon 4
    assert (v.r, v.c, v.hp) == (4, 2, 1000)
    assert (a.r, a.c, a.way, a.vec, a.ind) == (1, 2, 1, 2, 4)
    assert quiet(d, 10) and draws_ok(d, t)
    d = s.step("x+x")  # step 11: both fire; the second update_bull of the step moves both one cell
    assert draws_ok(d, t + 1)
    live = sorted((b.r, b.c, b.way, b.damage, b.owner) for b in d.bullets if b.alive)
    assert live == [(3, 2, 1, 300, 3), (4, 3, 4, 500, 1)]
    slot_of_player = [i for i, b in enumerate(d.bullets) if b.alive and b.owner == 1][0]
    assert slot_of_player == (0 if r_h == 1 else 1)
    d = s.step("+")    # step 12
    assert draws_ok(d, t + 2) and quiet(d, 10)
    order = [0, 1] if r_b == 1 else [1, 0]           # slots in the order update_bull walked them
    last_is_player = order[-1] == slot_of_player
    assert d.humans[1].hp == 1000 - (500 if last_is_player else 300)
    assert d.humans[1].mindamage == 100 - 105
    live = [(b.r, b.c, b.way, b.damage, b.owner, b.ref) for b in d.bullets if b.alive]
    if last_is_player:   # the mate's bullet was the orphan: it flew into the wall below X
        assert live == []
    else:                # the player's bullet was the orphan: alone on (4,1) it is a cell's bullet again
        assert live == [(4, 1, 4, 500, 1, 1)]
    # the hit is friendly fire: applied, not credited (gameplay.hpp:613-619)
    assert (d.humans[0].damage, d.humans[2].damage) == (0, 0)
    d = s.step("+")    # step 13: (4,0) is the border wall
    assert sum(b.alive for b in d.bullets) == 0


# ---- a blocked shot still costs --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS_REF)
def test_a_blocked_shot_spends_stamina_and_ammunition(impl):
    """Solo, the player at (1,1) turned to the border wall above it.  obey's shot path runs punch / throw_it / shot_it —
    which mutate the bullet slot and spend stamina and the throwable — BEFORE it looks at the target cell
    (gameplay.hpp:803-813); the slot is only activated if the cell lets a bullet in (:814-818).  Throwing the gas:
    count 1 -> 0, stamina -15, vec back to -1 (Character.hpp:410-427).  Firing the AK_47: stamina -50
    (Character.hpp:399-408).  A punch costs nothing.  No bullet ever lives."""
    w = world([(0, 1, 1), (0, 2, 1)], abi.MODE_SOLO, 1, player=config.HUMAN_ENEMY_TOKENS)
    s = make(impl, w, 1700000000, player=config.HUMAN_ENEMY_TOKENS)
    for n, (ch, stamina, gas, vec) in enumerate([("q", 1000000, 1, -1), ("q", 1000000, 1, -1),   # way 1 -> 2 -> 3: up
                                                 ("k", 1000000, 1, 1), ("x", 999985, 0, -1),      # the throw
                                                 ("m", 999985, 0, 2), ("x", 999935, 0, 2), ("x", 999885, 0, 2),
                                                 ("z", 999885, 0, 2)], start=1):
        d = s.step(ch)
        h = d.humans[0]
        assert (h.stamina, h.throw_cnt[0], h.vec) == (stamina, gas, vec), (n, ch)
        assert sum(b.alive for b in d.bullets) == 0 and quiet(d, 1) and draws_ok(d, n)
    assert d.humans[0].way == 3 and (d.humans[0].r, d.humans[0].c) == (1, 1)


# ---- a bullet on a player-built entrance ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS_REF)
def test_bullets_on_a_built_entrance_are_absorbed_until_it_breaks(impl):
    """Solo.  ']' builds the exit 'O' on (2,1) (exit 0), a turn and a second ']' the entrance '^' on (1,2)
    (gameplay.hpp:716-733).  An AK_47 shot (300) at the entrance is put ON it — a '^' lets a bullet in only because it
    is destructible, s[10] (gameplay.hpp:814) — and update_tmp of the same half-tick absorbs it: dmg += 300, bullet
    gone (gameplay.hpp:1343-1353).  The fourth shot takes dmg to 1200 >= lim_portal = 1000: entrance and exit vanish
    and exit 0 is free again (gameplay.hpp:1355-1362)."""
    w = world([(0, 1, 1), (0, 2, 1), (0, 1, 2)], abi.MODE_SOLO, 1, player=config.HUMAN_ENEMY_TOKENS)
    s = make(impl, w, 1700000000, player=config.HUMAN_ENEMY_TOKENS)
    O, UP = 2 * COLS + 1, 1 * COLS + 2
    for ch in "]q]m":
        d = s.step(ch)
    assert d.flags[O] == abi.CELL_POUT | abi.CELL_TEMP and d.flags[UP] == abi.CELL_PIN_UP | abi.CELL_TEMP and d.pidx[UP] == 0
    assert d.portals[0].active == 1 and d.humans[0].way == 2
    for n in (1, 2, 3):
        d = s.step("x")
        assert d.dmg[UP] == 300 * n and d.flags[UP] == abi.CELL_PIN_UP | abi.CELL_TEMP
        assert sum(b.alive for b in d.bullets) == 0 and d.humans[0].stamina == 1000000 - 50 * n
    d = s.step("x")
    assert d.flags[UP] == 0 and d.flags[O] == 0 and d.dmg[UP] == 0 and d.pidx[UP] == -1
    assert d.portals[0].active == 0 and quiet(d, 1) and draws_ok(d, 8)
    d = s.step("x")  # the cell is plain floor now: the shot flies (1,2) -> (1,3) is wall: it dies in the first update_bull
    assert sum(b.alive for b in d.bullets) == 0 and d.dmg[UP] == 0


# ---- a covered exit radiates only while a bullet slot is free (App. E-9) ------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS_REF)
def test_a_covered_exit_radiates_only_while_the_bullet_pool_has_a_free_slot(impl):
    """Solo, a bullet pool of ONE slot (B = 1; the reference compiled with that pool).  The player builds the exit 'O' on
    (2,1) and the entrance '^' on (1,2) as above, takes the AK_47 in hand and steps onto its own entrance: teleport() puts
    it on the exit (gameplay.hpp:517-530), facing right (a move does not turn).  An exit that no longer shows 'O' radiates:
    portal_damage() puts a bullet of damage 20, effect -10, range 1 on it (gameplay.hpp:1279-1297), hit_human of the same
    half-tick gives it to the man standing there and frees the slot (:611-634): Hp -20, mindamage -10 per step.
    Then the player fires along the corridor (2,2)..(2,7): the shot takes the pool's only slot in human_action (after
    that step's radiation), moves one cell per update_bull — two per step — and dies at the wall (2,8) in the first
    update_bull of the third step after the shot (it is put on column 2 and needs columns 3..7: five moves, the sixth
    meets the wall; gameplay.hpp:1080-1093).  portal_damage runs BEFORE that update_bull, so for three steps
    `index = b_ind()` is -1 and the function returns without radiating (:1287-1289): the Hp stands still, then falls again."""
    w = world([(0, 1, 1), (0, 2, 1), (0, 1, 2)] + [(0, 2, c) for c in range(2, 8)], abi.MODE_SOLO, 1,
              player=config.HUMAN_ENEMY_TOKENS, B=1)
    s = make(impl, w, 1700000000, player=config.HUMAN_ENEMY_TOKENS, native_caps=False)
    for ch in "]q]md":
        d = s.step(ch)
    h = d.humans[0]
    assert (h.r, h.c, h.way, h.vec, h.ind, h.hp, h.mindamage) == (2, 1, 2, 2, 4, 1000, 100) and quiet(d, 1)
    d = s.step("+")                                   # step 6: the first radiation
    assert (d.humans[0].hp, d.humans[0].mindamage) == (980, 90) and sum(b.alive for b in d.bullets) == 0
    d = s.step("x")                                   # step 7: radiation, then the shot: on (2,2), moved to (2,3)
    assert (d.humans[0].hp, d.humans[0].mindamage, d.humans[0].stamina) == (960, 80, 1000000 - 50)
    assert [(b.r, b.c, b.way) for b in d.bullets if b.alive] == [(2, 3, 2)]
    for n, col in ((8, 5), (9, 7), (10, None)):      # the pool is dry when portal_damage looks: no radiation
        d = s.step("+")
        assert (d.humans[0].hp, d.humans[0].mindamage) == (960, 80), n
        assert [(b.r, b.c) for b in d.bullets if b.alive] == ([(2, col)] if col else []), n
    d = s.step("+")                                   # step 11: the slot is free again
    assert (d.humans[0].hp, d.humans[0].mindamage) == (940, 70)
    assert quiet(d, 1) and draws_ok(d, 11)
    close(s)


@pytest.mark.parametrize("impl", IMPLS)
def test_a_bullet_on_an_entrance_somebody_stands_on_hits_him_instead(impl):
    """Squad, three commanded humans.  Team mate 1 stands on (1,2) facing down: ']' -> exit 'O' on (2,2); 'e' (way 1 -> 4,
    left), ']' -> entrance '^' on (1,1) leading to exit 0.  Nobody can WALK onto an exit ('O' is not among the cells a
    move may enter, gameplay.hpp:750), so mate 1 covers it the only way there is: it steps onto its own entrance and is
    teleported there (gameplay.hpp:517-530).  A covered exit no longer shows 'O', so teleport() leaves the next one who
    steps on the entrance where he is (:521-523).  The player walks (3,1) -> (2,1) -> (1,1): onto the '^' (a human may enter '^',
    gameplay.hpp:750) and stays.  Team mate 2, moved to (1,2) and turned left, shoots at (1,1): showit() of that cell is
    the human on it, not '^' (priority, gameplay.hpp:321-341), so update_tmp does not absorb the bullet
    (gameplay.hpp:1348-1349) and hit_human gives it to the player: Hp 1000 - 300, mindamage 300 - 105; the entrance's
    dmg stays 0.  (The man on the exit takes the radiation meanwhile: 20 per step, gameplay.hpp:1279-1297.)"""
    opened = SQUAD_START + [(0, 2, 2), (0, 1, 1), (0, 2, 1)]
    w = world(opened, abi.MODE_SQUAD, 3)
    s = make(impl, w, 1700000000)
    #          player  mate1  mate2
    script = ["+]",            # exit on (2,2)
              "+e", "+]",      # entrance on (1,1)
              "+a",            # mate 1 onto its entrance (1,1) -> teleported to its exit (2,2)
              "w+a",           # player (3,1) -> (2,1); mate 2 (1,3) -> (1,2)
              "w+e",           # player onto the '^' (1,1); mate 2 faces left
              "++m"]
    for chars in script:
        d = s.step(chars)
    UP = 1 * COLS + 1
    assert d.flags[UP] == abi.CELL_PIN_UP | abi.CELL_TEMP and d.pidx[UP] == 0
    assert (d.humans[0].r, d.humans[0].c) == (1, 1), "the exit is covered: no teleport"
    assert (d.humans[1].r, d.humans[1].c) == (2, 2) and (d.humans[2].r, d.humans[2].c, d.humans[2].way) == (1, 2, 4)
    hp0 = d.humans[0].hp
    d = s.step("++x")
    assert d.humans[0].hp == hp0 - 300 and d.humans[0].mindamage == 300 - 105
    assert d.dmg[UP] == 0 and sum(b.alive for b in d.bullets) == 0
    assert quiet(d, 10)


# ---- Battle: the match ends when the rivals are dead ---------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS)
def test_battle_ends_when_every_rival_is_dead(impl):
    """Battle, three players of three teams in a corridor.  '_' takes a player's Hp to 0 (gameplay.hpp:696-699) and
    hit_human clears its slot in the same step (:641-645).  check_end's first test is `online && rivals_are_dead()`
    (:1103,497-505): with one rival left the match goes on; once the second is gone too the next loop top ends it, won,
    whatever the kill counters say."""
    opened = [(0, 1, c) for c in range(1, 8)]
    w = world(opened, abi.MODE_BATTLE, 3, teams=[1, 2, 3], H=4)
    s = make(impl, w, 1700000000)
    d = s.step("+_+")
    assert (d.humans[1].alive, d.humans[1].hp) == (0, 0) and (d.hdr.done, d.hdr.outcome) == (0, abi.RUNNING)
    d = s.step("++_")
    assert d.humans[2].alive == 0 and (d.hdr.done, d.hdr.outcome) == (1, abi.WON)
    assert (d.hdr.kills, d.hdr.teams_kills, d.hdr.loot) == (0, 0, 0)


# ---- Squad: dead rivals alone do not end the game ------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS)
def test_squad_needs_kills_and_dead_rivals(impl):
    """Squad ends on `level * 10 <= teams_kills && rivals_are_dead()` (gameplay.hpp:1186).  All five opponents leave by
    '_' (Hp 0, gameplay.hpp:696-699; nobody's kill: teams_kills stays 0): rivals_are_dead() is true (:497-505) but the
    game goes on; the player's own death then ends it, lost (:1131)."""
    w = world(SQUAD_START, abi.MODE_SQUAD, 10)
    s = make(impl, w, 1700000000)
    d = s.step("+++++" + "_____")
    assert [h.alive for h in d.humans[5:10]] == [0] * 5 and [h.alive for h in d.humans[:5]] == [1] * 5
    assert (d.hdr.teams_kills, d.hdr.done) == (0, 0)
    d = s.step("+")
    assert d.hdr.done == 0
    d = s.step("_")
    assert (d.hdr.done, d.hdr.outcome) == (1, abi.DIED)


# ---- a hit is credited to whoever holds the owner's slot now (App. E-11) ------------------------------------------------------
def slot_reuse_world():
    """Two corridors, columns 1 and 3 (rows 1..28), joined along row 28; everything else wall."""
    opened = [(0, r, 1) for r in range(1, 29)] + [(0, r, 3) for r in range(1, 29)] + [(0, 28, 2)]
    return world(opened, abi.MODE_SOLO, 1, player=ref_player(), H=6, Z=4, B=48, P=4)


def slot_reuse_script(shoot, steps=66):
    """The player: down its corridor (27 steps), over to the foot of the other one (2), facing up (turn_l twice,
    Character.hpp:752-758), AK_47 in hand ('m'), one shot at step `shoot`, waiting otherwise."""
    script = ["s"] * 27 + ["d", "d", "q", "q", "m"]
    script += ["+"] * (steps - len(script))
    script[shoot] = "x"
    return script


@pytest.mark.parametrize("impl", IMPLS_REF)
def test_a_hit_is_credited_to_whoever_holds_the_owners_slot_now(impl):
    """Solo, the level-10 account.  A bullet's owner is a `Human*` into hum[] (gameplay.hpp:613, Item.hpp:156-157), so a
    hit is credited to whoever holds that SLOT when it lands (App. E-11).  The world is two corridors joined at the
    bottom; the seed puts the NPC human of the first loop top into the right-hand one, at (12,3) (found by
    tests/tools/find_slot_reuse_credit.py; NPC humans never turn — human_rnpc_bot has no 'q' / 'e', gameplay.hpp:1927-1940 —
    so it fires its gun down the corridor whenever its draw says 'x').  The player walks round to the corridor's foot,
    (28,3), turns up and fires ONE shot at step 35 (damage 1075): it reaches the NPC at step 43, Hp 1000 - 1075, the
    player's kill.  The NPC's last bullet (damage 350, effect -125) is then still on its way down.  The spawn of frame
    101 (the loop top behind step 49, gameplay.hpp:1448-1449) finds a free cell, (1,3), and h_ind() hands out the lowest
    free slot, the dead NPC's (gameplay.hpp:215-221): a fresh NPC — Hp 1000, nothing to its name, no weapon in hand
    (it cannot have hit anybody).  In step 51 the old bullet reaches the player: Hp -350, mindamage -125 — and the
    350 / -125 go onto the account of the NEWCOMER in slot 1."""
    tb, shoot = 1700051496, 35
    w = slot_reuse_world()
    s = make(impl, w, tb, player=ref_player())
    d = s.dump()
    npc = d.humans[1]
    assert (npc.alive, npc.r, npc.c, npc.rnpc, npc.team, npc.hp, npc.way) == (1, 12, 3, 1, 0, 1000, 1)
    snap = {}
    for n, ch in enumerate(slot_reuse_script(shoot, steps=53)):
        snap[n] = d = s.step(ch)
        assert sum(z.alive for z in d.zombies) == 0 and sum(h.alive for h in d.humans) <= 2, n
        assert not (s.r.ended if isinstance(s, RefSim) else d.hdr.done), n  # one kill of the five a level-1 Solo game needs
    me = lambda n: snap[n].humans[0]
    one = lambda n: snap[n].humans[1]
    flying = lambda n: sorted((b.r, b.c, b.way, b.damage, b.owner) for b in snap[n].bullets if b.alive)
    assert (me(34).r, me(34).c, me(34).way, me(34).vec, me(34).ind) == (28, 3, 3, 2, 4)
    assert (26, 3, 3, 1075, 1) in flying(35)                                   # the player's shot, two cells up already
    assert one(42).alive == 1 and flying(42) == [(12, 3, 3, 1075, 1)]           # one cell below the NPC, (11,3)
    assert (one(43).alive, one(43).hp) == (0, 1000 - 1075)
    assert (me(43).kills, me(43).damage, snap[43].hdr.kills) == (1, 1075, 1)
    old = (one(43).damage, one(43).effect, one(43).kills)                       # what the dead NPC's own hits had earned it
    assert old == (4 * 350, 4 * -125, 0)
    assert flying(48) == [(23, 3, 1, 350, 2)] and one(48).alive == 0           # its last bullet, owner = slot 1
    assert (one(48).damage, one(48).effect) == old[:2]
    new = one(49)                                                               # the loop top behind step 49: frame 101
    assert (new.alive, new.r, new.c, new.hp, new.damage, new.effect, new.kills, new.vec, new.rnpc) == (1, 1, 3, 1000, 0, 0, 0, -1, 1)
    assert flying(50) == [(27, 3, 1, 350, 2)] and (me(50).hp, me(50).mindamage) == (me(49).hp, me(49).mindamage)
    assert flying(51) == [] and (me(51).hp, me(51).mindamage) == (me(50).hp - 350, me(50).mindamage - 125)
    assert (one(51).damage, one(51).effect, one(51).kills, one(51).vec) == (350, -125, 0, -1)   # credited to the newcomer
    if impl is Oracle:
        assert s.sim.events()["credit_slot_reused"] == 1
    close(s)


# ---- the NPC policy's draw paths ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", IMPLS_REF)
def test_npc_humans_draw_one_or_three_times_and_pick_weapons_on_the_fiftieth_frame(impl):
    """Solo, level 3, three open floors.  The first NPC human is spawned at the very first loop top (frame 1,
    gameplay.hpp:1448-1449,559-572: three coordinate draws, no fourth; team 0, Hp 1000 at any level — the level-ups of
    gen_human raise def_Hp only, Character.hpp:873-888, App. E-6), the next ones at frames 51, 101.  Every step human_action asks human_rnpc_bot for its
    command (gameplay.hpp:1927-1940) before the sweep draw: on a frame with frame % 50 <= 1 ONE draw picks a weapon,
    "cvbnm,./"[d % 8] (all eight owned at level 1: obey selects it, vec = 2, ind = d % 8, gameplay.hpp:781-791);
    otherwise d1 % 5 < 3 -> 'x' after ONE draw, else THREE draws (d2 chooses the table, d3 the entry).  The test walks
    the generator's known answers (random.hpp:54-62) through every step by hand — zombie_action's share of each step's
    draws is taken from the implementation under test (it is not what this scenario is about) — and checks the number of
    draws human_action made in every step and, on the weapon-pick frames, the weapon the NPC holds."""
    opened = [(f, r, c) for f in range(3) for r in range(1, ROWS - 1) for c in range(1, COLS - 1)]  # three open floors
    w = world(opened, abi.MODE_SOLO, 1, level=3, player=ref_player(), H=8, Z=16, B=64)
    tb = 1700000000
    s = make(impl, w, tb, player=ref_player())
    d = s.dump()
    assert sum(h.alive for h in d.humans) == 2, "the frame-1 spawn found a free cell (28 x 98 of 30 x 100 are)"
    npc = d.humans[1]
    assert (npc.alive, npc.team, npc.rnpc, npc.hp, npc.mindamage, npc.stamina, npc.way) == (1, 0, 1, 1000, 100, 1000000, 1)
    k = kat(tb, 6000)
    picks = threes = 0
    for step in range(0, 85):
        before = d.hdr.jomle - 18 - 1024           # draws made so far
        frame_in_human_action = d.hdr.frame + 1    # the first half-tick's ++frame has happened (gameplay.hpp:1460)
        npcs = [i for i, h in enumerate(d.humans) if i != 0 and h.alive and h.rnpc]
        alive_before = sum(h.alive for h in d.humans)
        d = s.step("+")
        if sum(h.alive for h in d.humans) < alive_before or not d.humans[0].alive:
            break                                  # somebody died in this step: who was still asked is not visible from outside
        za = phase_draws(s)[0]                      # zombie_action's draws of this step (not under test here)
        i = before + za + 1                         # + the first update_bull's draw: index of the first NPC's first draw
        want = 0
        for slot in npcs:                           # slot order, gameplay.hpp:985
            d1 = k[i]
            if frame_in_human_action % 50 <= 1:
                n = 1
                picks += 1
                assert (d.humans[slot].vec, d.humans[slot].ind) == (2, d1 % 8), (step, slot)
            elif d1 % 5 < 3:
                n = 1
            else:
                n = 3
                threes += 1
            i += n
            want += n
        assert phase_draws(s)[2] == want + 1, (step, npcs, phase_draws(s))   # + the sweep-direction draw
    assert picks >= 1 and threes >= 5 and step >= 60
    close(s)
