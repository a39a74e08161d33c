"""Hand-derived known answers from the reference's formulas, checked on the oracle, on the device core run on the wave
emulator and (-m gpu) on the device itself through the C-ABI.  Each expectation cites the reference lines it was derived from
(G = gameplay.hpp, CH = Character.hpp, IT = Item.hpp, files under /root/reference/StrikeForce-client).

The player is sealed into a 3x5 room (rows 1-3, cols 1-5) of an otherwise solid map, so the periodic random spawns
(G:532-572) almost never land next to it; every scenario asserts that none did."""
import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import abi, config


def room_workload(mode=abi.MODE_SOLO, player=None, timer_frames=0, auto_reset=0, seed=1700000000):
    rows = cols = 16
    grid = [["#"] * cols for _ in range(rows)]
    for r in range(1, 4):
        for c in range(1, 6):
            grid[r][c] = "."
    m = "".join("".join(row) for row in grid).encode()
    cfg = config.make_config(1, rows, cols, H=2, Z=2, B=8, P=4, mode=mode, auto_reset=auto_reset,
                             player_tokens=player or config.HUMAN_ENEMY_TOKENS, timer_frames=timer_frames)
    w = config.Workload("room", cfg, m, [-1] * (rows * cols))
    w.seed = seed
    return w


def run(w, script, impl):
    sim = impl(w)
    tb, sr = w.seeds(base_tb=w.seed)
    sim.reset(tb, sr)
    snaps = [sim.dump(0)]
    for ch in script:
        sim.step(np.array([ord(ch)], dtype=np.uint8))
        snaps.append(sim.dump(0))
    return sim, snaps


def quiet(d):
    """no random spawn reached the room"""
    return (sum(z.alive for z in d.zombies) == 0 and sum(h.alive for h in d.humans[1:]) == 0 and d.hdr.chests == 0)


def Device(w):
    """The HIP path through the C-ABI (same call surface as Oracle / Emu): only under `-m gpu`."""
    from strikeforce_amd import env
    return env.ArenaBatch(w)


IMPLS = [pytest.param(Oracle, id="oracle"), pytest.param(Emu, id="emu"), pytest.param(Device, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("impl", IMPLS)
def test_initial_placement_and_profile(impl):
    # Solo: player at (0,1,1), way 1, team 1 (G:1905-1920); human_enemy profile: Hp 1000, mindamage 100,
    # stamina 1e6, one of each consumable/throwable, 8 blocks, 1 portal (CH:78-79,650-709)
    _, s = run(room_workload(), "", impl)
    h = s[0].humans[0]
    assert (h.alive, h.f, h.r, h.c, h.way, h.team) == (1, 0, 1, 1, 1, 1)
    assert (h.hp, h.mindamage, h.stamina) == (1000, 100, 1000000)
    assert list(h.cons) == [1, 1, 1, 1] and list(h.throw_cnt) == [1, 1, 1, 1]
    assert (h.blocks, h.portals, h.portal_ind, h.vec, h.ind) == (8, 1, -1, -1, -1)
    assert s[0].hdr.frame == 1


@pytest.mark.parametrize("impl", IMPLS)
def test_turns_and_moves(impl):
    # 'e' turn_r: 1->4, else way-1; 'q' turn_l: 4->1, else way+1 (CH:745-759); moves s,d,w,a = dir 0..3 with
    # wdx={1,0,-1,0}, wdy={0,1,0,-1} (G:742-758); walls block (showit '#', G:321-323)
    _, s = run(room_workload(), "eeqsdwa" + "w", impl)
    ways = [x.humans[0].way for x in s]
    assert ways[:4] == [1, 4, 3, 4]
    pos = [(x.humans[0].r, x.humans[0].c) for x in s]
    assert pos[4:] == [(2, 1), (2, 2), (1, 2), (1, 1), (1, 1)]  # last 'w' runs into the border wall
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_block_build_punch_and_break(impl):
    # '[' builds a destructible wall in front (G:700-714): blocks 8 -> 7.  A punch is a range-1 bullet of damage
    # max(compute_damage(100,1)=10, mindamage=100) = 100 placed on the block (CH:391-397, G:796-819); update_tmp adds
    # it to the cell's dmg and removes the block at dmg >= 1100 (G:1343-1375): 11 punches.
    w = room_workload()
    _, s = run(w, "[" + "z" * 11, impl)
    cell = 2 * 16 + 1  # (2,1): in front of (1,1) facing down
    assert s[1].humans[0].blocks == 7
    assert s[1].flags[cell] == abi.CELL_WALL | abi.CELL_TEMP
    for n in range(1, 11):
        assert s[1 + n].dmg[cell] == 100 * n, n
        assert s[1 + n].flags[cell] == abi.CELL_WALL | abi.CELL_TEMP
    assert s[12].flags[cell] == 0 and s[12].dmg[cell] == 0
    assert all(sum(b.alive for b in x.bullets) == 0 for x in s)  # every punch was absorbed in its own step
    assert s[12].humans[0].stamina == 1000000  # punches cost no stamina
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_portal_pair_teleport_and_radiation(impl):
    # ']' with portals>0 builds the exit 'O' in front and remembers its index (G:723-733); the next ']' builds the
    # entrance '^' with that index (G:716-722).  Stepping onto '^' teleports to the exit (G:517-530).  A covered exit
    # radiates: every step a bullet of damage 20 / effect -10 lands on it (G:1279-1297) and hits whoever stands there:
    # Hp -= 20, mindamage -= 10 (CH:242-246).
    script = "]" + "q" + "]" + "d" + "++"
    _, s = run(room_workload(), script, impl)
    O, UP = 2 * 16 + 1, 1 * 16 + 2
    assert s[1].flags[O] == abi.CELL_POUT | abi.CELL_TEMP
    assert (s[1].humans[0].portals, s[1].humans[0].portal_ind) == (0, 0)
    assert s[1].portals[0].active == 1 and (s[1].portals[0].r, s[1].portals[0].c) == (2, 1)
    assert s[2].humans[0].way == 2  # turn_l: 1 -> 2 (facing right, '>')
    assert s[3].flags[UP] == abi.CELL_PIN_UP | abi.CELL_TEMP and s[3].pidx[UP] == 0
    assert s[3].humans[0].portal_ind == -1
    h = s[4].humans[0]
    assert (h.r, h.c) == (2, 1), "moved right onto '^' and was teleported to the exit"
    # the teleport happens in human_action, after this step's portal_damage: first radiation hit is next step
    assert (h.hp, h.mindamage) == (1000, 100)
    assert (s[5].humans[0].hp, s[5].humans[0].mindamage) == (980, 90)
    assert (s[6].humans[0].hp, s[6].humans[0].mindamage) == (960, 80)
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_consumables(impl):
    # 'f' selects energy_drink if owned (G:759-769); 'u' uses it: stamina +20, Hp +0, mindamage +20; the count
    # drops to 0 so vec returns to -1 (CH:379-389).  'g','u': first_aid_box: Hp +200, mindamage +10.
    _, s = run(room_workload(), "fugu" + "u", impl)
    h = s[1].humans[0]
    assert (h.vec, h.ind) == (0, 0)
    h = s[2].humans[0]
    assert (h.stamina, h.hp, h.mindamage, h.cons[0], h.vec, h.ind) == (1000020, 1000, 120, 0, -1, 0)
    h = s[4].humans[0]
    assert (h.stamina, h.hp, h.mindamage, h.cons[1], h.vec, h.ind) == (1000020, 1200, 130, 0, -1, 1)
    h5 = s[5].humans[0]
    assert (h5.hp, h5.mindamage) == (1200, 130)  # 'u' with vec == -1 does nothing (CH:380)
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_gun_shot_flight_and_wall(impl):
    # 'm' selects AK_47 (weapon 4, level 1 => damage 150+50, effect -55-50, range 100, stamina -50: Items/w4.txt,
    # IT:105-111, CH:680-681).  'x' fires: stamina -50, bullet damage max(compute_damage(200,100)=1, 200+100) = 300
    # (CH:399-408) placed one cell ahead; it then advances one cell per update_bull, two per step (G:1059-1100),
    # and dies when the next cell is an indestructible wall.
    w = room_workload()
    _, s = run(w, "q" + "m" + "x" + "+" + "+", impl)  # face right: cells (1,2)..(1,5), wall at (1,6)
    h = s[3].humans[0]
    assert (h.vec, h.ind, h.stamina) == (2, 4, 1000000 - 50)
    live = [b for b in s[3].bullets if b.alive]
    # fired in human_action at (1,2); the second update_bull of the same step moved it to (1,3)
    assert len(live) == 1
    b = live[0]
    assert (b.r, b.c, b.way, b.damage, b.effect, b.range, b.owner, b.traveled, b.ref) == (1, 3, 2, 300, -105, 100, 1, 1, 1)
    live = [b for b in s[4].bullets if b.alive]
    assert len(live) == 1 and (live[0].c, live[0].traveled) == (5, 3)
    assert sum(b.alive for b in s[5].bullets) == 0  # (1,6) is '#': `else mb[_] = false` G:1092-1093
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_throwable(impl):
    # 'k' selects gas (throw0: stamina -15, damage 50, effect -20, range 100).  throw_it: damage max(50, 50+100) = 150,
    # stamina -15, count 1 -> 0 so vec = -1 (CH:410-427).  A second 'x' with vec == -1 returns before spending anything.
    _, s = run(room_workload(), "q" + "k" + "x" + "x", impl)
    h = s[3].humans[0]
    assert (h.vec, h.ind, h.throw_cnt[0], h.stamina) == (-1, 0, 0, 1000000 - 15)
    b = [x for x in s[3].bullets if x.alive][0]
    assert (b.damage, b.effect, b.range, b.owner) == (150, -20, 100, 1)
    assert s[4].humans[0].stamina == 1000000 - 15
    assert all(quiet(x) for x in s)


@pytest.mark.parametrize("impl", IMPLS)
def test_suicide_ends_episode(impl):
    # '_' sets Hp to 0 (G:696-699); hit_human of the same step clears mh (G:641-645); the next loop top's check_end
    # sees Hp <= 0 (G:1131).
    sim, s = run(room_workload(), "+" + "_", impl)
    assert s[2].humans[0].hp == 0 and s[2].humans[0].alive == 0
    assert (s[2].hdr.done, s[2].hdr.outcome) == (1, abi.DIED)
    assert sim.done()[0] == 1
    r = sim.results()[0, 0]
    assert r[5] == 0 and r[7] == abi.DIED and r[6] == s[2].hdr.frame == 5


@pytest.mark.parametrize("impl", IMPLS)
def test_timer_frame_clock(impl):
    # Timer: the episode ends when the clock runs out; kills < 5*level => lost (G:1145-1153), on the frame clock that
    # replaces time(0): frame - 1 >= level * timer_frames.
    w = room_workload(mode=abi.MODE_TIMER, timer_frames=8)
    _, s = run(w, "++++", impl)
    assert [x.hdr.done for x in s] == [0, 0, 0, 0, 1]
    assert s[4].hdr.frame == 9 and s[4].hdr.outcome == abi.TIME_LOST


@pytest.mark.parametrize("impl", IMPLS)
def test_punch_only_profile_cannot_select_weapons(impl):
    # character/human.txt owns no weapons or items: selections are refused (G:766,777,788), 'x' without a selection
    # returns before touching anything (G:808-809); blocks placement still works.
    _, s = run(room_workload(player=config.HUMAN_TOKENS), "m" + "k" + "f" + "x", impl)
    for x in s:
        h = x.humans[0]
        assert (h.vec, h.ind, h.stamina) == (-1, -1, 1000)
        assert sum(b.alive for b in x.bullets) == 0


# ---- zombies: found by searching seeds on the oracle (the search only chose the seed; every expectation below is derived
# by hand from the reference lines cited) --------------------------------------------------------------------------
def small_open_workload(seed):
    """8x8 map with an open 6x6 floor, one zombie slot, no NPC slot, no chests: the only entity besides the player is
    the zombie that the first loop top spawns (G:1446-1447: frame 1 % 40 <= 1)."""
    rows = cols = 8
    grid = [["#"] * cols for _ in range(rows)]
    for r in range(1, rows - 1):
        for c in range(1, cols - 1):
            grid[r][c] = "."
    m = "".join("".join(x) for x in grid).encode()
    cfg = config.make_config(1, rows, cols, H=1, Z=1, B=8, P=4, chests=0, auto_reset=0)
    w = config.Workload("small", cfg, m, [-1] * (rows * cols))
    w.seed = seed
    return w


@pytest.mark.parametrize("impl", IMPLS)
def test_zombie_next_to_the_player_punches_every_step_and_draws_nothing(impl):
    # seed 1700000041: the zombie walks onto (2,1), the cell in front of the player, within two steps.  From then on
    # zombie_action finds a human on a neighbour cell: it punches (a range-1 bullet of damage max(0, mindamage) = 100,
    # effect 0, on the human's cell, CH:838-844) and `continue`s before any rand() (G:664-677); hit_human applies it in
    # the same half-tick: Hp -= 100, mindamage += 0 (CH:242-246, G:611-634).
    _, s = run(small_open_workload(1700000041), "++" + "+++", impl)
    z = s[2].zombies[0]
    assert (z.alive, z.r, z.c, z.super_, z.hp, z.mindamage) == (1, 2, 1, 0, 400, 100)
    assert s[2].humans[0].hp == 1000
    for k in (3, 4, 5):
        h, z = s[k].humans[0], s[k].zombies[0]
        assert (h.hp, h.mindamage) == (1000 - 100 * (k - 2), 100)
        assert (z.r, z.c, z.hp) == (2, 1, 400)
        assert sum(b.alive for b in s[k].bullets) == 0  # the punch was consumed by the hit
        # draws of such a step: update_bull twice (G:1073), human_action's sweep direction once (G:1002) = 3, plus
        # three coordinates per spawn attempt whose frame is due (G:532-572; both attempts end at the full slot pools)
        f = s[k - 1].hdr.frame  # the frame at the loop top that preceded this step was counted in the step before
        spawn = 3 * ((s[k].hdr.frame % 40 <= 1) + (s[k].hdr.frame % 50 <= 1))
        assert s[k].hdr.jomle - s[k - 1].hdr.jomle == 3 + spawn, (k, f)


@pytest.mark.parametrize("impl", IMPLS)
def test_player_punches_the_zombie_dead_and_is_credited(impl):
    # same seed; the player (facing down, G:1905-1920) punches the zombie on (2,1): max(compute_damage(100,1) = 10,
    # mindamage 100) = 100 per punch (CH:391-397); zombie_damage credits the owner's damage and, on death, kills;
    # owner == hum[ind] on the player's team: teams_kills + 1, loot += 500/10 + 500*9/10, kills + 1 (G:574-598).
    # The zombie keeps punching until it dies: four exchanges.
    _, s = run(small_open_workload(1700000041), "++" + "zzzz" + "+", impl)
    for k in range(1, 5):
        h, z = s[2 + k].humans[0], s[2 + k].zombies[0]
        assert h.hp == 1000 - 100 * k and h.damage == 100 * k
        assert (z.alive, z.hp) == ((1, 400 - 100 * k) if k < 4 else (0, z.hp))
    d = s[6]
    assert (d.hdr.kills, d.hdr.teams_kills, d.hdr.loot, d.humans[0].kills) == (1, 1, 500, 1)
    assert s[7].humans[0].hp == 600 and s[7].hdr.done == 0  # nobody left to punch; Solo level 1 needs 5 kills (G:1147-1160)


@pytest.mark.parametrize("impl", IMPLS)
def test_chest_pickup(impl):
    # seed 1700000095: the first loop top (frame 1, G:1444-1445) drops a chest of type 2 (Items/cons2.txt: stamina 20,
    # Hp 50, effect 10) on (1,2) and no zombie.  'd' moves the player onto it (showit '?' is enterable, G:750-756) and
    # claim_chest (G:507-515, CH:372-377) adds stamina / Hp / effect-to-mindamage, clears s[4] and decrements `chest`.
    w = small_open_workload(1700000095)
    w.cfg.cap_chests = 5
    _, s = run(w, "d" + "+", impl)
    cell = 1 * 8 + 2
    assert s[0].flags[cell] == abi.CELL_CHEST | (2 << abi.CELL_CONS_SHIFT) and s[0].hdr.chests == 1
    assert sum(z.alive for z in s[0].zombies) == 0
    h = s[1].humans[0]
    assert (h.r, h.c) == (1, 2)
    assert (h.stamina, h.hp, h.mindamage) == (1000000 + 20, 1000 + 50, 100 + 10)
    assert s[1].flags[cell] == 0 and s[1].hdr.chests == 0
    assert (s[2].humans[0].stamina, s[2].humans[0].hp) == (1000020, 1050)  # claimed once


# ---------------------------------------------------------------------------------------------------------------------
# Two players: whose bullet, whose team, whose credit (human_damage G:611-634, Character::hit CH:242-246)
# ---------------------------------------------------------------------------------------------------------------------
def corridor_workload(teams, seed=1700000000):
    """Battle start (random cells and facings, G:1846-1858) in a one-row corridor, so that the three players stand in line."""
    rows = cols = 16
    grid = [["#"] * cols for _ in range(rows)]
    for c in range(1, 8):
        grid[1][c] = "."
    m = "".join("".join(row) for row in grid).encode()
    cfg = config.make_config(1, rows, cols, H=4, Z=2, B=8, P=4, mode=abi.MODE_BATTLE, n_agents=3, teams=teams, auto_reset=0)
    w = config.Workload("corridor", cfg, m, [-1] * (rows * cols))
    w.seed = seed
    return w


def _duel(impl, friendly):
    # where the players start does not depend on their teams (the placement draws come first): look, then pick the teams
    probe = impl(corridor_workload([1, 2, 3]))
    probe.reset(*probe.w.seeds(base_tb=probe.w.seed))
    h = probe.dump(0).humans
    cols = [h[i].c for i in range(3)]
    assert all(h[i].alive and h[i].r == 1 for i in range(3)) and len(set(cols)) == 3
    others = sorted((abs(cols[i] - cols[0]), i) for i in (1, 2))
    tgt = others[0][1]                       # the nearer one: nobody stands between it and player 0
    third = 3 - tgt
    teams = [1, 0, 0]
    teams[tgt], teams[third] = (1, 2) if friendly else (2, 3)
    w = corridor_workload(teams)
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=w.seed))
    d0 = sim.dump(0)
    assert [d0.humans[i].c for i in range(3)] == cols
    right = cols[tgt] > cols[0]
    want_way = 2 if right else 4             # way - 1 indexes wdx/wdy = {1,0,-1,0}/{0,1,0,-1} (G:742-758): 2 = right, 4 = left
    script = "q" * ((want_way - d0.humans[0].way) % 4)             # turn_l: 4 -> 1, else way + 1 (CH:745-759)
    script += ("d" if right else "a") * (abs(cols[tgt] - cols[0]) - 1)  # walk up to the cell next to the target
    script += "m"                                                  # AK_47, as in test_gun_shot_flight_and_wall
    snaps = []
    for ch in script + "xxxx":
        sim.step(np.array([ord(ch), ord("+"), ord("+")], dtype=np.uint8))
        snaps.append(sim.dump(0))
    assert len(script) + 4 < 19  # before the first periodic spawn (frame 20) could put anyone else into the corridor
    return sim, snaps[-4:], tgt, third


@pytest.mark.parametrize("impl", IMPLS)
def test_shooting_a_rival_dead(impl):
    # Each shot is put on the neighbour's cell (G:796-819) and lands in the same step's hit_human: Hp -= 300,
    # mindamage += -105 (CH:242-246; the bullet of test_gun_shot_flight_and_wall).  The shooter is of another team, so its
    # own damage / effect counters grow by the bullet's (G:615-618), and when the fourth shot takes Hp to -200:
    # mh = false, ++teams_kills, loot += 100, and because the owner is the player itself loot += 900, ++kills (G:623-627);
    # increase_kills() on the owner (G:628-629).  A player of a third team is still alive: no end of the match.
    sim, s, tgt, third = _duel(impl, friendly=False)
    for n, d in enumerate(s, start=1):
        v, me = d.humans[tgt], d.humans[0]
        assert (v.hp, v.mindamage) == (1000 - 300 * n, 100 - 105 * n)
        assert (me.damage, me.effect, me.stamina, me.mindamage) == (300 * n, -105 * n, 1000000 - 50 * n, 100)
        assert sum(b.alive for b in d.bullets) == 0
        assert v.alive == (1 if n < 4 else 0)
        assert (me.kills, d.hdr.kills, d.hdr.teams_kills, d.hdr.loot) == ((0, 0, 0, 0) if n < 4 else (1, 1, 1, 1000))
    assert s[-1].humans[third].alive == 1 and (s[-1].hdr.done, s[-1].hdr.outcome) == (0, abi.RUNNING)


@pytest.mark.parametrize("impl", IMPLS)
def test_shooting_a_team_mate_dead(impl):
    # The same four shots at a player of the shooter's own team: the victim is hit all the same (CH:242-246 knows no
    # teams), but none of it is credited — no damage / effect (G:615), no teams_kills / loot / kills (the dead player is of
    # hum[ind]'s team, G:623), no increase_kills (G:628).
    sim, s, tgt, third = _duel(impl, friendly=True)
    for n, d in enumerate(s, start=1):
        v, me = d.humans[tgt], d.humans[0]
        assert (v.hp, v.mindamage, v.alive) == (1000 - 300 * n, 100 - 105 * n, 1 if n < 4 else 0)
        assert (me.damage, me.effect, me.kills, me.stamina) == (0, 0, 0, 1000000 - 50 * n)
        assert (d.hdr.kills, d.hdr.teams_kills, d.hdr.loot) == (0, 0, 0)
    assert s[-1].humans[third].alive == 1 and (s[-1].hdr.done, s[-1].hdr.outcome) == (0, abi.RUNNING)
