"""Hand-derived known answers for the observation encoder, describe() + gameplay::bot()
(bots/bot-0.5/Custom.hpp:29-159, file under /root/reference/StrikeForce-client): the only evidence for SURVEY §8 row
a20 that does not come from this repo's own restatement.  Every expected 32-vector below is written out by hand from
the cited lines (values BEFORE the final map); the map itself, `pow(abs(x) / 10, 0.2)` with the division in float and
the power in double (Custom.hpp:157), is applied here with numpy.  Checked on the oracle, on the device source run on
the wave emulator and (-m gpu) on the device, within 1 float ulp (ocml pow vs libm).

Channel order (Custom.hpp:31-134):
   0 character   1 bullet   2 wall   3 chest   4 portal entrance   5 portal exit   6 temporary (player-built)
   7 team-mate   8 enemy    9 npc (team 0)   10 zombie    11 kills   12 blocks   13 portals   14 portal pending
  15 stops humans   16 stops bullets   17 destructible   18 Hp / 1000
  19 is a bullet   20-23 attack vector (V > A <)   24 damage / 1000   25 -effect / 1000   26 stamina / 1000
  27-29 chest: stamina, effect, Hp (/ 1000)      30 damage dealt / 1000   31 -effect dealt / 1000
"""
import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import abi, config

R = 15  # the observer sits at window cell (15, 15) (Custom.hpp:143-145)


def vec(**kw):
    v = [0.0] * 32
    for k, x in kw.items():
        v[int(k[1:])] = x
    return v


def mapped(v):
    x = np.abs(np.asarray(v, dtype=np.float32))
    q = (x / np.float32(10)).astype(np.float32)  # float / int -> float (Custom.hpp:157)
    return np.power(q.astype(np.float64), 0.2).astype(np.float32)  # std::pow(float, double) -> double, stored as float


def cell_of(obs, arena, agent, pr, pc, r, c):
    """the 32 channels of map cell (r, c) in the window of an observer standing at (pr, pc)"""
    return obs[arena, agent, :, R + r - pr, R + c - pc]


def check(got, want_raw, what):
    want = mapped(want_raw)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, "%s: channels %s differ: got %s want %s" % (
        what, np.nonzero(ulp > 1)[0].tolist(), got[ulp > 1].tolist(), want[ulp > 1].tolist())
    assert ((got == 0) == (want == 0)).all(), what  # the zero pattern is exact


# ---- expected vectors, by kind -----------------------------------------------------------------------------------
EMPTY = vec()                                   # '.', and the default node used outside the map (Custom.hpp:147-148)
WALL = vec(c2=1, c15=1, c16=1)                   # s[3]: stops humans and bullets, not destructible (Custom.hpp:68-70)
# fresh zombie: gen_npc CH:850-857 Hp 400 / mindamage 100 (super: 800 / 200); attack vector 0.01 each (Custom.hpp:98-100)
ZOMBIE = vec(c0=1, c10=1, c15=1, c16=1, c17=1, c18=0.4, c20=0.01, c21=0.01, c22=0.01, c23=0.01, c24=0.1)
SUPER = vec(c0=1, c10=1, c15=1, c16=1, c17=1, c18=0.8, c20=0.01, c21=0.01, c22=0.01, c23=0.01, c24=0.2)
# fresh human of character/human_enemy.txt (Hp 1000, mindamage 100, stamina 1e6, 8 blocks, 1 portal, CH:78-79,650-709),
# way 1, nothing selected: damage = max(compute_damage(100, 1) = 10, mindamage = 100) = 100, effect 0 (CH:429-443)
def human(team_channel, way=1, **kw):
    v = vec(c0=1, c12=8, c13=1, c15=1, c16=1, c17=1, c18=1.0, c24=0.1, c26=1000.0)
    v[team_channel] = 1
    v[19 + way] = 1  # sit[way - 1] -> channel 20 + way - 1
    for k, x in kw.items():
        v[int(k[1:])] = x
    return v
# chests: gen_item(type) -> Items/cons<type>.txt, channels stamina, effect, Hp (Custom.hpp:116-123)
CHEST = [vec(c3=1, c27=0.02, c28=0.02, c29=0.0), vec(c3=1, c27=0.0, c28=0.01, c29=0.2),
         vec(c3=1, c27=0.02, c28=0.01, c29=0.05), vec(c3=1, c27=0.02, c28=0.02, c29=0.4)]


def open_workload(arenas, mode=abi.MODE_SOLO, n_agents=1, teams=None, device=0):
    rows = cols = 14
    grid = [["#"] * cols for _ in range(rows)]
    for r in range(1, rows - 1):
        for c in range(1, cols - 1):
            grid[r][c] = "."
    m = "".join("".join(x) for x in grid).encode()
    cfg = config.make_config(arenas, rows, cols, H=4, Z=4, B=8, P=4, mode=mode, auto_reset=0, n_agents=n_agents,
                             teams=teams, device=device)
    return config.Workload("open", cfg, m, [-1] * (rows * cols))


def room_workload(device=0):
    rows = cols = 16
    grid = [["#"] * cols for _ in range(rows)]
    for r in range(1, 4):
        for c in range(1, 6):
            grid[r][c] = "."
    m = "".join("".join(row) for row in grid).encode()
    cfg = config.make_config(1, rows, cols, H=2, Z=2, B=8, P=4, auto_reset=0, device=device)
    return config.Workload("room", cfg, m, [-1] * (rows * cols))


def spawns_after_reset(impl):
    """The first loop top (frame 1) tries a chest, a zombie and an NPC human at random cells (G:1444-1449): 40 seeds of
    an open 12x12 floor give every kind.  Which kind stands where is read from the state dump; what the encoder must
    say about that kind is the literal vector above."""
    A = 40
    w = open_workload(A)
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=1700000000))
    obs = sim.observe()
    seen = set()
    for a in range(A):
        d = sim.dump(a)
        me = d.humans[0]
        assert (me.r, me.c, me.way, me.team) == (1, 1, 1, 1)                    # G:1905-1920
        check(cell_of(obs, a, 0, 1, 1, 1, 1), human(7), "the observer's own cell")
        check(cell_of(obs, a, 0, 1, 1, 0, 0), WALL, "border wall")
        check(cell_of(obs, a, 0, 1, 1, -3, 5), EMPTY, "outside the map")
        for z in d.zombies:
            if z.alive:
                check(cell_of(obs, a, 0, 1, 1, z.r, z.c), SUPER if z.super_ else ZOMBIE, "zombie")
                seen.add("super" if z.super_ else "zombie")
        for h in d.humans[1:]:
            if h.alive:  # gen_human: team 0, way 1 (CH:873-888)
                assert (h.team, h.way) == (0, 1)
                check(cell_of(obs, a, 0, 1, 1, h.r, h.c), human(9), "npc human")
                seen.add("npc")
        for ci, f in enumerate(d.flags.tolist()):
            if f & abi.CELL_CHEST:
                t = (f >> abi.CELL_CONS_SHIFT) & 3
                check(cell_of(obs, a, 0, 1, 1, ci // 14, ci % 14), CHEST[t], "chest type %d" % t)
                seen.add("chest%d" % t)
        occupied = {(z.r, z.c) for z in d.zombies if z.alive} | {(h.r, h.c) for h in d.humans if h.alive}
        for r, c in ((5, 5), (6, 7), (9, 3)):
            if (r, c) not in occupied and d.flags[r * 14 + c] == 0:
                check(cell_of(obs, a, 0, 1, 1, r, c), EMPTY, "empty floor")
    assert seen == {"zombie", "super", "npc", "chest0", "chest1", "chest2", "chest3"}, seen


def built_objects(impl):
    """Player-built block with accumulated damage, portal exit, selected gun, pending portal (the sealed 3x5 room of
    tests/test_oracle_scenarios.py; the player starts at (1,1) facing down)."""
    w = room_workload()
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=1700000000))
    for ch in "q[zzze]c":
        sim.step(np.array([ord(ch)], dtype=np.uint8))
    d = sim.dump(0)
    me = d.humans[0]
    assert sum(z.alive for z in d.zombies) == 0 and sum(h.alive for h in d.humans[1:]) == 0 and d.hdr.chests == 0
    # 'q': way 1 -> 2; '[': block on (1,2), 7 blocks left (G:700-714); 3 punches of max(10, 100) = 100 each: dmg 300
    # (CH:391-397, G:1343-1375); 'e': way 2 -> 1; ']': exit on (2,1), portals 1 -> 0, portal_ind = 0 (G:723-733);
    # 'c': push_dagger selected (vec 2, ind 0; G:781-791)
    assert (me.r, me.c, me.way, me.blocks, me.portals, me.portal_ind, me.vec, me.ind) == (1, 1, 1, 7, 0, 0, 2, 0)
    assert d.dmg[1 * 16 + 2] == 300 and d.flags[2 * 16 + 1] == abi.CELL_POUT | abi.CELL_TEMP
    obs = sim.observe()
    # own cell: the level-1 push_dagger is upgraded once by Human::build (CH:680-681; w0.txt damage 150, effect -50,
    # stamina -25, range 1 -> 200 / -100): get_damage_effect CH:436-440 = max(compute_damage(200, 1) = 14, 200 + 100,
    # 100) = 300 and effect -100; stamina 1e6 allows the shot
    check(cell_of(obs, 0, 0, 1, 1, 1, 1), human(7, way=1, c12=7, c13=0, c14=1, c24=0.3, c25=0.1), "own cell, gun selected")
    # destructible wall: Hp = (lim_block 1100 - dmg 300) / 1000 (Custom.hpp:75-77)
    check(cell_of(obs, 0, 0, 1, 1, 1, 2), vec(c2=1, c6=1, c15=1, c16=1, c17=1, c18=0.8), "player-built block")
    # uncovered exit: s[7] and s[10]; stops humans only; radiates 20 / -10 (Custom.hpp:81-84,108-111)
    check(cell_of(obs, 0, 0, 1, 1, 2, 1), vec(c5=1, c6=1, c15=1, c24=0.02, c25=0.01), "portal exit")
    check(cell_of(obs, 0, 0, 1, 1, 3, 3), EMPTY, "floor of the room")
    check(cell_of(obs, 0, 0, 1, 1, 4, 1), WALL, "solid rock")


def portal_entrance(impl):
    w = room_workload()
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=1700000000))
    for ch in "]q]":
        sim.step(np.array([ord(ch)], dtype=np.uint8))
    d = sim.dump(0)
    me = d.humans[0]
    # exit on (2,1), then the entrance '^' on (1,2) bound to it; nothing pending any more (G:716-733)
    assert (me.way, me.portals, me.portal_ind) == (2, 0, -1)
    assert d.flags[1 * 16 + 2] == abi.CELL_PIN_UP | abi.CELL_TEMP and d.pidx[1 * 16 + 2] == 0
    obs = sim.observe()
    check(cell_of(obs, 0, 0, 1, 1, 1, 1), human(7, way=2, c13=0, c14=0), "own cell after both halves of the portal")
    # player-built entrance: Hp = (lim_portal 1000 - 0) / 1000 (Custom.hpp:78-79)
    check(cell_of(obs, 0, 0, 1, 1, 1, 2), vec(c4=1, c6=1, c15=1, c16=1, c17=1, c18=1.0), "portal entrance")
    check(cell_of(obs, 0, 0, 1, 1, 2, 1), vec(c5=1, c6=1, c15=1, c24=0.02, c25=0.01), "portal exit")


def bullet_in_flight(impl):
    w = room_workload()
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=1700000000))
    for ch in "qkx":
        sim.step(np.array([ord(ch)], dtype=np.uint8))
    d = sim.dump(0)
    me = d.humans[0]
    b = [x for x in d.bullets if x.alive]
    # 'k' selects gas (throw0.txt: stamina -15, damage 50, effect -20, range 100); 'x' throws it: damage
    # max(50, 50 + 100) = 150 (CH:410-427), placed on (1,2), advanced once by this step's last update_bull (G:1471)
    assert len(b) == 1 and (b[0].r, b[0].c, b[0].way, b[0].traveled, b[0].damage, b[0].effect, b[0].range) == (1, 3, 2, 1, 150, -20, 100)
    assert (me.stamina, me.vec, list(me.throw_cnt)) == (1000000 - 15, -1, [0, 1, 1, 1])
    obs = sim.observe()
    # remaining range (100 - 1) / 100 on the bullet's own direction (Custom.hpp:102-107)
    check(cell_of(obs, 0, 0, 1, 1, 1, 3), vec(c1=1, c19=1, c21=0.99, c24=0.15, c25=0.02), "bullet in flight")
    check(cell_of(obs, 0, 0, 1, 1, 1, 1), human(7, way=2, c26=999.985), "thrower: last gas gone, nothing selected")


def other_players(impl):
    """Battle match, three players on teams 1, 2, 1: the first sees an enemy and a team-mate (Custom.hpp:40-48); all are
    built from the same record and placed by _rand() with a random facing (G:1846-1859)."""
    w = open_workload(1, mode=abi.MODE_BATTLE, n_agents=3, teams=[1, 2, 1])
    sim = impl(w)
    sim.reset(*w.seeds(base_tb=1700000005))
    d = sim.dump(0)
    obs = sim.observe()
    hs = d.humans
    assert [h.team for h in hs[:3]] == [1, 2, 1] and all(h.alive for h in hs[:3])
    for viewer in range(3):
        v = hs[viewer]
        for other in range(3):
            o = hs[other]
            ch = 7 if o.team == v.team else 8
            check(cell_of(obs, 0, viewer, v.r, v.c, o.r, o.c), human(ch, way=o.way), "player %d seen by %d" % (other, viewer))


SCENARIOS = [spawns_after_reset, built_objects, portal_entrance, bullet_in_flight, other_players]


@pytest.mark.parametrize("scenario", SCENARIOS, ids=lambda f: f.__name__)
@pytest.mark.parametrize("impl", [Oracle, Emu], ids=["oracle", "emu"])
def test_describe_known_answers(impl, scenario):
    scenario(impl)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", SCENARIOS, ids=lambda f: f.__name__)
def test_describe_known_answers_on_the_device(scenario):
    from strikeforce_amd import env
    scenario(env.ArenaBatch)
