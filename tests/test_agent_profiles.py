"""One character record per commanded human (ABI 2, sf_config.agent_profile): the account blobs the players of a
lock-step match exchange before the first tick (give_info / get_info gameplay.hpp:120-151; Human::log_file / scan_file
Character.hpp:570-648; Human::build Character.hpp:650-709).  Expected stats below are derived by hand from the records
and the cited lines."""
import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import Oracle, diff_dumps
from strikeforce_amd import abi, config


def Device(w):
    """The HIP path through the C-ABI: only under `-m gpu`."""
    from strikeforce_amd import env
    return env.ArenaBatch(w)


IMPLS = [pytest.param(Oracle, id="oracle"), pytest.param(Emu, id="emu"), pytest.param(Device, id="device", marks=pytest.mark.gpu)]


def _fresh(impl):
    w = config.baseline_workload("KITS", arenas=2)
    sim = impl(w)
    sim.reset(*w.seeds())
    return w, sim


@pytest.mark.parametrize("impl", IMPLS)
def test_every_player_is_built_from_its_own_record(impl):
    _, sim = _fresh(impl)
    h = sim.dump(0).humans
    # character/human.txt: Hp 1000, mindamage 100, stamina 1000, nothing owned but the four level-1 throwables at 0 pcs
    assert (h[0].hp, h[0].mindamage, h[0].stamina, list(h[0].cons), list(h[0].throw_cnt)) == (1000, 100, 1000, [0, 0, 0, 0], [0, 0, 0, 0])
    # character/human_enemy.txt: stamina 1e6, one of everything
    assert (h[1].hp, h[1].mindamage, h[1].stamina, list(h[1].cons), list(h[1].throw_cnt)) == (1000, 100, 1000000, [1, 1, 1, 1], [1, 1, 1, 1])
    # the levelled-up account: Hp 15000, mindamage 1000, stamina 15000 (Character.hpp:667: def_* as they stand in the
    # record); levels 10/10/10 add a block and a portal at every odd level 3..9, three times: 8 + 12, 1 + 12
    assert (h[2].hp, h[2].mindamage, h[2].stamina, h[2].blocks, h[2].portals) == (15000, 1000, 15000, 20, 13)
    assert (h[1].blocks, h[1].portals) == (8, 1)
    # players 4 and 5 repeat records 0 and 1 (six players, four records)
    assert (h[4].stamina, h[5].stamina) == (1000, 1000000)
    assert all(x.profile == 0 for x in h[:6])


@pytest.mark.parametrize("impl", IMPLS)
def test_a_shot_carries_the_shooters_own_weapon_level(impl):
    """Weapon 4 (w4.txt: damage 150, effect -55, range 100) at level L is upgraded L times by Human::build
    (Character.hpp:680-681, Weapon::upgrade Item.hpp:105-111: +50 / -50 per level).  shot_it (Character.hpp:399-408):
    damage = max(compute_damage(d, 100), d + mindamage).  Player 1 (human_enemy, level 1): d = 200, +100 = 300,
    effect -105.  Player 3 (level-3 guns): d = 300, +100 = 400, effect -205.  Player 0 (human.txt) owns weapon 4 at
    level 1 too (record: 1 0 1 0 1 0 1 0) but has stamina 1000: the shot costs 50."""
    _, sim = _fresh(impl)
    n = 2 * 6  # two arenas
    sel = np.full(n, ord("m"), dtype=np.uint8)  # 'm' selects weapon index 4 (gameplay.hpp:781-791)
    sim.step(sel)
    fire = np.full(n, ord("x"), dtype=np.uint8)
    sim.step(fire)
    d = sim.dump(0)
    by_owner = {b.owner - 1: b for b in d.bullets if b.alive}
    assert (by_owner[1].damage, by_owner[1].effect, by_owner[1].range) == (300, -105, 100)
    assert (by_owner[3].damage, by_owner[3].effect, by_owner[3].range) == (400, -205, 100)
    assert d.humans[0].stamina in (950, 1000)  # 950 if the cell in front let the bullet be allocated at all
    assert d.humans[3].stamina == 1000000 - 50


def test_lockstep_parity_with_different_kits():
    w = config.baseline_workload("KITS", arenas=3)
    o, e = Oracle(w), Emu(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), e.reset(tb, sr)
    cmds, _ = config.bench_commands(3, w.cfg.n_agents, 160)
    for s in range(160):
        o.step(cmds[s]), e.step(cmds[s])
        if s % 8 == 0 or s == 159:
            for a in range(3):
                dd = diff_dumps(o.dump(a).as_dict(), e.dump(a).as_dict())
                assert dd is None, "step %d arena %d: %s" % (s, a, dd)
    x, y = o.observe(), e.observe()
    assert np.array_equal(x.view(np.uint32), y.view(np.uint32))  # get_damage_effect reads each human's own record
    assert (o.digest() == e.digest()).all()


def test_agent_profiles_must_cover_every_agent():
    w = config.baseline_workload("KITS", arenas=1)
    w.cfg.n_agent_profiles = 3  # neither 0 nor n_agents
    with pytest.raises(Exception):
        Emu(w)


@pytest.mark.gpu
def test_different_kits_on_the_device():
    from strikeforce_amd import env
    w = config.baseline_workload("KITS", arenas=4)
    o, g = Oracle(w), env.ArenaBatch(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), g.reset(tb, sr)
    cmds, _ = config.bench_commands(4, w.cfg.n_agents, 300)
    from oracle_lib import ArenaDump
    for s in range(300):
        o.step(cmds[s]), g.step(cmds[s])
        if s % 25 == 0 or s == 299:
            for a in range(4):
                dd = diff_dumps(o.dump(a).as_dict(), ArenaDump(*g.dump_raw(a)).as_dict())
                assert dd is None, "step %d arena %d: %s" % (s, a, dd)
    x, y = o.observe(), g.observe()
    ulp = np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64)).max()
    assert ulp <= 1 and ((x == 0) == (y == 0)).all()
    assert (o.digest() == g.digest()).all() and (o.results() == g.results()).all()
