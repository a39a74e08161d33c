"""sf_step_begin / sf_step_end (the iteration cut where the reference queries the agents of humans other than `ind`,
gameplay.hpp:988-999) and sf_agent_alive (Human::active_agent, deleteAgent gameplay.hpp:648-649): on the oracle, on the
emulated device core and (-m gpu) on the device; and against the reference itself built with its own switch
USE_AGENT_IN_SQUAD_NPCS (oracle/ref_tick.py, Squad mode: ten Agents in one process), which is the one configuration of
the reference in which an agent other than the player's exists."""
import ctypes as C

import numpy as np
import pytest

import ref_cases
import reftick
from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import abi, config


def Device(w):
    from strikeforce_amd import env
    return env.ArenaBatch(w)


IMPLS = [pytest.param(Oracle, id="oracle"), pytest.param(Emu, id="emu"), pytest.param(Device, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("name,steps", [("C3", 200), ("FLOORS", 200), ("C5", 60), ("STRESS", 300), ("C4", 80)])
def test_two_halves_equal_one_step(impl, name, steps):
    """Same commands through sf_step and through sf_step_begin + sf_step_end: identical digests after every step
    (auto-reset on: restarts happen inside the second half), identical done counts and agent liveness."""
    A = 3
    w1, w2 = config.baseline_workload(name, arenas=A), config.baseline_workload(name, arenas=A)
    a, b = impl(w1), impl(w2)
    a.reset(*w1.seeds()), b.reset(*w2.seeds())
    cmds, _ = config.bench_commands(A, w1.cfg.n_agents, steps)
    for s in range(steps):
        a.step(cmds[s])
        b.step_begin()
        if s % 16 == 0:
            b.observe()  # observing in the middle changes nothing
        b.step_end(cmds[s])
        assert (a.digest() == b.digest()).all(), "%s step %d" % (name, s)
        assert (a.done() == b.done()).all()
        assert (a.agent_alive() == b.agent_alive()).all()


@pytest.mark.gpu
def test_misuse_is_refused_on_the_device():
    from strikeforce_amd import env
    w = config.baseline_workload("C1", arenas=2)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    cmd = np.full(2, ord("+"), dtype=np.uint8)
    with pytest.raises(env.StrikeForceError):
        g.step_end(cmd)          # no begin
    g.step_begin()
    with pytest.raises(env.StrikeForceError):
        g.step_begin()           # twice
    with pytest.raises(env.StrikeForceError):
        g.step(cmd)              # a whole step in the middle of one
    g.step_end(cmd)
    g.step(cmd)


needs_ref = pytest.mark.skipif(not reftick.available(), reason="oracle/_ref/sf_ref_tick not built (no reference checkout)")


@needs_ref
@pytest.mark.parametrize("impl", IMPLS)
def test_squad_agents_observe_and_die_as_in_the_reference(impl):
    """The reference with USE_AGENT_IN_SQUAD_NPCS (ten scripted Agents: the player's + nine Squad humans') beside this
    repo with n_agents = 10.  Every step: the observation the reference hands the player's agent at the loop top equals
    sf_observe before the step; the observations it hands agents 1-9 inside human_action equal sf_observe between
    sf_step_begin and sf_step_end (bit-exact on the CPU, 1 ulp on the device); the whole state agrees after the step; the
    set of humans that still have an Agent (deleteAgent on death; the slot later re-used by an NPC) equals
    sf_agent_alive; and the reference's call sequence per step is predict(0), update(0), then predict(i), update(i) in
    slot order for the living agents (gameplay.hpp:970-999)."""
    ulp_tol = 0 if impl in (Oracle, Emu) else 1
    w = ref_cases.native(abi.MODE_SQUAD, 2, ref_cases.RICH, maps="shipped")
    w.cfg.n_agents = 10
    sim = impl(w)
    r = reftick.RefTick(w, ref_cases.RICH, agents=True, squad_agents=True)
    tb, serial = 1700000000, 123456789
    sim.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(serial))
    r.reset(tb, serial)
    assert [c[1] for c in r.calls()] == ["N"] * 10  # prepare(): ten Agents (gameplay.hpp:1744,1883-1885,1896-1898)
    rng = np.random.RandomState(1)
    acts = "+xzqeawsd"  # the action string of Custom.hpp:162
    deaths = 0

    def ulps(a, b):
        return int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max())

    for s in range(320):
        chars = "".join(acts[i] for i in rng.randint(0, 9, size=10))
        top = sim.observe()[0]
        sim.step_begin()
        mid = sim.observe()[0]
        # who is asked: the player at the loop top; the others if they are still alive when human_action runs (one shot
        # dead in the first half-tick has lost its Agent by then, gameplay.hpp:648-649)
        alive_before = sim.agent_alive()[0]
        alive_before[0] = 1
        sim.step_end(np.frombuffer(chars.encode(), dtype=np.uint8))
        r.step(chars)
        calls = r.calls()
        want = []
        for g in range(10):
            if alive_before[g]:
                want += [(g, "P"), (g, "U")]
        got = [(c[0], c[1]) for c in calls if c[1] in "PU"]
        assert got == want, "step %d: calls %s" % (s, got)
        assert [c[2] for c in calls if c[1] == "U"] == [acts.index(chars[g]) for g in range(10) if alive_before[g]]
        for g in range(10):
            if alive_before[g]:
                ref_obs = r.last_obs(g)
                ours = (top if g == 0 else mid)[g].reshape(-1)
                assert ulps(ref_obs, ours) <= ulp_tol, "step %d agent %d" % (s, g)
        d = reftick.first_difference(r.dump(), reftick.arrays_of(sim.dump(0) if impl is not Device else _gpu_dump(sim)))
        assert d is None, "step %d: %s" % (s, d)
        assert list(sim.agent_alive()[0]) == r.active_agents[:10], "step %d" % s
        # check_end's helper rivals_are_dead() (gameplay.hpp:497-505), and check_end itself (the Squad game goes on):
        # no living human of a team other than 0 and the player's
        dd = sim.dump(0) if impl is not Device else _gpu_dump(sim)
        mine = dd.humans[0].team
        assert r.rivals_are_dead() == (not any(h.alive and h.team not in (0, mine) for h in dd.humans)), "step %d" % s
        assert r.ended == bool(sim.done()[0]), "step %d: the reference's check_end() says %s" % (s, r.ended)
        deaths += sum(1 for c in calls if c[1] == "D")
        if sim.done()[0]:
            break
    assert deaths >= 3 and s >= 250  # agents did die (and their slots were re-used) within the run
    r.close()


def _gpu_dump(g):
    from oracle_lib import ArenaDump
    return ArenaDump(*g.dump_raw(0))


@pytest.mark.gpu
def test_device_resident_forms_of_the_new_entry_points():
    """sf_step_end_device (commands already in HBM) and sf_agent_alive_device against their host forms."""
    import torch
    from strikeforce_amd import env
    A = 8
    w1, w2 = config.baseline_workload("C5", arenas=A), config.baseline_workload("C5", arenas=A)
    a, b = env.ArenaBatch(w1), env.ArenaBatch(w2)
    a.reset(*w1.seeds()), b.reset(*w2.seeds())
    n = w1.cfg.n_agents
    cmds, _ = config.bench_commands(A, n, 120)
    d = torch.from_numpy(cmds).cuda()
    alive = torch.zeros(A * n, dtype=torch.uint8, device="cuda")
    for s in range(120):
        a.step(cmds[s])
        b.step_begin()
        b.step_end_device(d.data_ptr() + s * A * n)
        if s % 10 == 0:
            b.agent_alive_device(alive.data_ptr())
            b.synchronize()
            assert (alive.cpu().numpy().reshape(A, n) == a.agent_alive()).all()
    assert (a.digest() == b.digest()).all()
