"""tests/golden/ref_traj.json = per-step state digests and observations of the REFERENCE's own tick path
(oracle/_ref/sf_ref_tick: gameplay.hpp / Character.hpp / Item.hpp / random.hpp / bot-0.5's Custom.hpp compiled head-less,
oracle/ref_tick.py; generator tests/golden/make_ref_traj.py).  The oracle, the device core on the CPU wave emulator and
(-m gpu) the HIP path through the C-ABI must reproduce every digest of every step — every state word including the
generator's registers and draw count — and the observations (CPU: bit-exact; device: within 1 ulp of the reference's
libm pow).  Runs anywhere: the reference itself is not needed, its outputs are the fixture."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ref_cases
from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import config

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_traj.json")) as _f:
    GOLD = json.load(_f)
assert "sf_ref_tick" in GOLD["_generator"]


def Device(w):
    from strikeforce_amd import env
    return env.ArenaBatch(w)


IMPLS = [pytest.param(Oracle, 0, id="oracle"), pytest.param(Emu, 0, id="emu"),
         pytest.param(Device, 1, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("impl,ulp_tol", IMPLS)
@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_reproduces_the_references_trajectory(name, impl, ulp_tol):
    gold = GOLD["cases"][name]
    make = ref_cases.GOLDEN_CASES[name][0]
    w = make()
    sim = impl(w)
    sim.reset((C.c_uint64 * 1)(gold["tb"]), (C.c_uint64 * 1)(gold["serial"]))
    digests = gold["digests"]
    steps = len(digests) - 1
    if gold.get("commands"):  # a scripted game (the Squad game played to its end)
        assert steps >= 200 and len(gold["commands"]) >= steps
        cmds = np.frombuffer(gold["commands"].encode(), dtype=np.uint8).reshape(-1, 1, 1)
    else:
        assert steps >= 400
        cmds, _ = config.bench_commands(1, 1, steps, seed0=gold["command_seed"])
    for s in range(steps + 1):
        assert "%016x" % int(sim.digest()[0]) == digests[s], "%s: state after %d steps differs from the reference's" % (name, s)
        if str(s) in gold["obs_nonzero"]:
            want = np.zeros(32 * 31 * 31, dtype=np.uint32)
            for t in gold["obs_nonzero"][str(s)].split():
                i, v = t.split(":")
                want[int(i, 16)] = int(v, 16)
            got = np.ascontiguousarray(sim.observe()[0, 0]).reshape(-1).view(np.int32).astype(np.int64)
            ulp = int(np.abs(got - want.view(np.int32).astype(np.int64)).max())
            assert ulp <= ulp_tol, "%s: observation after %d steps differs from the reference's by %d ulp" % (name, s, ulp)
        if s < steps:
            sim.step(cmds[s])
    if gold.get("ended"):  # the reference's own check_end() ended the game here; so must ours, the same way
        assert bool(sim.done()[0]) and int(sim.results()[0, 0, 7]) == gold["outcome"]


@pytest.mark.parametrize("impl,ulp_tol", IMPLS)
def test_reproduces_the_references_squad_agents_game(impl, ulp_tol):
    """The reference built with USE_AGENT_IN_SQUAD_NPCS (ten Agents in one Squad game): digests of every step; the
    observation the reference gave agent 0 at the loop top = sf_observe before the step, the ones it gave the other
    agents inside human_action (gameplay.hpp:988-999) = sf_observe between sf_step_begin and sf_step_end; which humans
    still have an Agent = sf_agent_alive."""
    import test_squad_agents_example as X
    from strikeforce_amd import abi
    gold = GOLD["squad_agents"]
    w = ref_cases.native(abi.MODE_SQUAD, 2, ref_cases.RICH, maps="shipped")
    w.cfg.n_agents = 10
    sim = impl(w)
    sim.reset((C.c_uint64 * 1)(gold["tb"]), (C.c_uint64 * 1)(gold["serial"]))
    steps = len(gold["digests"]) - 1

    def dense(text):
        v = np.zeros(32 * 31 * 31, dtype=np.uint32)
        for t in text.split():
            i, x = t.split(":")
            v[int(i, 16)] = int(x, 16)
        return v.view(np.int32).astype(np.int64)

    for s in range(steps):
        assert "%016x" % int(sim.digest()[0]) == gold["digests"][s], "state after %d steps" % s
        want = gold["agent_obs_nonzero"].get(str(s))
        top = sim.observe()[0] if want else None
        sim.step_begin()
        if want:
            mid = sim.observe()[0]
            assert len(want) >= 5
            for g, text in want.items():
                got = np.ascontiguousarray((top if g == "0" else mid)[int(g)]).reshape(-1).view(np.int32).astype(np.int64)
                assert int(np.abs(got - dense(text)).max()) <= ulp_tol, "step %d agent %s" % (s, g)
        chars = "".join(X.ACTS[X.policy(g, s)] for g in range(10))
        sim.step_end(np.frombuffer(chars.encode(), dtype=np.uint8))
        assert "".join(str(int(x)) for x in sim.agent_alive()[0]) == gold["agent_alive_after_step"][s], "step %d" % s
    assert "%016x" % int(sim.digest()[0]) == gold["digests"][steps]
