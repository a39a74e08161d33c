"""The device core (strikeforce_amd/csrc/sf_core.hpp, sf_obs.hpp) run on the CPU wave emulator against
the oracle: bit-exact full-state dumps after every step, and exact observations.  This is the same source
the gfx950 kernels are built from; only the wave backend differs (tests/emu/wave_emu.hpp)."""
import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import Oracle, diff_dumps
from strikeforce_amd import config


def _lockstep(name, arenas, steps, check_every=1, **kw):
    w = config.baseline_workload(name, arenas=arenas, **kw)
    o, e = Oracle(w), Emu(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), e.reset(tb, sr)
    cmds, _ = config.bench_commands(arenas, w.cfg.n_agents, steps)
    for s in range(-1, steps):
        if s >= 0:
            o.step(cmds[s]), e.step(cmds[s])
        if s % check_every == 0 or s == steps - 1:
            for a in range(arenas):
                d = diff_dumps(o.dump(a).as_dict(), e.dump(a).as_dict())
                assert d is None, "%s step %d arena %d: %s" % (name, s, a, d)
                # the draws of every phase of the step (SURVEY §8c golden item 5: pins the ORDER in which phases draw)
                assert s < 0 or e.phase_draws(a) == o.phase_draws(a), "%s step %d arena %d: draws per phase %s, the oracle's %s" % (
                    name, s, a, e.phase_draws(a), o.phase_draws(a))
    assert (o.digest() == e.digest()).all()
    assert (o.results() == e.results()).all()
    assert (o.done() == e.done()).all()
    return o, e


@pytest.mark.parametrize("name,steps", [("C1", 120), ("C2", 120), ("C3", 100), ("C4", 60), ("C5", 40), ("STRESS", 150),
                                        ("MAXCAP", 40), ("FLOORS", 120), ("NATIVE", 150)])
def test_lockstep_state_parity(name, steps):
    _lockstep(name, 2, steps)


def test_multi_step_launch_equals_single_steps():
    w = config.baseline_workload("C3", arenas=3)
    a, b = Emu(w), Emu(w)
    tb, sr = w.seeds()
    a.reset(tb, sr), b.reset(tb, sr)
    cmds, _ = config.bench_commands(3, 1, 64)
    for s in range(64):
        a.step(cmds[s])
    b.step_many(cmds)
    assert (a.digest() == b.digest()).all()


def test_observation_parity_exact():
    for name, steps in (("C3", 90), ("C5", 30)):
        o, e = _lockstep(name, 2, steps, check_every=1000)
        x, y = o.observe(), e.observe()
        assert x.shape == y.shape
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
        assert np.count_nonzero(x) > 100


def test_done_counts_episodes_over_a_multi_step_launch():
    """With auto_reset, sf_done reports how many episodes ended during the last call, over all of its k iterations
    (strikeforce.h): equal on oracle and device source, and equal to the growth of the episode counter."""
    w = config.baseline_workload("C2", arenas=8)
    o, e = Oracle(w), Emu(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), e.reset(tb, sr)
    cmds, _ = config.bench_commands(8, 1, 1500)
    o.step_many(cmds), e.step_many(cmds)
    eps = np.array([o.dump(a).hdr.episodes for a in range(8)])
    assert (o.done() == eps).all() and (e.done() == eps).all() and eps.max() >= 2
    one, _ = config.bench_commands(8, 1, 1)
    o.step(one[0]), e.step(one[0])
    assert (o.done() == e.done()).all() and o.done().max() <= 1
