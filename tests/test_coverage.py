"""The parity runs must actually exercise the reference's branches: the oracle counts them (test aid), and this
test asserts that the configurations used by the lockstep / digest tests reach every one of them.  STRESS is a
non-BASELINE configuration with tiny slot pools (5 bullets, 3 portals, 6 chests) and a non-square map, so that
every allocator (`b_ind`, `p_ind`, `z_ind`, `h_ind`, chest cap; gameplay.hpp:209-235,533) runs dry."""
import numpy as np

from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import config


def _run(name, arenas, steps):
    w = config.baseline_workload(name, arenas=arenas)
    o, e = Oracle(w), Emu(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), e.reset(tb, sr)
    cmds, _ = config.bench_commands(arenas, w.cfg.n_agents, steps)
    o.step_many(cmds), e.step_many(cmds)
    assert (o.digest() == e.digest()).all(), name
    assert (o.results() == e.results()).all(), name
    return o.events()


def test_branch_coverage_of_the_parity_configs():
    total = {}
    for name, arenas, steps in (("C3", 4, 700), ("STRESS", 6, 900)):
        for k, v in _run(name, arenas, steps).items():
            total[k] = total.get(k, 0) + v
    # (credit_slot_reused — a hit credited to the newcomer in a dead owner's slot, App. E-11 — needs a bullet that outlives
    # its owner and a spawn in between: scripted in tests/test_order_scenarios.py, never met by random play)
    missing = [k for k, v in total.items() if v == 0 and k != "credit_slot_reused"]
    assert not missing, "branches never taken: %s" % missing


def test_stress_runs_every_pool_dry():
    ev = _run("STRESS", 6, 900)
    assert ev["no_bullet_slot"] > 0 and ev["zombie_spawn"] > 0 and ev["npc_spawn"] > 0
    assert ev["episode_end"] > 0  # 600-frame Timer clock: episodes end and restart inside the run


def test_human_action_takes_both_of_its_forms():
    """human_action evaluates the sweep lane-parallel when no two humans meet on a cell and falls back to one human at a
    time otherwise (sf_core.hpp): both forms must occur in the parity runs, or one of them is untested."""
    import ctypes as C
    import emu_lib
    L = emu_lib.lib()
    cn = (C.c_uint64 * 8)()
    L.sfe_counts(cn)  # clear
    for name, arenas, steps in (("C3", 4, 500), ("STRESS", 6, 600), ("MAXCAP", 2, 120)):
        _run(name, arenas, steps)
        L.sfe_counts(cn)
        assert cn[0] > 100 and cn[1] > 5, (name, cn[0], cn[1])
