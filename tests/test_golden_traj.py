"""Seeded trajectories against the committed digests (tests/golden/traj_digests.json, made by make_traj.py from the
oracle): the oracle must not drift, and the device core — on the CPU wave emulator here, on the MI355X in the gpu
test — must reproduce every checkpoint."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_traj  # noqa: E402

from emu_lib import Emu  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "traj_digests.json")))["cases"]


@pytest.mark.parametrize("key", sorted(GOLD))
def test_oracle_and_emulated_core_reproduce_the_fixtures(key):
    name, arenas, steps = key.split("/")
    assert make_traj.run_case(name, int(arenas), int(steps), Oracle) == GOLD[key]
    assert make_traj.run_case(name, int(arenas), int(steps), Emu) == GOLD[key]


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(GOLD))
def test_gpu_reproduces_the_fixtures(key):
    import numpy as np
    from strikeforce_amd import env

    class Gpu(env.ArenaBatch):
        def step_many(self, cmds):
            for c in cmds:
                self.step(np.ascontiguousarray(c))

    name, arenas, steps = key.split("/")
    assert make_traj.run_case(name, int(arenas), int(steps), Gpu, obs=False)["digests"] == GOLD[key]["digests"]
