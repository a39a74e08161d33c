#!/usr/bin/env python3
"""Generates tests/golden/traj_digests.json from the CPU oracle (oracle/sf_oracle.c): 64-bit state digests of seeded
runs.  These are regression vectors of the oracle (a changed digest means the restatement changed), not outputs of the
reference itself — the reference cannot be built in this image.

    python tests/golden/make_traj.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle_lib import Oracle  # noqa: E402
from strikeforce_amd import config  # noqa: E402

# lengths after SURVEY.md §8c: the native world for 2000 steps on three seeds, every BASELINE configuration for 1000
CASES = [("C1", 2, 1000), ("C2", 2, 1000), ("C3", 2, 1000), ("C4", 2, 1000), ("C5", 2, 1000), ("STRESS", 2, 400),
         ("FLOORS", 2, 400), ("MAXCAP", 1, 100), ("NATIVE", 3, 2000)]
CHECKPOINTS = 4


def run_case(name, arenas, steps, make=Oracle, obs=True):
    """{"digests": {step: [digest per arena]}, "obs_crc": {step: crc32 of every agent's 32x31x31 float32 observation}}.
    The digest covers every state word including jomle, i.e. the number of RNG draws so far.  The observation CRC is
    bit-exact and therefore only compared between CPU implementations (the GPU's pow may differ by one ulp)."""
    import zlib
    w = config.baseline_workload(name, arenas=arenas)
    sim = make(w)
    tb, sr = w.seeds()
    sim.reset(tb, sr)
    cmds, _ = config.bench_commands(arenas, w.cfg.n_agents, steps)
    out = {"digests": {"0": [int(x) for x in sim.digest()]}, "obs_crc": {}}
    chunk = steps // CHECKPOINTS
    for c in range(CHECKPOINTS):
        sim.step_many(cmds[c * chunk:(c + 1) * chunk])
        out["digests"][str((c + 1) * chunk)] = [int(x) for x in sim.digest()]
        if obs:
            out["obs_crc"][str((c + 1) * chunk)] = zlib.crc32(sim.observe().tobytes())
    return out


if __name__ == "__main__":
    data = {"_generator": "tests/golden/make_traj.py (oracle/sf_oracle.c)", "cases": {}}
    for name, arenas, steps in CASES:
        data["cases"]["%s/%d/%d" % (name, arenas, steps)] = run_case(name, arenas, steps)
    json.dump(data, open(os.path.join(HERE, "traj_digests.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(data["cases"]), "cases")
