#!/usr/bin/env python3
"""Generates tests/golden/policy_vectors.json from the PyTorch f32 restatement of the reference's bot network
(oracle/policy_ref.py): for seeded parameters and the oracle simulator's own observations of two agents over three
recurrent steps, the probabilities, value and a checksum of the recurrent state.  These are regression vectors of the
restatement (libtorch, hence the reference's own model, cannot be run in this image: DESIGN.md §10).

    python tests/golden/make_policy_vectors.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import policy_ref  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from strikeforce_amd import config, policy  # noqa: E402

PARAM_SEED, ARENAS, STEPS, SIM_STEPS_BETWEEN = 21, 2, 3, 25


def trajectory():
    """[(obs [2,32,31,31], action one-hot fed back), ...]: observations of workload C1 every 25 simulator steps."""
    w = config.baseline_workload("C1", arenas=ARENAS)
    o = Oracle(w)
    o.reset(*w.seeds())
    cmds, _ = config.bench_commands(ARENAS, w.cfg.n_agents, STEPS * SIM_STEPS_BETWEEN)
    out = []
    for t in range(STEPS):
        o.step_many(cmds[t * SIM_STEPS_BETWEEN:(t + 1) * SIM_STEPS_BETWEEN])
        out.append(o.observe().reshape(ARENAS, 32, 31, 31).copy())
    return out


def run():
    params = policy.init_parameters(seed=PARAM_SEED)
    h = np.zeros((2, ARENAS, 160), dtype=np.float32)
    a = np.eye(9, dtype=np.float32)[[0] * ARENAS]
    steps = []
    for obs in trajectory():
        probs, value, h = policy_ref.forward_batched(params, obs, h, a)
        act = probs.argmax(axis=1)
        a = np.eye(9, dtype=np.float32)[act]
        steps.append({"probs": [[float(x) for x in r] for r in probs], "value": [float(x) for x in value],
                      "h_abs_sum": [float(np.abs(h[g]).sum()) for g in range(2)], "action_fed_back": [int(x) for x in act],
                      "obs_nonzero": int((obs != 0).sum())})
    return {"_generator": "tests/golden/make_policy_vectors.py (oracle/policy_ref.py, torch f32 on the CPU)",
            "param_seed": PARAM_SEED, "workload": "C1", "arenas": ARENAS, "sim_steps_between": SIM_STEPS_BETWEEN,
            "steps": steps}


if __name__ == "__main__":
    path = os.path.join(HERE, "policy_vectors.json")
    json.dump(run(), open(path, "w"), indent=1)
    print("wrote", path)
