"""examples/drive_agent.cpp: a reference-style C++ Agent driven through the C-ABI by include/sf_agent_adapter.hpp
must walk the same trajectory as the same policy driven from Python (ctypes) and as the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import Oracle
from strikeforce_amd import abi, config, env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACTION = "+xzqeawsd"


def _workload(arenas):
    grid = [["."] * 32 for _ in range(32)]
    for i in range(32):
        grid[0][i] = grid[31][i] = grid[i][0] = grid[i][31] = "#"
    for r in range(4, 28, 5):
        for c in range(3, 29, 7):
            grid[r][c] = "#"
    m = "".join("".join(row) for row in grid).encode()
    cfg = config.make_config(arenas, 32, 32, H=4, Z=8, B=16, P=4, mode=abi.MODE_SOLO, auto_reset=1,
                             player_tokens=config.HUMAN_ENEMY_TOKENS)
    cfg.cap_portals = 4
    return config.Workload("example", cfg, m, [-1] * 1024)


def _drive(sim, arenas, steps):
    """The Agent of drive_agent.cpp: index (t + 1) % 9 (its own cell always shows a character), fresh per episode."""
    t = np.zeros(arenas, dtype=np.int64)
    ended = 0
    for _ in range(steps):
        obs = sim.observe()
        me = obs[:, 0, 0, 15, 15] > 0
        idx = (t + me.astype(np.int64)) % 9
        t += 1
        sim.step(np.array([ord(ACTION[i]) for i in idx], dtype=np.uint8))
        d = sim.done().astype(bool)
        ended += int(d.sum())
        t[d] = 0
    return ended


@pytest.mark.gpu
def test_cpp_adapter_example_matches_python_and_oracle(tmp_path):
    exe = str(tmp_path / "drive_agent")
    lib = os.path.join(ROOT, "strikeforce_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "drive_agent.cpp"), "-L", lib, "-lstrikeforce_amd",
                           "-Wl,-rpath," + lib, "-o", exe])
    arenas, steps = 8, 300
    out = subprocess.check_output([exe, str(arenas), str(steps)], text=True).split("\n")
    ended_cpp = int(out[0].split()[1])
    dig_cpp = [int(ln.split()[2]) for ln in out if ln.startswith("digest")]
    w = _workload(arenas)
    tb, sr = w.seeds()
    g, o = env.ArenaBatch(w), Oracle(w)
    g.reset(tb, sr), o.reset(tb, sr)
    ended_gpu = _drive(g, arenas, steps)
    ended_orc = _drive(o, arenas, steps)
    assert dig_cpp == [int(x) for x in g.digest()] == [int(x) for x in o.digest()]
    assert ended_cpp == ended_gpu == ended_orc


@pytest.mark.gpu
def test_the_python_closed_loop_example_runs():
    """examples/policy_loop.py (list observation -> sf_policy_predict_sparse -> step, everything on the device)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "policy_loop.py"), "64", "30"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "agent-steps/s" in out.stdout
