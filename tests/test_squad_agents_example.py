"""examples/squad_agents.cpp (ten reference-style Agents in one Squad game through include/sf_agent_adapter.hpp) against
the reference built with USE_AGENT_IN_SQUAD_NPCS and scripted with the same policy: the order of predict / update calls
of every step, which agents are destroyed when, and the final state.  On the CPU the program is linked against the
emulator-backed test library (tests/emu/libsf_emu_abi.so) and compared with the reference live; on the GPU (-m gpu) it is
linked against libstrikeforce_amd.so and compared with tests/golden/ref_squad_calls.json, which the CPU test's generator
wrote from the reference (`PYTHONPATH=.:tests python tests/test_squad_agents_example.py`)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import ref_cases
import reftick
from oracle_lib import Oracle
from strikeforce_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "ref_squad_calls.json")
ACTS = "+xzqeawsd"
STEPS = 300


def policy(agent, step):
    return (agent * 7 + step * 3 + step // 5) % 9  # the Agent of examples/squad_agents.cpp


def reference_log(steps=STEPS):
    """Per step: [[agent, 'P'|'U', action] ...] in call order, sorted ids destroyed during the step; final digest of the
    oracle run alongside (equal to the reference's state at every step)."""
    w = ref_cases.native(abi.MODE_SQUAD, 2, ref_cases.RICH, maps="shipped")
    w.cfg.n_agents = 10
    o = Oracle(w)
    r = reftick.RefTick(w, ref_cases.RICH, agents=True, squad_agents=True)
    o.reset((C.c_uint64 * 1)(1700000000), (C.c_uint64 * 1)(123456789))
    r.reset(1700000000, 123456789)
    r.calls()
    out = []
    for s in range(steps):
        chars = "".join(ACTS[policy(g, s)] for g in range(10))
        o.step(np.frombuffer(chars.encode(), dtype=np.uint8))
        r.step(chars)
        calls = r.calls()
        assert reftick.first_difference(r.dump(), reftick.arrays_of(o.dump(0))) is None
        out.append({"calls": [[c[0], c[1], c[2]] for c in calls if c[1] in "PU"],
                    "destroyed": sorted(c[0] for c in calls if c[1] == "D")})
        if o.done()[0]:
            break
    r.close()
    return {"steps": out, "digest": "%016x" % int(o.digest()[0])}


def program_log(lib_dir, lib_name, tmp_path, steps=STEPS):
    exe = str(tmp_path / "squad_agents")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "squad_agents.cpp"), "-L", lib_dir, "-l" + lib_name,
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    text = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "maps"), str(steps)], text=True)
    out, cur, digest = [], None, None
    for ln in text.split("\n"):
        t = ln.split()
        if not t:
            continue
        if t[0] == "S":
            cur = {"calls": [], "destroyed": []}
            out.append(cur)
        elif t[0] in "PU" and cur is not None:
            cur["calls"].append([int(t[1]), t[0], int(t[2])])
        elif t[0] == "D" and cur is not None:
            cur["destroyed"].append(int(t[1]))
        elif t[0] == "E":
            cur = None  # the Agents destroyed when the runner goes out of scope are not deaths
        elif t[0] == "digest":
            digest = t[1]
    for st in out:
        st["destroyed"].sort()
    return {"steps": out, "digest": digest}


def compare(got, want):
    assert len(got["steps"]) == len(want["steps"])
    for s, (g, w) in enumerate(zip(got["steps"], want["steps"])):
        assert g["calls"] == w["calls"], "step %d: the adapter asked %s, the reference %s" % (s, g["calls"], w["calls"])
        assert g["destroyed"] == w["destroyed"], "step %d: destroyed %s, in the reference %s" % (s, g["destroyed"], w["destroyed"])
    assert got["digest"] == want["digest"]


@pytest.mark.skipif(not reftick.available(), reason="oracle/_ref/sf_ref_tick not built (no reference checkout)")
def test_adapter_on_the_emulator_asks_what_the_reference_asks(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libsf_emu_abi.so"])
    want = reference_log()
    assert sum(len(s["destroyed"]) for s in want["steps"]) >= 2  # agents die in this game
    compare(program_log(os.path.join(ROOT, "tests", "emu"), "sf_emu_abi", tmp_path), want)
    with open(GOLD) as f:
        assert json.load(f)["log"] == want, "tests/golden/ref_squad_calls.json is stale: python tests/test_squad_agents_example.py"


@pytest.mark.gpu
def test_adapter_on_the_device_asks_what_the_reference_asked(tmp_path):
    with open(GOLD) as f:
        gold = json.load(f)
    assert "sf_ref_tick" in gold["_generator"]
    compare(program_log(os.path.join(ROOT, "strikeforce_amd"), "strikeforce_amd", tmp_path), gold["log"])


if __name__ == "__main__":
    with open(GOLD, "w") as f:
        json.dump({"_generator": "tests/test_squad_agents_example.py reference_log(): call sequence of oracle/_ref/sf_ref_tick_squadagents "
                                 "(the reference built with USE_AGENT_IN_SQUAD_NPCS, oracle/ref_tick.py) scripted with examples/squad_agents.cpp's policy",
                   "log": reference_log()}, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", GOLD)
