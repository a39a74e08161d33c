"""include/sf_lockstep.hpp (the client side of the reference's lock-step match protocol for C++ hosts of the C-ABI) through
examples/match_client.cpp: the C++ client joins a match on the REFERENCE's own server (oracle/_ref/sf_match_server =
StrikeForce-server/server.cpp compiled as it lies) beside two strikeforce_amd.lockstep clients with different account
records.  A shadow oracle with the C++ client's `ind`, fed the commands the C++ client says it stepped, must reproduce
its digest after every iteration; those commands must be the ones the Python clients stepped.  CPU: the program is
linked against the emulator-backed test library; -m gpu: against libstrikeforce_amd.so."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

import test_lockstep_server as T
from oracle_lib import Oracle, ROOT
from strikeforce_amd import abi, config, lockstep

pytestmark = pytest.mark.skipif(not os.path.exists(T.SERVER), reason="oracle/_ref/sf_match_server not built (no reference checkout)")
DIMS = dict(H=12, Z=10, B=48, P=8)


def build(tmp_path, lib_dir, lib_name):
    exe = str(tmp_path / "match_client")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "match_client.cpp"), "-L", lib_dir, "-l" + lib_name,
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def match(tmp_path, exe, ticks=150):
    teams = [1, 2, 3]
    port, password = T._free_port(), "sesame"
    proc = T._start_server(port, password, teams)
    m, portal = config.synthetic_map(28, 36, wall_p=0.04, portal_pairs=1)
    world = tmp_path / "world.txt"
    world.write_text("28 36 %d %d %d %d\n%s\n%s\n" % (DIMS["H"], DIMS["Z"], DIMS["B"], DIMS["P"], m.decode(), " ".join(map(str, portal))))
    relayed, errors, cpp_out = {}, [], []

    def ours(k):
        try:
            c = lockstep.MatchClient("127.0.0.1", port, password, T.RECORDS[k], name="p%d" % k).connect()
            sim = Oracle(c.workload(28, 36, m, portal, **DIMS))
            rng = np.random.RandomState(1000 + c.ind)
            step0 = sim.step

            def step(cmd):
                relayed.setdefault(c.ind, []).append(bytes(cmd))
                step0(cmd)
            sim.step = step
            lockstep.play(c, sim, lambda _s, _it: abi.BENCH_COMMANDS[rng.randint(0, 28)], max_iterations=ticks)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    def cpp():
        try:
            cpp_out.append(subprocess.check_output([exe, "127.0.0.1", str(port), password, "cpp", str(world), str(ticks), "4242"]
                                                   + [str(t) for t in T.RECORDS[1]], text=True, timeout=100))
        except Exception as e:  # noqa: BLE001
            errors.append(("cpp", repr(e)))

    threads = [threading.Thread(target=ours, args=(0,)), threading.Thread(target=cpp), threading.Thread(target=ours, args=(2,))]
    for t in threads:
        t.start()
        time.sleep(0.4)  # connection order = player index
    for t in threads:
        t.join(timeout=120)
    try:
        proc.stdin.write("done!\n")
        proc.stdin.flush()
        proc.wait(timeout=20)
    except Exception:  # noqa: BLE001
        proc.kill()
    assert not errors, errors
    lines = cpp_out[0].strip().split("\n")
    tb, serial, n, ind, team = (int(x) for x in lines[0].split()[1:])
    assert (n, ind, team) == (3, 1, 2)
    its = [ln.split(" ") for ln in lines if ln.startswith("it ")]
    # (the match's seed is the server's clock: the match may end before `ticks` — this player dead, or both rivals)
    end = lines[-1].split()
    assert end[0] == "end" and int(end[1]) == len(its) and end[2] in ("quit", "won", "died"), lines[-1]
    assert 30 < len(its) <= ticks and (end[2] != "quit" or len(its) == ticks)
    # what the C++ client stepped is what the two Python clients stepped — as far as each of them played: the server
    # seeds the match from its clock, and a player that dies leaves (gameplay.hpp:1103-1143) while the others go on
    stepped = [t[2].encode() for t in its]
    for k in (0, 2):
        assert len(relayed[k]) > 30 and relayed[k] == stepped[:len(relayed[k])], k
    # a shadow oracle in the C++ client's seat (its ind, every player's record as the server relayed it)
    cfg = config.make_config(1, 28, 36, mode=abi.MODE_BATTLE, level=1, n_agents=3, teams=teams, auto_reset=0,
                             player_tokens=T.RECORDS[1], ind=1, agent_tokens=T.RECORDS, **DIMS)
    shadow = Oracle(config.Workload("shadow", cfg, m, portal))
    shadow.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(serial))
    for k, t in enumerate(its):
        shadow.step(np.frombuffer(t[2].encode(), dtype=np.uint8))
        assert t[3] == "%016x" % int(shadow.digest()[0]), "iteration %d" % k


def test_cpp_client_in_a_match_on_the_reference_server(tmp_path):
    d = os.path.join(ROOT, "tests", "emu")
    if os.path.isdir("/root/reference") or not os.path.exists(os.path.join(d, "libsf_emu_abi.so")):
        subprocess.check_call(["make", "-s", "-C", d, "libsf_emu_abi.so"])
    match(tmp_path, build(tmp_path, d, "sf_emu_abi"))


@pytest.mark.gpu
def test_cpp_client_on_the_device_in_a_match_on_the_reference_server(tmp_path):
    match(tmp_path, build(tmp_path, os.path.join(ROOT, "strikeforce_amd"), "strikeforce_amd"))
