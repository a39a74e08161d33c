"""The lock-step wire adapter (strikeforce_amd/lockstep.py) against the REFERENCE match server itself:
oracle/_ref/sf_match_server is StrikeForce-server/server.cpp compiled unmodified (oracle/Makefile `ref`).  Three
clients join one match, each simulating the whole world with its own `ind`; the server hands out seed, indices and
teams and relays the commands.  All worlds must stay identical, and the server must see the match through."""
import os
import socket
import subprocess
import threading
import time

import numpy as np
import pytest

from oracle_lib import Oracle, ROOT
from strikeforce_amd import abi, config, lockstep

SERVER = os.path.join(ROOT, "oracle", "_ref", "sf_match_server")
pytestmark = pytest.mark.skipif(not os.path.exists(SERVER), reason="oracle/_ref/sf_match_server not built (no reference checkout)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _start_server(port, password, teams):
    try:  # the "local" branch dereferences gethostbyname(hostname) unchecked (server.cpp:156-158)
        socket.gethostbyname(socket.gethostname())
        kind = "L"
    except OSError:
        kind = "G"  # prints nothing useful without a network, but does not crash
    script = "%s\n%d\n%s\n%d %d\n%s\n" % (kind, port, password, len(teams), max(teams), " ".join(map(str, teams)))
    proc = subprocess.Popen([SERVER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    proc.lines = []
    ready = threading.Event()

    def pump():  # the server announces "Server is running..." once it listens (server.cpp:204)
        for line in proc.stdout:
            proc.lines.append(line)
            if "Server is running" in line:
                ready.set()

    threading.Thread(target=pump, daemon=True).start()
    proc.stdin.write(script)
    proc.stdin.flush()
    assert ready.wait(20), "reference server did not start: %r" % proc.lines
    return proc


def _world(d):
    return ([(h.alive, h.f, h.r, h.c, h.way, h.team, h.hp, h.stamina, h.mindamage, h.vec, h.ind, tuple(h.cons),
              tuple(h.throw_cnt), h.blocks, h.portals) for h in d.humans],
            [(z.alive, z.r, z.c, z.hp, z.mindamage) for z in d.zombies],
            [(b.alive, b.r, b.c, b.way, b.damage, b.ref) for b in d.bullets],
            d.flags.tobytes(), d.hdr.frame, d.hdr.jomle, tuple(d.hdr.rng))


def run_match(make_sim, teams, quit_at=None, ticks=400, records=None):
    port, password = _free_port(), "sesame"
    proc = _start_server(port, password, teams)
    m, portal = config.synthetic_map(28, 36, wall_p=0.04, portal_pairs=1)
    n = len(teams)
    worlds = [dict() for _ in range(n)]
    results = [None] * n
    errors = []

    def client_thread(k):
        try:
            rec = records[k] if records else config.HUMAN_ENEMY_TOKENS
            c = lockstep.MatchClient("127.0.0.1", port, password, rec, name="p%d" % k).connect()
            sim = make_sim(c.workload(28, 36, m, portal, H=12, Z=10, B=48, P=8))
            rng = np.random.RandomState(1000 + c.ind)

            def policy(_sim, it):
                if quit_at is not None and c.ind == quit_at[0] and it == quit_at[1]:
                    return "_"
                return abi.BENCH_COMMANDS[rng.randint(0, 28)]

            def snap(it, s):
                worlds[c.ind][it] = _world(s.dump(0))

            results[c.ind] = (c.ind, c.team, c.tb, c.serial) + lockstep.play(c, sim, policy, max_iterations=ticks,
                                                                             on_iteration=snap)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=client_thread, args=(k,)) for k in range(n)]
    for t in threads:
        t.start()
        time.sleep(0.3)  # connection order = player index (the server hands out indices in accept order)
    for t in threads:
        t.join(timeout=120)
    try:
        proc.stdin.write("done!\n")
        proc.stdin.flush()
        proc.wait(timeout=20)
    except Exception:  # noqa: BLE001
        proc.kill()
    time.sleep(0.1)
    out = "".join(proc.lines)
    assert not errors, errors
    return results, worlds, out


def _check(results, worlds, out, teams):
    n = len(teams)
    assert all(r is not None for r in results)
    assert sorted(r[0] for r in results) == list(range(n))                # server-assigned indices (server.cpp:243-246)
    assert [r[1] for r in sorted(results)] == teams                       # and teams
    assert len({(r[2], r[3]) for r in results}) == 1                      # one shared (tb, serial) seed
    compared = 0
    for it in range(max(max(w) if w else 0 for w in worlds) + 1):
        views = [w[it] for w in worlds if it in w]
        for v in views[1:]:
            assert v == views[0], "iteration %d: the clients' worlds differ" % it
        compared += len(views) > 1
    assert compared > 30
    assert "Final result" in out                                          # the server ran the match to its end


def test_three_clients_through_the_reference_server():
    teams = [1, 2, 1]
    results, worlds, out = run_match(Oracle, teams, quit_at=(1, 60), ticks=250)
    _check(results, worlds, out, teams)
    assert [r[5] for r in sorted(results)][1] == "quit"
    assert "quited" in out  # the server's own log line for the '_' it relayed (server.cpp:88-89)


# three different account records, exchanged through the reference server as Human::log_file blobs (gameplay.hpp:120-151)
RECORDS = [config.HUMAN_ENEMY_TOKENS,
           [15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34] + [1] * 16 + [56],
           config.HUMAN_TOKENS]


def _check_records(worlds):
    first = worlds[0][0][0]  # humans of iteration 0 as player 0 sees them: (alive, f, r, c, way, team, hp, stamina, ...)
    assert [h[6] for h in first[:3]] == [1000, 15000, 1000]          # Hp of the three records
    assert [h[7] for h in first[:3]] == [1000000, 15000, 1000]       # stamina


def test_three_clients_with_different_account_records():
    teams = [1, 2, 3]
    results, worlds, out = run_match(Oracle, teams, ticks=150, records=RECORDS)
    _check(results, worlds, out, teams)
    _check_records(worlds)


@pytest.mark.gpu
def test_three_gpu_clients_with_different_account_records():
    from strikeforce_amd import env
    teams = [1, 2, 3]
    results, worlds, out = run_match(env.ArenaBatch, teams, ticks=150, records=RECORDS)
    _check(results, worlds, out, teams)
    _check_records(worlds)


@pytest.mark.gpu
def test_three_gpu_clients_through_the_reference_server():
    from strikeforce_amd import env
    teams = [1, 2, 3]
    results, worlds, out = run_match(env.ArenaBatch, teams, ticks=200)
    _check(results, worlds, out, teams)
