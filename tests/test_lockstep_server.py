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
    # Hp and stamina of the three records — behind the first iteration, whose seed is the server's clock: somebody may
    # already have been punched (Hp down by a few hundred) or have spent stamina on a shot (at most 50)
    for h, hp, st in zip(first[:3], [1000, 15000, 1000], [1000000, 15000, 1000]):
        assert hp - 800 < h[6] <= hp and st - 50 <= h[7] <= st, (h[6], h[7], hp, st)


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


# ---- the REFERENCE CLIENT in the match ---------------------------------------------------------------------------------
def test_the_reference_client_and_two_of_ours_play_one_match():
    """The online branch of load_data (gameplay.hpp:1795-1859: join, exchange records, the shared seed, random start
    cells and facings), Client::send_it / recieve (:113-118,170-193) and human_action's handling of remote players
    (:977-993), on the reference's own client: oracle/_ref/sf_ref_tick (oracle/ref_tick.py, mode "Battle Royal") joins the
    reference's own server as one of three players; the other two are strikeforce_amd.lockstep clients.  Afterwards a
    shadow oracle with the reference client's `ind`, fed with the commands the match relayed, must equal the reference
    client's whole world after every iteration — including the kill / loot counters, which are kept relative to `ind`."""
    import reftick
    if not reftick.available():
        pytest.skip("oracle/_ref/sf_ref_tick not built")
    from oracle_lib import Oracle
    rich = [15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34] + [1] * 16 + [56]  # nobody dies in 120 iterations
    # (the reference pools exits by B, `portal[B]` gameplay.hpp:51-53, and a level-10 account carries ten portals: P = B here)
    teams, ticks = [1, 2, 3], 120
    port, password = _free_port(), "sesame"
    proc = _start_server(port, password, teams)
    m, portal = config.synthetic_map(28, 36, wall_p=0.04, portal_pairs=1)
    cfg0 = config.make_config(1, 28, 36, H=12, Z=10, B=48, P=48, mode=abi.MODE_BATTLE, n_agents=3, teams=teams, auto_reset=0,
                              player_tokens=rich)
    ref = reftick.RefTick(config.Workload("match", cfg0, m, portal), rich, native_caps=False)
    errors, ours, relayed, ref_dumps, ref_info = [], {}, {}, {}, {}

    def ref_thread():
        try:
            tb, serial, ind, n, team = ref.join_match("127.0.0.1", port, password)
            ref_info.update(tb=tb, serial=serial, ind=ind, n=n, team=team)
            rng = np.random.RandomState(77)
            ref_dumps[-1] = ref.dump()
            for it in range(ticks):
                ref.step(abi.BENCH_COMMANDS[rng.randint(0, 28)])
                ref_dumps[it] = ref.dump()
        except Exception as e:  # noqa: BLE001
            errors.append(("reference client", repr(e)))

    def our_thread(k):
        try:
            c = lockstep.MatchClient("127.0.0.1", port, password, rich, name="p%d" % k).connect()
            sim = Oracle(c.workload(28, 36, m, portal, H=12, Z=10, B=48, P=48))
            ours[c.ind] = c
            rng = np.random.RandomState(1000 + c.ind)
            policy = lambda _s, it: abi.BENCH_COMMANDS[rng.randint(0, 28)]
            # the commands of every iteration, as this client stepped them
            orig_step = sim.step

            def step(cmd):
                relayed.setdefault(c.ind, []).append(bytes(cmd))
                orig_step(cmd)
            sim.step = step
            lockstep.play(c, sim, policy, max_iterations=ticks)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=our_thread, args=(0,)), threading.Thread(target=ref_thread), threading.Thread(target=our_thread, args=(2,))]
    for t in threads:
        t.start()
        time.sleep(0.4)  # connection order = player index
    for t in threads:
        t.join(timeout=120)
    ref.close()
    try:
        proc.stdin.write("done!\\n")
        proc.stdin.flush()
        proc.wait(timeout=20)
    except Exception:  # noqa: BLE001
        proc.kill()
    assert not errors, errors
    assert ref_info["ind"] == 1 and ref_info["n"] == 3 and ref_info["team"] == 2
    c0 = ours[0]
    assert (c0.tb, c0.serial) == (ref_info["tb"], ref_info["serial"])
    assert relayed[0] == relayed[2] and len(relayed[0]) == ticks
    # the shadow: this repo's simulation as the reference client's process sees the match (ind = 1)
    cfg = config.make_config(1, 28, 36, H=12, Z=10, B=48, P=48, mode=abi.MODE_BATTLE, level=1, n_agents=3, teams=c0.teams,
                             auto_reset=0, player_tokens=rich, ind=1, agent_tokens=c0.records)
    shadow = Oracle(config.Workload("shadow", cfg, m, portal))
    import ctypes as C
    shadow.reset((C.c_uint64 * 1)(c0.tb), (C.c_uint64 * 1)(c0.serial))
    d = reftick.first_difference(ref_dumps[-1], reftick.arrays_of(shadow.dump(0)))
    assert d is None, "after the placement: " + d
    for it in range(ticks):
        shadow.step(np.frombuffer(relayed[0][it], dtype=np.uint8))
        d = reftick.first_difference(ref_dumps[it], reftick.arrays_of(shadow.dump(0)))
        assert d is None, "iteration %d: %s" % (it, d)


def test_a_configs4_shaped_battle_with_the_reference_client_in_it():
    """BASELINE configs[4]'s shape — Battle Royale, 8 players, 256 x 256, pools H8 Z56 B128 — as a real match: the
    reference's own client compiled for those dimensions (oracle/ref_tick.py TEST_DIMS; load_data's online branch places
    the eight players on random '.' cells of the big map, gameplay.hpp:1846-1859) joins the reference's server in seat 3,
    seven strikeforce_amd.lockstep clients take the other seats.  A shadow oracle in the reference client's seat, fed the
    commands the match relayed, equals the reference client's whole world — 65 536 cells, every slot — after the
    placement and after every one of 60 iterations."""
    import reftick
    if not reftick.available():
        pytest.skip("oracle/_ref/sf_ref_tick not built")
    from oracle_lib import Oracle
    import ctypes as C
    rich = [15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34] + [1] * 16 + [56]  # nobody dies in 60 iterations
    n, seat, ticks = 8, 3, 60
    teams = list(range(1, n + 1))
    dims = dict(H=8, Z=56, B=128, P=128)  # (the reference pools exits by B, `portal[B]` gameplay.hpp:51-53)
    port, password = _free_port(), "sesame"
    proc = _start_server(port, password, teams)
    m, portal = config.synthetic_map(256, 256, portal_pairs=2)
    cfg0 = config.make_config(1, 256, 256, mode=abi.MODE_BATTLE, n_agents=n, teams=teams, auto_reset=0, player_tokens=rich, **dims)
    ref = reftick.RefTick(config.Workload("match", cfg0, m, portal), rich, native_caps=False)
    errors, ours, relayed, ref_dumps, ref_info = [], {}, {}, {}, {}

    def ref_thread():
        try:
            tb, serial, ind, nn, team = ref.join_match("127.0.0.1", port, password)
            ref_info.update(tb=tb, serial=serial, ind=ind, n=nn, team=team)
            rng = np.random.RandomState(77)
            ref_dumps[-1] = ref.dump()
            for it in range(ticks):
                ref.step(abi.BENCH_COMMANDS[rng.randint(0, 28)])
                ref_dumps[it] = ref.dump()
        except Exception as e:  # noqa: BLE001
            errors.append(("reference client", repr(e)))

    def our_thread(k):
        try:
            c = lockstep.MatchClient("127.0.0.1", port, password, rich, name="p%d" % k).connect()
            sim = Oracle(c.workload(256, 256, m, portal, **dims))
            ours[c.ind] = c
            rng = np.random.RandomState(1000 + c.ind)
            step0 = sim.step

            def step(cmd):
                relayed.setdefault(c.ind, []).append(bytes(cmd))
                step0(cmd)
            sim.step = step
            lockstep.play(c, sim, lambda _s, _it: abi.BENCH_COMMANDS[rng.randint(0, 28)], max_iterations=ticks)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=ref_thread) if k == seat else threading.Thread(target=our_thread, args=(k,)) for k in range(n)]
    for t in threads:
        t.start()
        time.sleep(0.4)  # connection order = player index
    for t in threads:
        t.join(timeout=180)
    ref.close()
    try:
        proc.stdin.write("done!\n")
        proc.stdin.flush()
        proc.wait(timeout=20)
    except Exception:  # noqa: BLE001
        proc.kill()
    assert not errors, errors
    assert (ref_info["ind"], ref_info["n"], ref_info["team"]) == (seat, n, teams[seat])
    c0 = ours[0]
    assert (c0.tb, c0.serial) == (ref_info["tb"], ref_info["serial"])
    assert len(relayed[0]) == ticks and all(relayed[k] == relayed[0] for k in ours)
    cfg = config.make_config(1, 256, 256, mode=abi.MODE_BATTLE, level=1, n_agents=n, teams=c0.teams, auto_reset=0,
                             player_tokens=rich, ind=seat, agent_tokens=c0.records, **dims)
    shadow = Oracle(config.Workload("shadow", cfg, m, portal))
    shadow.reset((C.c_uint64 * 1)(c0.tb), (C.c_uint64 * 1)(c0.serial))
    d = reftick.first_difference(ref_dumps[-1], reftick.arrays_of(shadow.dump(0)))
    assert d is None, "after the placement: " + d
    for it in range(ticks):
        shadow.step(np.frombuffer(relayed[0][it], dtype=np.uint8))
        d = reftick.first_difference(ref_dumps[it], reftick.arrays_of(shadow.dump(0)))
        assert d is None, "iteration %d: %s" % (it, d)


def test_the_reference_client_wins_the_match_when_both_rivals_have_quit():
    """check_end's online branch on the reference's own client (gameplay.hpp:1103-1119; kept in oracle/_ref/sf_ref_tick
    with only its end screen and key wait blanked, oracle/ref_tick.py): the two other players — strikeforce_amd.lockstep
    clients — leave the match with '_' at iterations 5 and 8 ('_' takes the player's Hp to 0 in every simulation,
    gameplay.hpp:696-699; its client is gone, :939-953).  The reference client's check_end() must say "go on" at every loop
    top before, and "over" (`online && rivals_are_dead()`) at the loop top behind iteration 8 — where a shadow oracle
    with the reference client's `ind`, fed the relayed commands, ends too, won."""
    import reftick
    if not reftick.available():
        pytest.skip("oracle/_ref/sf_ref_tick not built")
    from oracle_lib import Oracle
    import ctypes as C
    rich = [15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34] + [1] * 16 + [56]
    teams, quits = [1, 2, 3], {0: 5, 2: 8}
    port, password = _free_port(), "sesame"
    proc = _start_server(port, password, teams)
    m, portal = config.synthetic_map(28, 36, wall_p=0.04, portal_pairs=1)
    cfg0 = config.make_config(1, 28, 36, H=12, Z=10, B=48, P=48, mode=abi.MODE_BATTLE, n_agents=3, teams=teams, auto_reset=0,
                              player_tokens=rich)
    ref = reftick.RefTick(config.Workload("match", cfg0, m, portal), rich, native_caps=False)
    errors, ours, ref_cmds, ref_ended, ref_info = [], {}, [], [], {}

    def ref_thread():
        try:
            tb, serial, ind, n, team = ref.join_match("127.0.0.1", port, password)
            ref_info.update(tb=tb, serial=serial, ind=ind)
            rng = np.random.RandomState(78)
            for it in range(40):
                c = abi.BENCH_COMMANDS[rng.randint(0, 28)]
                ref_cmds.append(c)
                ref.step(c)
                ref_ended.append(ref.ended)
                if ref.ended:
                    break
        except Exception as e:  # noqa: BLE001
            errors.append(("reference client", repr(e)))

    def our_thread(k):
        try:
            c = lockstep.MatchClient("127.0.0.1", port, password, rich, name="p%d" % k).connect()
            sim = Oracle(c.workload(28, 36, m, portal, H=12, Z=10, B=48, P=48))
            ours[c.ind] = c
            rng = np.random.RandomState(2000 + c.ind)
            policy = lambda _s, it: "_" if it == quits[c.ind] else abi.BENCH_COMMANDS[rng.randint(0, 28)]
            how = lockstep.play(c, sim, policy, max_iterations=40)
            assert how == (quits[c.ind], "quit"), how
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=our_thread, args=(0,)), threading.Thread(target=ref_thread), threading.Thread(target=our_thread, args=(2,))]
    for t in threads:
        t.start()
        time.sleep(0.4)  # connection order = player index
    for t in threads:
        t.join(timeout=60)
    ref.close()
    try:
        proc.stdin.write("done!\\n")
        proc.stdin.flush()
        proc.wait(timeout=3)
    except Exception:  # noqa: BLE001
        proc.kill()
    assert not errors, errors
    assert ref_info["ind"] == 1
    assert ref_ended == [False] * 8 + [True], ref_ended
    # the shadow: the match as the reference client's process sees it (ind = 1); the others' commands are random until they quit
    c0 = ours[0]
    cfg = config.make_config(1, 28, 36, H=12, Z=10, B=48, P=48, mode=abi.MODE_BATTLE, level=1, n_agents=3, teams=c0.teams,
                             auto_reset=0, player_tokens=rich, ind=1, agent_tokens=c0.records)
    shadow = Oracle(config.Workload("shadow", cfg, m, portal))
    shadow.reset((C.c_uint64 * 1)(c0.tb), (C.c_uint64 * 1)(c0.serial))
    rngs = {k: np.random.RandomState(2000 + k) for k in (0, 2)}
    for it in range(9):
        cmd = [ord("+")] * 3
        cmd[1] = ord(ref_cmds[it])
        for k in (0, 2):
            if it < quits[k]:
                cmd[k] = ord(abi.BENCH_COMMANDS[rngs[k].randint(0, 28)])
            elif it == quits[k]:
                cmd[k] = ord("_")
        shadow.step(np.array(cmd, dtype=np.uint8))
        assert bool(shadow.done()[0]) == ref_ended[it], "iteration %d" % it
    assert shadow.dump(0).hdr.outcome == abi.WON
