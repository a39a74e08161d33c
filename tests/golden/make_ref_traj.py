#!/usr/bin/env python3
"""Generates tests/golden/ref_traj.json from the REFERENCE's own tick path: oracle/_ref/sf_ref_tick (the client's
headers compiled head-less with only the SFML / keyboard / menu member functions blanked, oracle/ref_tick.py).  For every
run of tests/ref_cases.GOLDEN_CASES: the 64-bit state digest (include/strikeforce.h sf_state_digest's definition,
computed here from the REFERENCE's state dump) after the reset and after every step, and the reference's observation
(gameplay::bot, bots/bot-0.5/Custom.hpp:137-159: its non-zero floats) every 200 steps.  The oracle runs alongside and
must equal the reference's whole state at every step (the same check as tests/test_ref_tick.py), so a file can only be
written from a run in which the two agree.  Needs the reference checkout (build container only); the vectors travel.

    python tests/golden/make_ref_traj.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
import ref_cases  # noqa: E402
import reftick  # noqa: E402
from strikeforce_amd import abi, config  # noqa: E402


def digest_of(ref, like, cfg):
    """sf_digest_from_dump (oracle/sf_oracle.c, the definition sf_state_digest shares) over the REFERENCE's dump; the
    words the reference has no counterpart for (a human's record index; done / outcome, 0 while the game runs) come
    from `like`, the oracle's dump of the same moment."""
    L = oracle_lib.lib()
    hdr = abi.ArenaHdr()
    (hdr.frame, hdr.kills, hdr.teams_kills, hdr.loot, hdr.chests, hdr.jomle, hdr.steps) = [int(x) for x in ref["hdr"][:7]]
    for i in range(18):
        hdr.rng[i] = int(ref["hdr"][7 + i])
    hdr.done, hdr.outcome = like.hdr.done, like.hdr.outcome
    hum = ref["humans"].copy()
    hum[:, 3] = [h.profile for h in like.humans]
    tabs = [np.ascontiguousarray(t, dtype=np.int32) for t in (hum, ref["zombies"], ref["bullets"], ref["portals"])]
    flags = np.ascontiguousarray(ref["flags"], dtype=np.uint8)
    dmg, pidx = np.ascontiguousarray(ref["dmg"], np.int32), np.ascontiguousarray(ref["pidx"], np.int32)
    L.sf_digest_from_dump.restype = C.c_uint64
    L.sf_digest_from_dump.argtypes = [C.c_int] * 5 + [C.c_void_p] * 8
    return int(L.sf_digest_from_dump(cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.cap_portals, flags.size,
                                     C.addressof(hdr), tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data,
                                     tabs[3].ctypes.data, flags.ctypes.data, dmg.ctypes.data, pidx.ctypes.data))


def sparse(obs):
    u = obs.view(np.uint32)
    return " ".join("%x:%x" % (int(i), int(u[i])) for i in np.flatnonzero(u))


def run_case(name):
    make, player, tb, serial, steps, cmd_seed, native_caps = ref_cases.GOLDEN_CASES[name]
    w = make()
    o = oracle_lib.Oracle(w)
    r = reftick.RefTick(w, player, native_caps=native_caps)
    o.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(serial))
    r.reset(tb, serial)
    scripted = isinstance(cmd_seed, str)
    if scripted:
        cmds = np.frombuffer(cmd_seed.encode(), dtype=np.uint8).reshape(steps, 1, 1)
    else:
        cmds, _ = config.bench_commands(1, 1, steps, seed0=cmd_seed)
    digests, obs = [], {}
    for s in range(steps + 1):
        od, rd = o.dump(0), r.dump()
        assert not r.over, "%s step %d: the reference holds more entities than the configuration's pools" % (name, s)
        d = reftick.first_difference(rd, reftick.arrays_of(od))
        assert d is None, "%s step %d: %s" % (name, s, d)
        dg = digest_of(rd, od, w.cfg)
        assert dg == int(o.digest()[0])
        digests.append("%016x" % dg)
        if s % ref_cases.OBS_EVERY == 0:
            obs[str(s)] = sparse(r.observe(0))
        if od.hdr.done:  # (the state at the loop top that ended the game is the last one compared)
            assert r.ended
            break
        if s < steps:
            o.step(cmds[s])
            r.step(cmds[s, 0, :1])
    r.close()
    return {"tb": tb, "serial": serial, "command_seed": None if scripted else cmd_seed, "commands": cmd_seed if scripted else None,
            "ended": bool(od.hdr.done), "outcome": int(od.hdr.outcome), "reference_build": "native" if native_caps else "patched dimensions (gameplay.hpp:37)",
            "digests": digests, "obs_nonzero": obs}


def run_squad_agents(steps=300, every=60):
    """The reference built with USE_AGENT_IN_SQUAD_NPCS (ten Agents, Squad on the shipped world), scripted with
    tests/test_squad_agents_example.policy: digests per step, and every `every` steps the observations the reference
    handed its agents — agent 0's at the loop top, the others' inside human_action (gameplay.hpp:988-999) — plus which
    humans still had an Agent after the step."""
    import test_squad_agents_example as X
    w = ref_cases.native(abi.MODE_SQUAD, 2, ref_cases.RICH, maps="shipped")
    w.cfg.n_agents = 10
    o = oracle_lib.Oracle(w)
    r = reftick.RefTick(w, ref_cases.RICH, agents=True, squad_agents=True)
    tb, serial = 1700000000, 123456789
    o.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(serial))
    r.reset(tb, serial)
    digests, obs, alive = [], {}, []
    for s in range(steps):
        od, rd = o.dump(0), r.dump()
        assert reftick.first_difference(rd, reftick.arrays_of(od)) is None
        digests.append("%016x" % digest_of(rd, od, w.cfg))
        chars = "".join(X.ACTS[X.policy(g, s)] for g in range(10))
        o.step(np.frombuffer(chars.encode(), dtype=np.uint8))
        r.step(chars)
        asked = sorted({c[0] for c in r.calls() if c[1] == "P"})
        if s % every == 0:
            obs[str(s)] = {str(g): sparse(r.last_obs(g)) for g in asked}
        r.dump()
        alive.append("".join(str(x) for x in r.active_agents[:10]))
        if o.done()[0]:
            break
    od, rd = o.dump(0), r.dump()
    digests.append("%016x" % digest_of(rd, od, w.cfg))
    r.close()
    return {"tb": tb, "serial": serial, "reference_build": "native, -DUSE_AGENT_IN_SQUAD_NPCS", "digests": digests,
            "agent_obs_nonzero": obs, "agent_alive_after_step": alive}


if __name__ == "__main__":
    if not reftick.available():
        raise SystemExit("oracle/_ref/sf_ref_tick is not built: needs the reference checkout (python oracle/ref_tick.py)")
    data = {"_generator": "tests/golden/make_ref_traj.py: digests and observations of oracle/_ref/sf_ref_tick = the reference's "
                          "own gameplay.hpp / Character.hpp / Item.hpp / random.hpp / Custom.hpp compiled head-less (oracle/ref_tick.py)",
            "cases": {}}
    for name in ref_cases.GOLDEN_CASES:
        data["cases"][name] = run_case(name)
        print(name, len(data["cases"][name]["digests"]) - 1, "steps,", len(data["cases"][name]["obs_nonzero"]), "observations")
    data["squad_agents"] = run_squad_agents()
    print("squad_agents", len(data["squad_agents"]["digests"]) - 1, "steps")
    with open(os.path.join(HERE, "ref_traj.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print("wrote ref_traj.json")
