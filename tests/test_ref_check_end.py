"""SURVEY §8 row a19, `check_end` (gameplay.hpp:1102-1229), pinned on the reference's own function: oracle/ref_tick.py
keeps check_end in the head-less build — its comparisons, bookkeeping and return value — and blanks only the statements
that draw the end screen and wait for the space key.  The driver calls it where play() does (`if(check_end()) break;`,
gameplay.hpp:1450) and reports what it returned; here whole games are played to their end side by side with the oracle:
the state after every step, and the end check at every loop top, must agree — up to and including the step at which
the reference says the game is over.  Runs were picked by playing the oracle alone under the random-action agent until
one ended each way.  (The Timer mode's clock is time(0) - tb in the reference, gameplay.hpp:1145, and the frame count
here: DESIGN §8; its end is not comparable under a fixed tb.  The online branch: tests/test_lockstep_server.py.)"""
import ctypes as C

import numpy as np
import pytest

import ref_cases
import reftick
from oracle_lib import Oracle
from strikeforce_amd import abi, config

pytestmark = pytest.mark.skipif(not reftick.available(), reason="oracle/_ref/sf_ref_tick not built (no reference checkout)")

SERIAL = 123456789


def play_to_the_end(w, player, tb, cmd_seed, max_steps):
    o = Oracle(w)
    r = reftick.RefTick(w, player, native_caps=False)
    try:
        o.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(SERIAL))
        r.reset(tb, SERIAL)
        cmds, _ = config.bench_commands(1, 1, max_steps, seed0=cmd_seed)
        for s in range(max_steps):
            o.step(cmds[s])
            r.step(cmds[s, 0, :1])
            od = o.dump(0)
            rd = r.dump()
            assert not r.over
            d = reftick.first_difference(rd, reftick.arrays_of(od))
            assert d is None, "step %d: %s" % (s, d)
            assert r.ended == bool(od.hdr.done), "step %d: the reference's check_end() says %s, ours %s" % (s, r.ended, od.hdr.done)
            if od.hdr.done:
                return s + 1, od.hdr
        raise AssertionError("the game did not end in %d steps" % max_steps)
    finally:
        r.close()
        o.close()


def solo(which, player):
    """A BASELINE world played as a Solo game by one player; the exit pool as large as the bullet pool (the reference
    pools exits by B, gameplay.hpp:51-53, and a level-10 account carries ten portals)."""
    w = ref_cases.baseline(which, player)
    w.cfg.mode = abi.MODE_SOLO
    w.cfg.cap_portals = w.cfg.cap_bullets
    return w


def test_solo_won_by_five_kills():
    """`level * 5 <= kills && mode == "Solo"` (gameplay.hpp:1179): the level-10 account on configs[2]'s world (up to 8
    humans and 24 zombies), level 1: the fifth kill ends the game at step 2346 — not one loop top earlier."""
    steps, hdr = play_to_the_end(solo("C3", ref_cases.RICH), ref_cases.RICH, 1700000220, 720, 2400)
    assert (steps, hdr.outcome, hdr.kills) == (2346, abi.WON, 5)


def test_squad_won_by_ten_team_kills_and_dead_rivals():
    """`level * 10 <= teams_kills && rivals_are_dead() && mode == "Squad"` (gameplay.hpp:1204-1229) on the reference's own
    function, on its shipped maps: the scripted game of tests/golden/squad_win_commands.txt (tests/tools/plan_squad_win.py:
    through the entrances of floors 1 and 2 to the five opponents on floor 3, a punch each, then whatever comes near).
    The reference's check_end() says "go on" at every loop top before — also at the one behind the fifth opponent's
    death, with `rivals_are_dead()` true and teams_kills short of ten — and "won" at the loop top behind the tenth kill."""
    head, cmds = ref_cases.squad_win_commands()
    w = ref_cases.native(abi.MODE_SQUAD, 1, ref_cases.RICH, maps="shipped")
    o = Oracle(w)
    r = reftick.RefTick(w, ref_cases.RICH)
    try:
        o.reset((C.c_uint64 * 1)(1700007777), (C.c_uint64 * 1)(424242))
        r.reset(1700007777, 424242)
        rivals_dead_at = None
        for s, ch in enumerate(cmds):
            o.step(np.array([ord(ch)], dtype=np.uint8))
            r.step(ch)
            od, rd = o.dump(0), r.dump()
            assert not r.over
            d = reftick.first_difference(rd, reftick.arrays_of(od))
            assert d is None, "step %d: %s" % (s, d)
            assert r.ended == bool(od.hdr.done), "step %d: the reference's check_end() says %s, ours %s" % (s, r.ended, od.hdr.done)
            if rivals_dead_at is None and r.rivals_are_dead():
                rivals_dead_at = (s, od.hdr.teams_kills)
            if od.hdr.done:
                break
        assert s == len(cmds) - 1 and od.hdr.outcome == abi.WON and od.hdr.teams_kills == 10
        assert rivals_dead_at is not None and rivals_dead_at[0] < s and rivals_dead_at[1] < 10, rivals_dead_at
    finally:
        r.close()
        o.close()


@pytest.mark.parametrize("which,player,tb,seed,steps", [("C2", config.HUMAN_ENEMY_TOKENS, 1700000108, 508, 147),
                                                          ("C1", config.HUMAN_TOKENS, 1700000104, 504, 71)])
def test_solo_lost_by_death(which, player, tb, seed, steps):
    """`hum[ind].get_Hp() <= 0` (gameplay.hpp:1131): the zombies get the player."""
    n, hdr = play_to_the_end(solo(which, player), player, tb, seed, steps + 50)
    assert (n, hdr.outcome) == (steps, abi.DIED)
