"""Writes tests/golden/squad_win_commands.txt: a scripted Squad game on the reference's shipped maps that ENDS WON —
`level * 10 <= teams_kills && rivals_are_dead() && mode == "Squad"` (gameplay.hpp:1204-1229), the one branch of check_end
no random-action run ever reached.  The player (the level-10 account: one punch kills) walks from its start cell
(0,3,1) through the '^' entrances of floors 1 and 2 (map/floor1-2.txt: exits 2 and 4) to the five opponents standing on
floor 3's row 1 (gameplay.hpp:1887-1897), punches each, then hunts whatever comes near until the team has ten kills.

The commands are planned against this repo's ORACLE (a breadth-first walk to the cell next to the nearest target, a turn,
a punch), one per step; tests/test_ref_check_end.py then plays the same string on the REFERENCE (oracle/_ref/sf_ref_tick)
in lock step and asserts the reference's own check_end() says "won" at the same loop top.  Checker tool.

    python tests/tools/plan_squad_win.py
"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import ref_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from strikeforce_amd import abi  # noqa: E402

TB, SERIAL, LEVEL = 1700007777, 424242, 1
DR, DC = (1, 0, -1, 0), (0, 1, 0, -1)  # way 1..4 = down, right, up, left (gameplay.hpp:43, Character.hpp:47)
MOVE = "sdwa"


def plan(max_steps=4000):
    w = ref_cases.native(abi.MODE_SQUAD, LEVEL, ref_cases.RICH, maps="shipped")
    o = Oracle(w)
    o.reset((C.c_uint64 * 1)(TB), (C.c_uint64 * 1)(SERIAL))
    F, N, M = w.cfg.floors, w.cfg.rows, w.cfg.cols
    chars = bytes(w._map.raw).decode("ascii")
    portal = list(w._portal)
    exits = [i for i, c in enumerate(chars) if c == "O"]  # exit k = the k-th 'O' in scan order (gameplay.hpp:1265-1270)
    cmds = []
    for _ in range(max_steps):
        d = o.dump(0)
        if d.hdr.done:
            break
        me = d.humans[0]
        flags = d.flags
        occupied = {}
        for h in d.humans:
            if h.alive:
                occupied[(h.f, h.r, h.c)] = ("h", h.team)
        for z in d.zombies:
            if z.alive:
                occupied[(z.f, z.r, z.c)] = ("z", 0)
        rivals = [(h.f, h.r, h.c) for h in d.humans if h.alive and h.team not in (0, me.team)]
        prey = rivals or [k for k, v in occupied.items() if v[0] == "z" or (v[0] == "h" and v[1] != me.team)]
        pos = (me.f, me.r, me.c)
        # an enemy on a neighbouring cell: face it, punch it
        near = [(w_, (me.f, me.r + DR[w_ - 1], me.c + DC[w_ - 1])) for w_ in (1, 2, 3, 4)]
        hit = [w_ for w_, q in near if q in occupied and (occupied[q][0] == "z" or occupied[q][1] != me.team) and (not rivals or q in rivals or occupied[q][0] == "z")]
        if hit:
            want = me.way if me.way in hit else hit[0]
            if want == me.way:
                cmds.append("z")
            else:  # turn_l: way 4 -> 1 else + 1 ('q'); turn_r: 1 -> 4 else - 1 ('e')   Character.hpp:745-759
                cmds.append("q" if (want - me.way) % 4 in (1, 2) else "e")
            o.step(np.array([ord(cmds[-1])], dtype=np.uint8))
            continue

        def walkable(q):
            f, r, c = q
            if not (0 <= r < N and 0 <= c < M):
                return False
            fl = flags[(f * N + r) * M + c]
            return not (fl & abi.CELL_WALL) and q not in occupied

        # breadth-first over (floor, row, col); stepping onto an entrance lands on its exit
        goal = set()
        for t in prey:
            for k in range(4):
                q = (t[0], t[1] + DR[k], t[2] + DC[k])
                if walkable(q) or q == pos:
                    goal.add(q)
        prev = {pos: None}
        queue = collections.deque([pos])
        found = None
        while queue and found is None:
            cur = queue.popleft()
            if cur in goal and cur != pos:
                found = cur
                break
            for k in range(4):
                q = (cur[0], cur[1] + DR[k], cur[2] + DC[k])
                if not walkable(q):
                    continue
                ci = (q[0] * N + q[1]) * M + q[2]
                land = q
                if chars[ci] in "^v":  # teleport(): to the exit if it shows 'O' (gameplay.hpp:517-530)
                    e = exits[portal[ci]]
                    land = (e // (N * M), (e // M) % N, e % M)
                    if land in occupied:
                        continue
                if land not in prev:
                    prev[land] = (cur, k)
                    queue.append(land)
        if found is None:  # nothing to reach right now (a zombie will come): wait
            cmds.append("+")
        else:
            cur = found
            while prev[cur][0] != pos:
                cur = prev[cur][0]
            cmds.append(MOVE[prev[cur][1]])
        o.step(np.array([ord(cmds[-1])], dtype=np.uint8))
    d = o.dump(0)
    assert d.hdr.done and d.hdr.outcome == abi.WON, (d.hdr.done, d.hdr.outcome, d.hdr.teams_kills, len(cmds))
    return "".join(cmds), d.hdr


if __name__ == "__main__":
    cmds, hdr = plan()
    path = os.path.join(ROOT, "tests", "golden", "squad_win_commands.txt")
    with open(path, "w") as f:
        f.write("# tests/tools/plan_squad_win.py: tb %d serial %d level %d; %d steps, teams_kills %d, kills %d\n" % (
            TB, SERIAL, LEVEL, len(cmds), hdr.teams_kills, hdr.kills))
        f.write(cmds + "\n")
    print("wrote", path, len(cmds), "steps; teams_kills", hdr.teams_kills, "kills", hdr.kills, "frame", hdr.frame)
