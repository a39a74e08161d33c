"""How many zombies / humans / bullets / exits a whole game of the REFERENCE holds (oracle/_ref/sf_ref_tick, its own pools
of 9000, gameplay.hpp:37): the numbers behind the device's slot caps.  Checker tool: this container only.

    python tests/tools/population_curve.py --mode timer --level 3 [--steps N] [--every 250]

Plays the shipped maps (tests/golden/maps) with the level-10 account and the suite's random-action commands and prints
the live populations every `--every` steps as JSON lines."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import ref_cases  # noqa: E402
import reftick  # noqa: E402
from strikeforce_amd import abi, config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="timer")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--steps", type=int, default=0, help="0: the Timer game's own length, level * 3750 steps")
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--tb", type=int, default=1700000999)
    ap.add_argument("--serial", type=int, default=55555)
    ap.add_argument("--cmd-seed", type=int, default=31)
    a = ap.parse_args()
    mode = {"timer": abi.MODE_TIMER, "solo": abi.MODE_SOLO, "squad": abi.MODE_SQUAD}[a.mode]
    steps = a.steps or a.level * 3750
    w = ref_cases.native(mode, a.level, ref_cases.RICH, maps="shipped")
    w.cfg.cap_humans, w.cfg.cap_zombies, w.cfg.cap_bullets, w.cfg.cap_portals = 2048, 2048, 2048, 256
    r = reftick.RefTick(w, ref_cases.RICH)
    r.reset(a.tb, a.serial)
    cmds, _ = config.bench_commands(1, 1, steps, seed0=a.cmd_seed)
    peak = {"zombies": 0, "humans": 0, "bullets": 0, "portals": 0}
    for s in range(steps):
        r.step(cmds[s, 0, :1])
        if (s + 1) % a.every == 0 or s + 1 == steps:
            d = r.dump()
            now = {"zombies": int(d["zombies"][:, 0].sum()), "humans": int(d["humans"][:, 0].sum()),
                   "bullets": int(d["bullets"][:, 0].sum()), "portals": int(d["portals"][:, 0].sum())}
            hi = {"zombie_slot": int(np.nonzero(d["zombies"][:, 0])[0].max(initial=-1)),
                  "human_slot": int(np.nonzero(d["humans"][:, 0])[0].max(initial=-1)),
                  "bullet_slot": int(np.nonzero(d["bullets"][:, 0])[0].max(initial=-1))}
            for k in peak:
                peak[k] = max(peak[k], now[k])
            print(json.dumps({"step": s + 1, **now, **hi, "over": r.over, "player_hp": int(d["humans"][0, 9])}), flush=True)
            if r.ended and mode != abi.MODE_TIMER:
                break
    print(json.dumps({"peak": peak}))
    r.close()


if __name__ == "__main__":
    main()
