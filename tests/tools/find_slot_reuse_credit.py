"""Search for the scenario of tests/test_order_scenarios.py::test_a_hit_is_credited_to_whoever_holds_the_owners_slot_now
(SURVEY App. E-11): prints (tb, shoot step) for which, on the two-corridor world of that test and under its script,
  * the NPC human of the first loop top (frame 1) is put into the right-hand corridor, above its lower end,
  * the player's one shot kills it before step 50 while some of its own bullets are still flying down the corridor,
  * the spawn of step 50 (frame 101, gameplay.hpp:1448-1449) finds a free cell and takes the dead NPC's slot, 1,
  * and one of the dead NPC's bullets then hits the player: the oracle's test-aid counter `credit_slot_reused` fires.
Run by hand (it needs a minute or two); the test carries what it found."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

import oracle_lib  # noqa: E402
import ref_cases  # noqa: E402
import test_order_scenarios as T  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from strikeforce_amd import abi  # noqa: E402


def first_spawn_cell(tb):
    """where the frame-1 human spawn lands: draws 6, 7, 8 (chest, zombie, human: three coordinate draws each)"""
    k = T.kat(tb, 9)
    return k[6] % 3, k[7] % T.ROWS, k[8] % T.COLS


def main():
    w = T.slot_reuse_world()
    found = 0
    for tb in range(1700000000, 1700400000):
        f, r, c = first_spawn_cell(tb)
        if not (f == 0 and c == 3 and 2 <= r <= 14):
            continue
        for shoot in range(34, 49):
            o = Oracle(w)
            o.reset((C.c_uint64 * 1)(tb), (C.c_uint64 * 1)(T.SERIAL))
            script = T.slot_reuse_script(shoot)
            ok = None
            for s, ch in enumerate(script):
                o.step(np.frombuffer(ch.encode(), dtype=np.uint8))
                d = o.dump(0)
                if d.hdr.done:
                    break
                if s == 49 and not (d.humans[1].alive and d.humans[1].hp == 1000 and d.humans[1].kills == 0):
                    break  # no fresh NPC in slot 1 at the loop top behind step 49
                if o.events().get("credit_slot_reused"):
                    ok = s
                    break
            o.close()
            if ok is not None:
                print("tb", tb, "shoot at step", shoot, "first credited hit in step", ok, flush=True)
                found += 1
                break
        if found >= 3:
            return


if __name__ == "__main__":
    main()
