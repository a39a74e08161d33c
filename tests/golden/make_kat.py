#!/usr/bin/env python3
"""Regenerates tests/golden/kat.json from the reference itself: oracle/_ref/libsf_refslice.so is random.hpp:27-77,
Item.hpp:27-194 and Character.hpp:29-47,225-287,832-871 compiled unmodified (oracle/ref_slices.py).  Run in the build
container, where /root/reference exists:

    python oracle/ref_slices.py && python tests/golden/make_kat.py

The fixture is data (inputs and the reference's outputs); tests/test_oracle_kat.py holds the oracle to it everywhere,
tests/test_ref_slices.py compares oracle and reference directly, over far more inputs, wherever the .so exists."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflib  # noqa: E402

SEEDS = [(1700000000, 123456789), (0, 0), (1771155561, 1073741823), (1700004095, 123456789), (999999999999999999, 1),
         (1, 999999999999999999), (123456789012345678, 876543210987654321), (1700000123, 987654321)]


def main():
    if reflib.lib() is None:
        raise SystemExit("oracle/_ref/libsf_refslice.so is missing: run oracle/ref_slices.py where /root/reference exists")
    rand = []
    for tb, sr in SEEDS:
        first, _ = reflib.rand(tb, sr, 16)
        _, st = reflib.rand(tb, sr, 4096)
        rand.append({"tb": tb, "serial": sr, "first": first, "state_after_4096": st})
    xs = [0, 1, 2, 3, 7, 8, 9, 15, 16, 26, 27, 63, 64, 99, 100, 105, 145, 150, 200, 225, 275, 999, 1000, 1045, 4095,
          4096, 9999, 15000, 20000]
    cd = {str(y): {str(x): reflib.lib().ref_compute_damage(x, y) for x in xs} for y in (1, 2, 3, 4, 64, 100)}
    bullets = []
    for cor0, cor1, rng in [((0, 5, 5), (0, 5, 5), 1), ((0, 5, 5), (0, 5, 5), 2), ((0, 5, 5), (0, 6, 5), 2),
                            ((0, 5, 5), (0, 5, 103), 100), ((0, 5, 5), (0, 5, 104), 100), ((1, 9, 9), (1, 3, 9), 7)]:
        bullets.append({"cor0": cor0, "cor1": cor1, "way": 2, "damage": 150, "effect": -55, "range": rng, "owner": 3,
                        "out": reflib.bullet(cor0, cor1, 2, 150, -55, rng, 3)})
    zombies = []
    for super_, hits, dmg, eff, way in [(0, 0, 0, 0, 0), (1, 0, 0, 0, 3), (0, 1, 150, -55, 1), (1, 3, 100, -90, 2),
                                        (0, 2, 250, 20, 0)]:
        zombies.append({"super": super_, "cor": [0, 5, 6], "hits": hits, "damage": dmg, "effect": eff, "way": way,
                        "out": reflib.zombie(super_, (0, 5, 6), hits, dmg, eff, way)})
    out = {
        "_provenance": "Outputs of the reference's own code: random.hpp:27-77, Item.hpp:27-194 and Character.hpp:29-47,"
                       "225-287,832-871 compiled unmodified into oracle/_ref/libsf_refslice.so (oracle/ref_slices.py)"
                       " and called by tests/golden/make_kat.py in the build container.",
        "rand": rand,
        "compute_damage": cd,
        "items": reflib.items(),
        "bullet": bullets,
        "character_hit": [{"hp": 400, "mindamage": 100, "damage": 150, "effect": -55,
                           "out": reflib.character_hit(400, 100, 150, -55)},
                          {"hp": 10, "mindamage": -5, "damage": 20, "effect": 20, "out": reflib.character_hit(10, -5, 20, 20)}],
        "zombie": zombies,
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote tests/golden/kat.json")


if __name__ == "__main__":
    main()
