#!/usr/bin/env python3
"""ref_slices.py — TEST INFRASTRUCTURE: build oracle/_ref/libsf_refslice.so from the reference checkout.

    python oracle/ref_slices.py [--ref /root/reference] [--quiet]

The reference client cannot be built here as a whole (every include chain reaches SFML through basic.hpp:41, and
stand-in headers are not allowed), but these ranges of it need nothing beyond the standard library:

    StrikeForce-client/random.hpp:27-77       _srand/_rand/binpow/make_p
    StrikeForce-client/Item.hpp:27-194        Item/ConsumableItem/Weapon/Bullet, the item tables, download_items()
    StrikeForce-client/Character.hpp:29-47    compute_damage, wdx/wdy
    StrikeForce-client/Character.hpp:225-287  class Character (hit)
    StrikeForce-client/Character.hpp:832-871  class Zombie, gen_zombie

Each range is cut out by line number into a temporary directory; the first and last line of every range is compared
with an anchor text, so that a reference whose lines have moved fails loudly instead of compiling something else.
The text is not edited.  oracle/ref_slices_wrap.cpp (ours) includes the cuts and exports C entry points; the result
is oracle/_ref/libsf_refslice.so.  The temporary directory is deleted: no reference source stays in the repo, and the
.so is git-ignored.  Without a checkout nothing is built and the script says so (exit code 0): the GPU box only ever
sees the prebuilt file.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))

# (output name, file, first line, last line, text the first line must contain, text the last line must contain)
SLICES = [
    ("slice_random.inc", "StrikeForce-client/random.hpp", 27, 77, "namespace Environment::Random{", "}"),
    ("slice_item.inc", "StrikeForce-client/Item.hpp", 27, 194, "namespace Environment::Item{", "}"),
    ("slice_char_cd.inc", "StrikeForce-client/Character.hpp", 29, 47, "int compute_damage(int x, int y){",
     "const int wdx[4] = {1, 0, -1, 0}, wdy[4] = {0, 1, 0, -1};"),
    ("slice_char_base.inc", "StrikeForce-client/Character.hpp", 225, 287, "class Character{", "};"),
    ("slice_char_zomb.inc", "StrikeForce-client/Character.hpp", 832, 871, "class Zombie: public Character{", "}"),
]
# lines that must sit right before / after a range, so that the range is known to be the whole construct
CONTEXT = [
    ("StrikeForce-client/Character.hpp", 27, "namespace Environment::Character{"),
    ("StrikeForce-client/Character.hpp", 289, "class Human: public Character{"),
    ("StrikeForce-client/Character.hpp", 866, "void gen_zombie(Zombie &z, bool super, std::vector<int> cor_, std::string name = \"\"){"),
    ("StrikeForce-client/Character.hpp", 873, "void gen_human("),
    ("StrikeForce-client/random.hpp", 54, "int _rand(){"),
    ("StrikeForce-client/Item.hpp", 179, "void download_items(){"),
]


def build(ref="/root/reference", quiet=False):
    out_dir = os.path.join(HERE, "_ref")
    out = os.path.join(out_dir, "libsf_refslice.so")
    if not os.path.isfile(os.path.join(ref, "StrikeForce-client", "random.hpp")):
        if not quiet:
            print("no reference checkout at %s: oracle/_ref/libsf_refslice.so left as is" % ref)
        return None
    cache = {}

    def lines_of(rel):
        if rel not in cache:
            with open(os.path.join(ref, rel), encoding="utf-8", errors="replace", newline="") as f:
                cache[rel] = f.read().split("\n")
        return cache[rel]

    def check(rel, no, text):
        got = lines_of(rel)[no - 1].strip()
        if text not in got or (len(text) <= 2 and got != text):
            raise SystemExit("ref_slices: %s:%d is %r, expected %r — the reference's lines have moved" % (rel, no, got, text))

    for rel, no, text in CONTEXT:
        check(rel, no, text)
    tmp = tempfile.mkdtemp(prefix="sf_refslice_")
    try:
        for name, rel, first, last, a0, a1 in SLICES:
            check(rel, first, a0)
            check(rel, last, a1)
            with open(os.path.join(tmp, name), "w", encoding="utf-8", newline="") as f:
                f.write("\n".join(lines_of(rel)[first - 1:last]) + "\n")
        os.makedirs(out_dir, exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O2", "-w", "-fPIC", "-shared", "-I", tmp,
               os.path.join(HERE, "ref_slices_wrap.cpp"), "-o", out]
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not quiet:
        print("built oracle/_ref/libsf_refslice.so from %s (random.hpp:27-77, Item.hpp:27-194, Character.hpp:29-47,"
              "225-287,832-871)" % ref)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.environ.get("REF", "/root/reference"))
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    build(a.ref, a.quiet)
    sys.exit(0)
