"""The oracle against the reference's own code, compiled here: oracle/_ref/libsf_refslice.so = random.hpp:27-77,
Item.hpp:27-194, Character.hpp:29-47,225-287,832-871 cut out by line range and compiled unmodified with standard
headers only (oracle/ref_slices.py; built by __graft_entry__.build() where /root/reference exists).  This pins SURVEY
§8 rows a1 (RNG), a2 (items, Bullet::shot/expire), a3 (compute_damage) and the Character::hit / Zombie parts of a4/a5
on the reference itself.  Skipped where the file was never built."""
import ctypes as C
import json
import os
import random

import pytest

import oracle_lib
import reflib
from strikeforce_amd import config

pytestmark = pytest.mark.skipif(reflib.lib() is None, reason="oracle/_ref/libsf_refslice.so not built (no reference checkout)")


def test_rand_against_reference_16_seeds_4096_draws():
    L = oracle_lib.lib()
    rnd = random.Random(7)
    seeds = [(1700000000 + i, 123456789) for i in (0, 1, 2, 4095)]
    seeds += [(0, 0), (10 ** 18 - 1, 10 ** 18 - 1), (1, 10 ** 17), (10 ** 17, 1)]
    seeds += [(rnd.randrange(10 ** 18), rnd.randrange(10 ** 18)) for _ in range(8)]
    n = 4096
    for tb, sr in seeds:
        want, st = reflib.rand(tb, sr, n)
        got = (C.c_int32 * n)()
        L.sfo_kat_rand(tb, sr, n, got)
        assert list(got) == want, (tb, sr)
        gst = (C.c_int64 * 19)()
        L.sfo_kat_rand_state(tb, sr, n, gst)
        assert list(gst) == st, (tb, sr)


def test_rand_across_the_jomle_wrap():
    """binpow's exponent is ++jomle mod 65536 (RN:44,58): compare past the wrap of the counter."""
    L = oracle_lib.lib()
    n = 70000
    want, st = reflib.rand(1700000007, 123456789, n)
    got = (C.c_int32 * n)()
    L.sfo_kat_rand(1700000007, 123456789, n, got)
    assert list(got) == want
    gst = (C.c_int64 * 19)()
    L.sfo_kat_rand_state(1700000007, 123456789, n, gst)
    assert list(gst) == st and st[18] == 18 + 1024 + n


def test_compute_damage_against_reference():
    L, R = oracle_lib.lib(), reflib.lib()
    for y in (1, 2, 3, 4, 7, 8, 64, 100, 1000):
        for x in range(0, 20001, 1 if y in (1, 100) else 7):
            assert L.sfo_kat_compute_damage(x, y) == R.ref_compute_damage(x, y), (x, y)


def test_item_tables_against_download_items():
    it = reflib.items()
    if it is None:
        pytest.skip("no reference client directory (Items/*.txt) on this machine")
    assert [tuple(r[3:6]) for r in it["cons"]] == config._CONS
    assert [tuple(r[3:7]) for r in it["throw"]] == config._THROW
    assert [tuple(r[3:7]) for r in it["weapon"]] == config._WEAPON
    # the committed fixture is this very table
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    assert gold["items"] == it


def test_bullet_shot_expire_against_reference():
    L = oracle_lib.lib()
    rnd = random.Random(11)
    for _ in range(2000):
        cor0 = [rnd.randrange(3), rnd.randrange(1, 100), rnd.randrange(1, 100)]
        way = rnd.randrange(1, 5)
        steps = rnd.randrange(0, 120)
        cor1 = list(cor0)
        cor1[1] += (1, 0, -1, 0)[way - 1] * steps
        cor1[2] += (0, 1, 0, -1)[way - 1] * steps
        dmg, eff, rng, owner = rnd.randrange(0, 500), rnd.randrange(-300, 50), rnd.choice([1, 2, 7, 100]), rnd.randrange(0, 64)
        out = (C.c_int32 * 6)()
        L.sfo_kat_bullet((C.c_int32 * 3)(*cor0), (C.c_int32 * 3)(*cor1), way, dmg, eff, rng, owner, out)
        assert list(out) == reflib.bullet(cor0, cor1, way, dmg, eff, rng, owner)


def test_character_hit_and_zombie_against_reference():
    L = oracle_lib.lib()
    rnd = random.Random(13)
    for _ in range(500):
        hp, md, dmg, eff = rnd.randrange(-50, 2000), rnd.randrange(-300, 400), rnd.randrange(0, 600), rnd.randrange(-300, 60)
        out = (C.c_int32 * 2)()
        L.sfo_kat_character_hit(hp, md, dmg, eff, out)
        assert list(out) == reflib.character_hit(hp, md, dmg, eff)
        sup, hits, way = rnd.randrange(2), rnd.randrange(0, 4), rnd.randrange(4)
        cor = [rnd.randrange(3), rnd.randrange(1, 29), rnd.randrange(1, 99)]
        z = (C.c_int32 * 11)()
        L.sfo_kat_zombie(sup, (C.c_int32 * 3)(*cor), hits, dmg, eff, way, z)
        assert list(z) == reflib.zombie(sup, cor, hits, dmg, eff, way)
