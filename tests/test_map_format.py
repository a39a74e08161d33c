"""The reference's map text format (map/floorK.txt, parser gameplay.hpp:1249-1274): parse, format, load a directory.
The files written here are synthetic; pointing load_reference_maps() at a reference checkout's map/ works the same."""
import os

import pytest

from oracle_lib import Oracle
from emu_lib import Emu
from strikeforce_amd import abi, config


def test_parse_matches_the_reference_reader():
    # "#^ 2 ^ 2 #" style rows: entrances carry their exit number, whitespace is free-form
    text = "###^ 2 ^ 2 ###\n#..O...#\n#.v 0 ....#\n########\n"
    chars, portal = config.parse_floor_text(text, 4, 8)
    assert chars == "###^^###" "#..O...#" "#.v....#" "########"
    assert portal[3] == 2 and portal[4] == 2 and portal[8 * 2 + 2] == 0
    assert sum(1 for p in portal if p >= 0) == 3
    assert config.parse_floor_text(config.format_floor_text(chars, portal, 4, 8), 4, 8) == (chars, portal)


def test_three_floor_directory_round_trip_and_parity(tmp_path):
    rows, cols = 30, 100  # the reference's N, M (gameplay.hpp:37)
    m, p = config.three_floor_map(rows, cols, wall_p=0.06, map_seed=11)
    per = rows * cols
    for k in range(3):
        (tmp_path / ("floor%d.txt" % (k + 1))).write_text(
            config.format_floor_text(m[k * per:(k + 1) * per].decode(), p[k * per:(k + 1) * per], rows, cols))
    m2, p2 = config.load_reference_maps(str(tmp_path))
    assert m2 == m and p2 == p
    cfg = config.make_config(2, rows, cols, floors=3, H=16, Z=24, B=64, P=16, mode=abi.MODE_SQUAD, level=3, n_agents=2)
    w = config.Workload("native", cfg, m2, p2)
    o, e = Oracle(w), Emu(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), e.reset(tb, sr)
    cmds, _ = config.bench_commands(2, 2, 300)
    o.step_many(cmds), e.step_many(cmds)
    assert (o.digest() == e.digest()).all()
    d = o.dump(0)
    assert {h.f for h in d.humans if h.alive} >= {0, 2}  # Squad: team-mates on floor 0, opponents on floor 2


REF_MAP_DIR = "/root/reference/StrikeForce-client/map"


def test_native_world_on_the_reference_maps_when_present():
    """The shipped 3 x 30 x 100 world (map/floor1-3.txt: 6 exits, '^'/'v' entrances in the top border rows) with the
    reference's level-10 account, Solo and Squad.  Reads the reference checkout, so it only runs where that exists
    (never on the GPU box); the data is not copied."""
    import os
    import pytest
    import fuzz_cases
    if not os.path.isdir(REF_MAP_DIR):
        pytest.skip("no reference checkout")
    m, p = config.load_reference_maps(REF_MAP_DIR)
    assert len(m) == 9000 and m.count(b"O") == 6
    for mode, agents in ((abi.MODE_SOLO, 1), (abi.MODE_SQUAD, 3)):
        cfg = config.make_config(2, 30, 100, floors=3, H=64, Z=64, B=256, P=32, mode=mode, level=4, n_agents=agents,
                                 player_tokens=fuzz_cases.ACCOUNT_1)
        w = config.Workload("native", cfg, m, p)
        o, e = Oracle(w), Emu(w)
        tb, sr = w.seeds()
        o.reset(tb, sr), e.reset(tb, sr)
        cmds, _ = config.bench_commands(2, agents, 500)
        o.step_many(cmds), e.step_many(cmds)
        assert (o.digest() == e.digest()).all()
        assert (o.results() == e.results()).all()


FIXTURE_MAP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps")


def test_committed_map_fixtures_equal_the_checkout():
    """tests/golden/maps/floor1-3.txt are the reference's map data, byte for byte."""
    if not os.path.isdir(REF_MAP_DIR):
        pytest.skip("no reference checkout")
    for k in (1, 2, 3):
        assert open(os.path.join(FIXTURE_MAP_DIR, "floor%d.txt" % k), "rb").read() == \
            open(os.path.join(REF_MAP_DIR, "floor%d.txt" % k), "rb").read()


def _native_world(impl_a, impl_b, steps):
    import fuzz_cases
    m, p = config.load_reference_maps(FIXTURE_MAP_DIR)
    assert len(m) == 9000 and m.count(b"O") == 6
    for mode, agents in ((abi.MODE_SOLO, 1), (abi.MODE_SQUAD, 3)):
        cfg = config.make_config(2, 30, 100, floors=3, H=64, Z=64, B=256, P=32, mode=mode, level=4, n_agents=agents,
                                 player_tokens=fuzz_cases.ACCOUNT_1)
        w = config.Workload("native", cfg, m, p)
        a, b = impl_a(w), impl_b(w)
        tb, sr = w.seeds()
        a.reset(tb, sr), b.reset(tb, sr)
        cmds, _ = config.bench_commands(2, agents, steps)
        for s in range(steps):
            a.step(cmds[s]), b.step(cmds[s])
        assert (a.digest() == b.digest()).all(), mode
        assert (a.results() == b.results()).all(), mode
        d = a.dump(0)
        assert sum(pp.active for pp in d.portals) >= 6  # the shipped world's six exits are live (gameplay.hpp:1265-1270)


def test_native_world_on_the_committed_maps_emulator():
    """The shipped world from the committed fixtures, Solo and Squad with the reference's level-4 account: device source
    on the wave emulator against the oracle (runs everywhere)."""
    _native_world(Oracle, Emu, 300)


@pytest.mark.gpu
def test_native_world_on_the_committed_maps_device():
    """The same on the GPU: the reference's real portal topology through the HIP kernels (3 floors, '^'/'v' links,
    Squad layout on floors 0/2; gameplay.hpp:1249-1274,1861-1903)."""
    from strikeforce_amd import env
    _native_world(Oracle, env.ArenaBatch, 400)
