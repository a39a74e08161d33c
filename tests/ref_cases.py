"""The runs on which the reference's own tick path (oracle/_ref/sf_ref_tick, oracle/ref_tick.py) is compared with this
repo: shared by tests/test_ref_tick.py (live, whole-state comparison with the oracle), tests/golden/make_ref_traj.py
(writes the reference's per-step digests and observations to tests/golden/ref_traj.json) and tests/test_ref_traj.py
(oracle, emulated device core and, -m gpu, the device reproduce that file where no reference exists)."""
import os

import fuzz_cases
from strikeforce_amd import abi, config

FIXTURE_MAP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps")
RICH = fuzz_cases.ACCOUNT_1  # the reference's level-10 account record: 15000 Hp, survives long runs


# Slot pools for the native world.  "game": what a whole game of the reference needs (its own pools hold 9000; measured
# with tests/tools/population_curve.py on the shipped maps, level-10 account: a level-3 Timer game ends with 237 zombies and
# 109 exits, a level-10 one has 379 / 166 when the player dies at step ~18 000; humans peak at 22, bullets at 25) — the
# device keeps such pools in LDS (sf_core.hpp ZL).  "lanes": the one-lane-per-slot pools (64 zombies, 32 exits), enough for
# about 1 600 steps: the register form of the kernels on the native world.
POOLS = {"game": dict(H=64, Z=1024, B=256, P=512), "lanes": dict(H=64, Z=64, B=256, P=32)}


def native(mode, level, player, maps="synthetic", map_seed=11, wall_p=0.06, pools="game"):
    """The reference's native world, gameplay.hpp:37: 3 floors x 30 x 100."""
    if maps == "shipped":
        m, p = config.load_reference_maps(FIXTURE_MAP_DIR)
    else:
        m, p = config.three_floor_map(30, 100, wall_p=wall_p, map_seed=map_seed)
    cfg = config.make_config(1, 30, 100, floors=3, mode=mode, level=level, n_agents=1,
                             player_tokens=player, auto_reset=0, timer_frames=1 << 20, **POOLS[pools])
    return config.Workload("native", cfg, m, p)


def baseline(which, player):
    """A BASELINE.json configuration (or STRESS / FLOORS), one arena, no auto-reset, the given player record."""
    w = config.baseline_workload(which, arenas=1, auto_reset=0)
    w.cfg.timer_frames_per_level = 1 << 20
    w.cfg.player = abi.Profile.from_tokens(player)
    w.cfg.n_agents = 1
    # the reference pools exits by B (`portal[B]`, gameplay.hpp:51-53): the same size here, so that a run's exits — the
    # map's own plus one per human that places its portal — never outgrow ours before the reference's
    w.cfg.cap_portals = w.cfg.cap_bullets
    return w


def plane(which, mode, player, level=1):
    """The dimensions and pools of BASELINE.json's configs[3] (128 x 128, H10 Z20 B64) or configs[4] (256 x 256, H8 Z56 B128)
    as a one-player Solo / Timer game: the reference compiled for them (oracle/ref_tick.py TEST_DIMS) plays what the
    device's HBM-plane kernels play — k_step<., true, true> at 128 x 128, k_step<., true, false> at 256 x 256.
    (Squad and Battle themselves cannot run on a one-floor reference: load_data places Squad opponents on floor 2,
    gameplay.hpp:1887-1897, and Battle needs the match server.)"""
    n, H, Z, B = {"C4": (128, 10, 20, 64), "C5": (256, 8, 56, 128)}[which]
    cfg = config.make_config(1, n, n, H=H, Z=Z, B=B, P=B, mode=mode, level=level, n_agents=1, player_tokens=player,
                             auto_reset=0, timer_frames=1 << 20)
    m, p = config.synthetic_map(n, n, portal_pairs=2)
    return config.Workload("%s-dims" % which, cfg, m, p)


def squad_win_commands():
    """The scripted Squad game that ends won (tests/tools/plan_squad_win.py wrote it): its header line and command string."""
    with open(os.path.join(os.path.dirname(FIXTURE_MAP_DIR), "squad_win_commands.txt")) as f:
        head, cmds = f.read().strip().split("\n")
    return head, cmds


# name -> (workload builder, player record, tb, serial, steps, command seed — or the command string itself —, native_caps)
# native_caps True: the reference as it stands (3 x 30 x 100, pools of 9000); False: a patched-dimensions build
# (gameplay.hpp:37 replaced) whose pools are exactly the configuration's
GOLDEN_CASES = {
    "native-solo-armed": (lambda: native(abi.MODE_SOLO, 2, config.HUMAN_ENEMY_TOKENS, pools="lanes"), config.HUMAN_ENEMY_TOKENS,
                          1700000000, 123456789, 400, 12345, True),
    "native-timer-level10": (lambda: native(abi.MODE_TIMER, 4, RICH, map_seed=5, wall_p=0.03), RICH,
                             1771155561, 1073741823, 1200, 99, True),
    "shipped-solo-fresh": (lambda: native(abi.MODE_SOLO, 1, config.HUMAN_TOKENS, maps="shipped", pools="lanes"), config.HUMAN_TOKENS,
                           1700000123, 987654321, 600, 7, True),
    # whole Timer games on the reference's own maps, to the end of the frame clock (level x 7500 frames = level x 3750
    # steps; the reference's own clock is time(0), gameplay.hpp:1145): 95 and 237 live zombies, 49 and 109 exits at the end
    "shipped-timer-level1-full": (lambda: native(abi.MODE_TIMER, 1, RICH, maps="shipped"), RICH, 1700000999, 55555, 3750, 31, True),
    "shipped-timer-level3-full": (lambda: native(abi.MODE_TIMER, 3, RICH, maps="shipped"), RICH, 1700000999, 55555, 11250, 31, True),
    "shipped-squad-level3": (lambda: native(abi.MODE_SQUAD, 3, RICH, maps="shipped"), RICH,
                             1700004245, 424242, 600, 8, True),
    # a Squad game played to its end, WON: ten team kills and every rival dead (gameplay.hpp:1204-1229), scripted
    "shipped-squad-won": (lambda: native(abi.MODE_SQUAD, 1, RICH, maps="shipped"), RICH, 1700007777, 424242,
                          len(squad_win_commands()[1]), squad_win_commands()[1], True),
    "C1": (lambda: baseline("C1", config.HUMAN_TOKENS), config.HUMAN_TOKENS, 1700000000, 123456789, 1000, 12345, False),
    "C2": (lambda: baseline("C2", config.HUMAN_ENEMY_TOKENS), config.HUMAN_ENEMY_TOKENS, 1700000002, 123456789, 1000, 12347, False),
    "C3": (lambda: baseline("C3", config.HUMAN_ENEMY_TOKENS), config.HUMAN_ENEMY_TOKENS, 1700000002, 123456789, 1000, 12347, False),
    "STRESS": (lambda: baseline("STRESS", RICH), RICH, 1700000001, 123456789, 800, 12346, False),
    "C4dims-solo": (lambda: plane("C4", abi.MODE_SOLO, RICH, level=4), RICH, 1700000004, 123456789, 700, 12349, False),
    "C4dims-timer": (lambda: plane("C4", abi.MODE_TIMER, config.HUMAN_ENEMY_TOKENS), config.HUMAN_ENEMY_TOKENS, 1700000005, 123456789, 700, 12350, False),
    "C5dims-solo": (lambda: plane("C5", abi.MODE_SOLO, RICH, level=4), RICH, 1700000006, 123456789, 700, 12351, False),
    "C5dims-timer": (lambda: plane("C5", abi.MODE_TIMER, RICH, level=2), RICH, 1700000007, 123456789, 700, 12352, False),
    "FLOORS-squad": (lambda: baseline("FLOORS", config.HUMAN_ENEMY_TOKENS), config.HUMAN_ENEMY_TOKENS,
                     1700000321, 123456789, 600, 77, False),
}
OBS_EVERY = 200
