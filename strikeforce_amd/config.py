"""Workload configuration: shipped stat tables, the synthetic map generator and the BASELINE configs.

Reference data sources (values only; /root/reference/StrikeForce-client):
  Items/cons0-3.txt, Items/throw0-3.txt, Items/w0-7.txt  (Item.hpp:69-74,113-118,149-154)
  character/human.txt, character/human_enemy.txt          (Character.hpp:650-709)
Synthetic inputs follow SURVEY.md §8(d).
"""
import ctypes as C

from . import abi

# name price vol lvl | stamina Hp effect
_CONS = [(20, 0, 20), (0, 200, 10), (20, 50, 10), (20, 400, 20)]
# stamina damage effect range
_THROW = [(-15, 50, -20, 100), (-20, 75, -200, 100), (-30, 100, -80, 100), (-35, 125, -80, 100)]
_WEAPON = [(-25, 150, -50, 1), (-40, 175, -60, 1), (-40, 200, -70, 1), (-45, 225, -80, 1),
           (-50, 150, -55, 100), (-50, 175, -65, 100), (-50, 200, -75, 100), (-50, 225, -85, 100)]

# character/human.txt and character/human_enemy.txt, name token removed
HUMAN_TOKENS = [1000, 100, 1000, 1, 1, 1, 1000, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0,
                0, 0, 0, 0, 0, 0, 0, 0, 1]
HUMAN_ENEMY_TOKENS = [1000, 100, 1000000, 1, 1, 1, 1000] + [1] * 25


def default_items():
    it = abi.Items()
    for i, row in enumerate(_CONS):
        for j, v in enumerate(row):
            it.cons[i][j] = v
    for i, row in enumerate(_THROW):
        for j, v in enumerate(row):
            it.thr[i][j] = v
    for i, row in enumerate(_WEAPON):
        for j, v in enumerate(row):
            it.weapon[i][j] = v
    return it


def _lcg(seed):
    x = seed & 0xFFFFFFFF
    while True:
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        yield x >> 16


def synthetic_map(rows, cols, floors=1, wall_p=0.08, map_seed=2024, portal_pairs=0, keep_clear=()):
    """Border '#', interior '#' with probability wall_p, (1,1) forced '.'; optional static portal
    pairs ('^ k' entrances in the top border, 'O' exits in the interior, like map/floor1.txt).
    Returns (chars: bytes, portal: list[int]) in floor-major, row-major order."""
    g = _lcg(map_seed)
    thr = int(wall_p * 65536)
    chars = []
    portal = []
    for f in range(floors):
        for r in range(rows):
            for c in range(cols):
                border = r == 0 or c == 0 or r == rows - 1 or c == cols - 1
                ch = "#" if border or next(g) < thr else "."
                chars.append(ch)
                portal.append(-1)
    idx = lambda f, r, c: (f * rows + r) * cols + c
    for f in range(floors):
        chars[idx(f, 1, 1)] = "."
    for (f, r, c) in keep_clear:
        chars[idx(f, r, c)] = "."
    # exits are numbered in scan order by the parser (gameplay.hpp:1265-1270): place exits so that
    # pair k's exit is the k-th 'O' met in floor-major, row-major order.
    for k in range(portal_pairs):
        er, ec = 2 + (rows - 5) * (k + 1) // (portal_pairs + 1), cols // 2 + 3 * k
        chars[idx(0, er, ec)] = "O"
        for dc in (-1, 0, 1):  # three adjacent entrances per pair, as in map/floor1.txt
            cc = 3 + k * (cols // (portal_pairs + 1)) + dc + 1
            chars[idx(0, 0, cc)] = "^"
            portal[idx(0, 0, cc)] = k
            chars[idx(0, 1, cc)] = "."  # the cell below an entrance must be walkable
    return "".join(chars).encode("ascii"), portal


def three_floor_map(rows, cols, wall_p=0.05, map_seed=7):
    """Three floors; floor f has one exit 'O' (exit number f in scan order), '^' entrances in its top border that
    lead to floor (f+1) % 3 and 'v' entrances that lead to floor (f-1) % 3 (cf. map/floor1.txt .. floor3.txt)."""
    g = _lcg(map_seed)
    thr = int(wall_p * 65536)
    chars, portal = [], []
    for f in range(3):
        for r in range(rows):
            for c in range(cols):
                border = r == 0 or c == 0 or r == rows - 1 or c == cols - 1
                chars.append("#" if border or next(g) < thr else ".")
                portal.append(-1)
    idx = lambda f, r, c: (f * rows + r) * cols + c
    for f in range(3):
        for (r, c) in [(1, 1), (3, 1)] + [(1, i + 1) for i in range(1, 10)]:
            chars[idx(f, r, c)] = "."
        chars[idx(f, rows // 2, cols // 2 + f)] = "O"
        for j, (sym, dest) in enumerate((("^", (f + 1) % 3), ("v", (f - 1) % 3))):
            for dc in range(2):
                cc = 12 + 5 * j + dc
                chars[idx(f, 0, cc)] = sym
                portal[idx(f, 0, cc)] = dest
                chars[idx(f, 1, cc)] = "."
    return "".join(chars).encode("ascii"), portal


def parse_floor_text(text, rows, cols):
    """One `map/floorK.txt` of the reference, read the way gameplay::setup() does (gameplay.hpp:1249-1274): characters
    are taken with `f >> c` (whitespace skipped), and a '^' or 'v' is followed by the number of the exit it leads to.
    Returns (chars: str of rows*cols, portal: list[int])."""
    chars, portal = [], []
    i, n = 0, len(text)
    while len(chars) < rows * cols:
        while i < n and text[i].isspace():
            i += 1
        if i >= n:
            raise ValueError("map text ends after %d of %d cells" % (len(chars), rows * cols))
        c = text[i]
        i += 1
        idx = -1
        if c in "^v":
            while i < n and text[i].isspace():
                i += 1
            j = i
            while j < n and (text[j].isdigit() or (j == i and text[j] == "-")):
                j += 1
            if j == i:
                raise ValueError("portal entrance without an exit number at cell %d" % len(chars))
            idx = int(text[i:j])
            i = j
        elif c not in "#.O":
            c = "."  # setup() leaves any other character as an empty cell
        chars.append(c)
        portal.append(idx)
    return "".join(chars), portal


def format_floor_text(chars, portal, rows, cols):
    """The inverse of parse_floor_text: the reference's text layout (one row per line, '^ k' / 'v k' entrances)."""
    lines = []
    for r in range(rows):
        row = []
        for c in range(cols):
            ch = chars[r * cols + c]
            row.append("%s %d " % (ch, portal[r * cols + c]) if ch in "^v" else ch)
        lines.append("".join(row))
    return "\n".join(lines) + "\n"


def load_reference_maps(directory, floors=3, rows=30, cols=100):
    """Reads map/floor1.txt .. floor<floors>.txt of a reference checkout (native dims gameplay.hpp:37)."""
    import os
    chars, portal = "", []
    for k in range(floors):
        c, p = parse_floor_text(open(os.path.join(directory, "floor%d.txt" % (k + 1))).read(), rows, cols)
        chars += c
        portal += p
    return chars.encode("ascii"), portal


class Workload:
    """Owns the ctypes Config and the buffers it points into."""

    def __init__(self, name, cfg, map_bytes, portal):
        self.name = name
        self.cfg = cfg
        self._map = C.create_string_buffer(map_bytes, len(map_bytes))
        self._portal = (C.c_int16 * len(portal))(*portal)
        cfg.map = C.cast(self._map, C.c_char_p)
        cfg.map_portal = C.cast(self._portal, C.POINTER(C.c_int16))

    @property
    def cells(self):
        return self.cfg.floors * self.cfg.rows * self.cfg.cols

    def seeds(self, base_tb=1_700_000_000, serial=123_456_789, first_arena=0):
        """tb = 1 700 000 000 + arena_id, serial = 123 456 789."""
        n = self.cfg.arenas
        tb = (C.c_uint64 * n)(*[base_tb + first_arena + i for i in range(n)])
        sr = (C.c_uint64 * n)(*[serial] * n)
        return tb, sr


def make_config(arenas, rows, cols, floors=1, H=1, Z=16, B=32, P=8, chests=9000, mode=abi.MODE_SOLO, level=1,
                n_agents=1, teams=None, auto_reset=1, player_tokens=None, npc_tokens=None, device=0,
                timer_frames=0, reseed_stride=0, ind=0, agent_tokens=None):
    cfg = abi.Config()
    cfg.abi_version = abi.SF_ABI_VERSION
    cfg.arenas = arenas
    cfg.floors, cfg.rows, cfg.cols = floors, rows, cols
    cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.cap_portals, cfg.cap_chests = H, Z, B, P, chests
    cfg.mode, cfg.level, cfg.n_agents = mode, level, n_agents
    cfg.ind = ind
    for i in range(abi.MAX_AGENTS):
        cfg.agent_team[i] = (teams[i] if teams and i < len(teams) else i + 1)
    cfg.auto_reset = auto_reset
    cfg.reseed_stride = reseed_stride
    cfg.timer_frames_per_level = timer_frames
    cfg.device = device
    cfg.player = abi.Profile.from_tokens(player_tokens or HUMAN_ENEMY_TOKENS)
    cfg.npc = abi.Profile.from_tokens(npc_tokens or HUMAN_ENEMY_TOKENS)
    cfg.items = default_items()
    cfg.n_agent_profiles = 0
    if agent_tokens is not None:  # one character record per commanded human (a lock-step match's account blobs)
        if len(agent_tokens) != n_agents:
            raise ValueError("agent_tokens must hold one record per commanded human")
        cfg.n_agent_profiles = n_agents
        for i, t in enumerate(agent_tokens):
            cfg.agent_profile[i] = abi.Profile.from_tokens(t)
    return cfg


def baseline_workload(which, arenas=None, device=0, auto_reset=1):
    """The five configurations of BASELINE.json (`configs[0..4]`), caps as in SURVEY §8d."""
    if which == "C1":  # 32x32, 1 player + 4 zombies, Solo, character/human.txt (punch only)
        cfg = make_config(arenas or 1, 32, 32, H=1, Z=4, B=16, P=4, player_tokens=HUMAN_TOKENS, device=device,
                          auto_reset=auto_reset)
        m, p = synthetic_map(32, 32)
    elif which == "C2":  # 64x64, 1 player + 16 zombies, Solo, random-action agent
        cfg = make_config(arenas or 4096, 64, 64, H=1, Z=16, B=32, P=4, device=device, auto_reset=auto_reset)
        m, p = synthetic_map(64, 64)
    elif which == "C3":  # 64x64, 32 entities, Timer mode, bullets/throwables active
        cfg = make_config(arenas or 4096, 64, 64, H=8, Z=24, B=64, P=8, mode=abi.MODE_TIMER, device=device,
                          auto_reset=auto_reset)
        m, p = synthetic_map(64, 64)
    elif which == "C4":  # 128x128, Squad 5v5 + 20 zombies + blocks/portals
        cfg = make_config(arenas or 4096, 128, 128, H=10, Z=20, B=64, P=16, mode=abi.MODE_SQUAD, device=device,
                          auto_reset=auto_reset)
        keep = [(0, 3, 1)] + [(0, 1, i + 1) for i in range(1, 10)]
        m, p = synthetic_map(128, 128, portal_pairs=2, keep_clear=keep)
    elif which == "C5":  # 256x256, Battle Royale 8 agents + 56 zombies, full item set
        cfg = make_config(arenas or 4096, 256, 256, H=8, Z=56, B=128, P=16, mode=abi.MODE_BATTLE, n_agents=8,
                          teams=list(range(1, 9)), device=device, auto_reset=auto_reset)
        m, p = synthetic_map(256, 256, portal_pairs=2)
    elif which == "KITS":  # Battle match whose players bring different character records (ABI 2: sf_config.agent_profile):
        # the fresh character/human.txt, the armed human_enemy.txt, a levelled-up account, and a player with level-3 guns
        rich = [15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34] + [1] * 16 + [56]
        guns = list(HUMAN_ENEMY_TOKENS)
        guns[23:31] = [3, 0, 2, 0, 3, 1, 0, 2]
        recs = [HUMAN_TOKENS, HUMAN_ENEMY_TOKENS, rich, guns]
        cfg = make_config(arenas or 2, 40, 40, H=12, Z=12, B=64, P=8, mode=abi.MODE_BATTLE, n_agents=6,
                          teams=[1, 2, 3, 1, 2, 3], device=device, auto_reset=auto_reset,
                          agent_tokens=[recs[i % 4] for i in range(6)])
        m, p = synthetic_map(40, 40, wall_p=0.05, portal_pairs=1)
    elif which == "MAXCAP":  # every slot pool at the device kernels' maximum (64/64/256/64), 16 commanded humans
        cfg = make_config(arenas or 2, 48, 48, H=64, Z=64, B=256, P=64, mode=abi.MODE_BATTLE, n_agents=16,
                          teams=[1 + (i % 3) for i in range(16)], device=device, auto_reset=auto_reset)
        m, p = synthetic_map(48, 48, wall_p=0.04)
    elif which == "FLOORS":  # three floors linked by '^' / 'v' entrances like map/floor1-3.txt, Squad start layout
        cfg = make_config(arenas or 2, 20, 30, floors=3, H=12, Z=10, B=32, P=8, mode=abi.MODE_SQUAD, n_agents=3,
                          device=device, auto_reset=auto_reset)
        m, p = three_floor_map(20, 30)
    elif which == "NATIVE":  # the reference's own dimensions F=3, N=30, M=100 (gameplay.hpp:37) on a synthetic 3-floor map
        cfg = make_config(arenas or 2, 30, 100, floors=3, H=64, Z=64, B=256, P=32, mode=abi.MODE_SOLO, level=2,
                          device=device, auto_reset=auto_reset)
        m, p = three_floor_map(30, 100, wall_p=0.06, map_seed=11)
    elif which == "NATIVEGAME":  # a whole game on the reference's own dimensions: Timer level 3 (11 250 steps), pools of 1024
        # zombies and 512 exits — more than 64 slots: the device keeps those tables in LDS (sf_core.hpp ZL)
        cfg = make_config(arenas or 2, 30, 100, floors=3, H=64, Z=1024, B=256, P=512, mode=abi.MODE_TIMER, level=3,
                          device=device, auto_reset=auto_reset)
        m, p = three_floor_map(30, 100, wall_p=0.06, map_seed=11)
    elif which == "STRESS":  # not a BASELINE config: tiny slot pools so that every allocator runs dry
        cfg = make_config(arenas or 8, 24, 40, H=6, Z=12, B=5, P=3, chests=6, mode=abi.MODE_TIMER, device=device,
                          auto_reset=auto_reset, timer_frames=600)
        m, p = synthetic_map(24, 40, wall_p=0.05, portal_pairs=1)
    else:
        raise ValueError(which)
    return Workload(which, cfg, m, p)


def bench_commands(arenas, n_agents, steps, seed0=12345):
    """Random-action agent of SURVEY §8d: per (arena, agent) LCG x <- 1664525 x + 1013904223,
    command = BENCH_COMMANDS[(x >> 16) % 28], seed 12345 + arena*n_agents + agent.
    Returns a numpy uint8 array [steps][arenas][n_agents] and the final LCG states."""
    import numpy as np
    x = (np.arange(arenas * n_agents, dtype=np.uint64) + seed0).astype(np.uint32)
    table = np.frombuffer(abi.BENCH_COMMANDS.encode("ascii"), dtype=np.uint8)
    out = np.empty((steps, arenas * n_agents), dtype=np.uint8)
    for s in range(steps):
        x = (x * np.uint32(1664525) + np.uint32(1013904223)).astype(np.uint32)
        out[s] = table[(x >> np.uint32(16)) % 28]
    return out.reshape(steps, arenas, n_agents), x
