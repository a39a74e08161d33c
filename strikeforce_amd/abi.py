"""ctypes mirror of include/strikeforce.h (the C-ABI boundary).

Only data layout lives here; the loader for the HIP library is in ``env.py``.  The same structures
are used by ``tests/`` to talk to the CPU oracle, which exports the same calls with an ``sfo_`` prefix.
"""
import ctypes as C

SF_ABI_VERSION = 2
OBS_CHANNELS = 32
OBS_WINDOW = 31
OBS_FLOATS = OBS_CHANNELS * OBS_WINDOW * OBS_WINDOW  # 30752, bots/bot-0.5/Custom.hpp:137-159

MAX_HUMANS = 64
MAX_ZOMBIES = 9000
MAX_BULLETS = 256
MAX_PORTALS = 9000
MAX_AGENTS = 16

MODE_SOLO, MODE_TIMER, MODE_SQUAD, MODE_BATTLE = 0, 1, 2, 3
RUNNING, DIED, WON, TIME_LOST, TIME_WON, QUIT = 0, 1, 2, 3, 4, 5

CELL_WALL, CELL_TEMP, CELL_PIN_UP, CELL_PIN_DN, CELL_POUT, CELL_CHEST = 1, 2, 4, 8, 16, 32
CELL_CONS_SHIFT = 6

# gameplay.hpp:45 valid_commands plus '_' (gameplay.hpp:696): the 30 command codes.
VALID_COMMANDS = "+qe3uzxawsdfghjkl;'cvbnm,./[]"
ALL_COMMANDS = VALID_COMMANDS + "_"
# bench/random-agent command set: the 30 codes minus the UI toggle '3' and suicide '_'.
BENCH_COMMANDS = "+qeuzxawsdfghjkl;'cvbnm,./[]"
assert len(ALL_COMMANDS) == 30 and len(BENCH_COMMANDS) == 28


class Profile(C.Structure):
    _fields_ = [
        ("def_hp", C.c_int32), ("mindamage_def", C.c_int32), ("def_stamina", C.c_int32),
        ("level_solo", C.c_int32), ("level_timer", C.c_int32), ("level_squad", C.c_int32),
        ("money", C.c_int32),
        ("rate_solo", C.c_int32), ("rate_timer", C.c_int32), ("rate_squad", C.c_int32), ("rate", C.c_int32),
        ("cons", C.c_int32 * 4),
        ("throw_lvl_cnt", (C.c_int32 * 2) * 4),
        ("weapon_lvl", C.c_int32 * 8),
        ("backpack_lvl", C.c_int32),
    ]

    @classmethod
    def from_tokens(cls, tokens):
        """Build from the 32 integers of a reference character file (name token removed)."""
        t = [int(x) for x in tokens]
        if len(t) != 32:
            raise ValueError("a character record has 32 integer tokens after the name, got %d" % len(t))
        p = cls()
        (p.def_hp, p.mindamage_def, p.def_stamina, p.level_solo, p.level_timer, p.level_squad, p.money,
         p.rate_solo, p.rate_timer, p.rate_squad, p.rate) = t[:11]
        for i in range(4):
            p.cons[i] = t[11 + i]
        for i in range(4):
            p.throw_lvl_cnt[i][0] = t[15 + 2 * i]
            p.throw_lvl_cnt[i][1] = t[16 + 2 * i]
        for i in range(8):
            p.weapon_lvl[i] = t[23 + i]
        p.backpack_lvl = t[31]
        return p


class Items(C.Structure):
    _fields_ = [
        ("cons", (C.c_int32 * 3) * 4),
        ("thr", (C.c_int32 * 4) * 4),
        ("weapon", (C.c_int32 * 4) * 8),
    ]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("arenas", C.c_int32),
        ("floors", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("cap_humans", C.c_int32), ("cap_zombies", C.c_int32), ("cap_bullets", C.c_int32),
        ("cap_portals", C.c_int32), ("cap_chests", C.c_int32),
        ("mode", C.c_int32),
        ("level", C.c_int32),
        ("n_agents", C.c_int32),
        ("ind", C.c_int32),
        ("agent_team", C.c_int32 * MAX_AGENTS),
        ("auto_reset", C.c_int32),
        ("reseed_stride", C.c_int32),
        ("timer_frames_per_level", C.c_int32),
        ("device", C.c_int32),
        ("map", C.c_char_p),
        ("map_portal", C.POINTER(C.c_int16)),
        ("player", Profile),
        ("npc", Profile),
        ("items", Items),
        ("n_agent_profiles", C.c_int32),
        ("agent_profile", Profile * MAX_AGENTS),
    ]


class ArenaHdr(C.Structure):
    _fields_ = [
        ("frame", C.c_int64), ("kills", C.c_int64), ("teams_kills", C.c_int64), ("loot", C.c_int64),
        ("chests", C.c_int64), ("jomle", C.c_int64), ("tb", C.c_int64), ("serial", C.c_int64),
        ("steps", C.c_int64), ("episodes", C.c_int64),
        ("rng", C.c_int32 * 18),
        ("done", C.c_int32), ("outcome", C.c_int32),
    ]


class HumanRec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("alive", "remote", "rnpc", "profile", "f", "r", "c", "way", "team", "hp", "stamina", "mindamage",
                 "kills", "damage", "effect", "vec", "ind")] + [
        ("cons", C.c_int32 * 4), ("throw_cnt", C.c_int32 * 4),
        ("blocks", C.c_int32), ("portals", C.c_int32), ("portal_ind", C.c_int32)]


class ZombieRec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("alive", "f", "r", "c", "hp", "mindamage", "super_")]


class BulletRec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("alive", "f", "r", "c", "way", "traveled", "damage", "effect", "range", "owner", "ref")]


class PortalRec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("active", "f", "r", "c")]


def struct_to_dict(s):
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        out[name] = list(v) if hasattr(v, "__len__") else v
    return out


def bind(lib, prefix):
    """Declare argtypes/restypes of the C-ABI on a loaded library (prefix 'sf_' or 'sfo_')."""
    g = lambda n: getattr(lib, prefix + n)
    vp = C.c_void_p
    g("reset").argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    g("step").argtypes = [vp, C.c_char_p]
    g("observe").argtypes = [vp, C.POINTER(C.c_float)]
    g("results").argtypes = [vp, C.POINTER(C.c_int32)]
    g("done").argtypes = [vp, C.POINTER(C.c_uint8)]
    g("state_digest").argtypes = [vp, C.POINTER(C.c_uint64)]
    g("dump_arena").argtypes = [vp, C.c_int32, C.POINTER(ArenaHdr), C.POINTER(HumanRec), C.POINTER(ZombieRec),
                                C.POINTER(BulletRec), C.POINTER(PortalRec), C.POINTER(C.c_uint8),
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    g("step_begin").argtypes = [vp]
    g("step_end").argtypes = [vp, C.c_char_p]
    g("agent_alive").argtypes = [vp, C.POINTER(C.c_uint8)]
    # (the oracle keeps its own per-arena entry, sfo_phase_draws, with 64-bit counts; a library of an earlier round,
    # loaded for an A/B through SF_LIBRARY_PATH, lacks the entry)
    if prefix != "sfo_" and hasattr(lib, prefix + "phase_draws"):
        g("phase_draws").argtypes = [vp, C.POINTER(C.c_int32)]
        g("phase_draws").restype = C.c_int
    for n in ("reset", "step", "observe", "results", "done", "state_digest", "dump_arena", "step_begin", "step_end",
              "agent_alive"):
        g(n).restype = C.c_int
    g("destroy").argtypes = [vp]
    return lib
