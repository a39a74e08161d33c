"""`.sf_sample` trajectories — the reference's own record/replay format.

Written by gameplay::load_data / human_action when logging is on (gameplay.hpp:1784-1794,1910-1914,966-967) and read
back in replay mode (gameplay.hpp:1771-1782,968-969); the player blob is Human::log_file / scan_file
(Character.hpp:619-648,570-617).  Offline Solo / Timer files look like

    <tb> <serial>
    1 <ind> <team>
    <name>                       \
    <def_Hp> ... 31 more integers, one per line   (the character record)
    <one command char per loop iteration, one per line>

This module reads and writes that format and replays a sample through any backend with the
reset/step surface (ArenaBatch on the GPU; the oracle and the emulator in tests).
"""
import ctypes as C

import numpy as np

from . import abi, config


class Sample:
    def __init__(self, tb, serial, profile_tokens, commands, name="player", ind=0, team=1):
        if len(profile_tokens) != 32:
            raise ValueError("a character record has 32 integers after the name")
        self.tb, self.serial = int(tb), int(serial)
        self.profile_tokens = [int(x) for x in profile_tokens]
        self.commands = str(commands)
        self.name, self.ind, self.team = name, int(ind), int(team)


def write_sample(path, sample):
    """Same byte layout as the reference's logger: header, Human::log_file blob, then `command << '\\n'` per iteration."""
    with open(path, "w") as f:
        f.write("%d %d\n" % (sample.tb, sample.serial))
        f.write("1 %d %d\n" % (sample.ind, sample.team))
        f.write(sample.name + "\n")
        for t in sample.profile_tokens:
            f.write("%d\n" % t)
        for c in sample.commands:
            f.write(c + "\n")


def read_sample(path):
    """Parses like the reference does: whitespace-separated tokens (operator>>), one non-blank char per command."""
    tok = open(path).read().split()
    tb, serial, players, ind, team = int(tok[0]), int(tok[1]), int(tok[2]), int(tok[3]), int(tok[4])
    if players != 1:
        raise ValueError("only offline (one player) samples are supported, got players=%d" % players)
    name = tok[5]
    prof = [int(x) for x in tok[6:38]]
    cmds = tok[38:]
    if any(len(c) != 1 for c in cmds):
        # `replay_file >> command[ind]` reads one char at a time: a token like "ab" is two commands
        cmds = [ch for c in cmds for ch in c]
    return Sample(tb, serial, prof, "".join(cmds), name=name, ind=ind, team=team)


def workload_for(sample, rows, cols, map_bytes, portal=None, floors=1, mode=abi.MODE_SOLO, level=1, H=64, Z=64, B=256,
                 P=16, chests=9000, device=0):
    """A one-arena workload that replays `sample` on the given map with the sample's character record."""
    cfg = config.make_config(1, rows, cols, floors=floors, H=H, Z=Z, B=B, P=P, chests=chests, mode=mode, level=level,
                             auto_reset=0, player_tokens=sample.profile_tokens, device=device)
    return config.Workload("replay", cfg, map_bytes, portal or [-1] * (floors * rows * cols))


def replay(sample, sim):
    """Feeds the sample's command stream to `sim` (already constructed on workload_for(sample, ...)); stops when the
    episode ends, like the reference's loop.  Returns the number of iterations played."""
    tb = (C.c_uint64 * 1)(sample.tb)
    sr = (C.c_uint64 * 1)(sample.serial)
    sim.reset(tb, sr)
    n = 0
    for ch in sample.commands:
        if sim.done()[0]:
            break
        sim.step(np.array([ord(ch)], dtype=np.uint8))
        n += 1
    return n
