"""Slot pools of more than 64 zombies / exits (the device keeps them in LDS, sf_core.hpp `ZL`) on every kernel variant:
flag plane in LDS (64 x 64), in HBM with the cell bitmaps in LDS (120 x 120), in HBM without bitmaps (160 x 160).
The worlds start with 90 exits of their own (two 64-slot words of the exit table from the first step on) and are
played long enough for the zombie table to reach its third word; games end and restart on the way (auto_reset: the
words an ended game had in use are cleared).  Emulated device core against the oracle here, the HIP kernels with
`-m gpu`; the whole native games of the REFERENCE on these kernels are tests/test_ref_traj.py's."""
import numpy as np
import pytest

import ref_cases
from emu_lib import Emu
from oracle_lib import Oracle, diff_dumps
from strikeforce_amd import abi, config


def world(n, arenas, exits=90):
    m, p = config.synthetic_map(n, n, wall_p=0.05, map_seed=77 + n, portal_pairs=2)
    chars = bytearray(m)
    r = np.random.RandomState(n)
    placed = 0
    while placed < exits:  # extra exits of the map's own ('O' cells: exit k = the k-th in scan order, gameplay.hpp:1265-1270)
        i, j = int(r.randint(2, n - 2)), int(r.randint(2, n - 2))
        if chars[i * n + j] == ord("."):
            chars[i * n + j] = ord("O")
            placed += 1
    # a player that lives long and kills little (30000 Hp, its punch does 1), one NPC human: the herd grows
    tank = [30000, 1, 1000, 1, 1, 1, 1000, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0] + [0] * 8 + [1]
    cfg = config.make_config(arenas, n, n, H=2, Z=320, B=128, P=200, mode=abi.MODE_TIMER, level=4, n_agents=1,
                             player_tokens=tank, auto_reset=1, timer_frames=3300)  # (level 4: a game lasts 6600 steps unless the player dies)
    return config.Workload("large-pools-%d" % n, cfg, bytes(chars), p)


def run(impl, n, arenas, steps, k):
    w = world(n, arenas)
    o, g = Oracle(w), impl(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), g.reset(tb, sr)
    cmds, _ = config.bench_commands(arenas, 1, steps, seed0=4321)
    most = 0
    for s0 in range(0, steps, k):
        o.step_many(cmds[s0:s0 + k])
        g.step_many(cmds[s0:s0 + k])
        assert (o.digest() == g.digest()).all(), "%d x %d: digests differ after %d steps" % (n, n, s0 + k)
        most = max(most, max(sum(z.alive for z in o.dump(a).zombies) for a in range(arenas)))
    for a in range(arenas):
        d = diff_dumps(o.dump(a).as_dict(), g.dump(a).as_dict())
        assert d is None, d
    assert (o.results() == g.results()).all()
    d0 = o.dump(0)
    assert most > 128, "the zombie table never reached its third word (%d)" % most
    assert sum(p.active for p in d0.portals) > 64
    assert sum(o.dump(a).hdr.episodes for a in range(arenas)) > 0, "no game ended and restarted"
    return o, g


@pytest.mark.parametrize("n", [64, 120, 160])
def test_large_pools_on_the_emulated_core(n):
    run(Emu, n, 2, 5000, 100)


class _Device:
    def __init__(self, w):
        from strikeforce_amd import env
        self.g = env.ArenaBatch(w)
        self.n = w.cfg.arenas * w.cfg.n_agents

    def reset(self, tb, sr):
        self.g.reset(tb, sr)

    def step_many(self, cmds):
        import torch
        d = torch.from_numpy(np.ascontiguousarray(cmds)).cuda()
        self.g.step_device(d.data_ptr(), cmds.shape[0])
        self.g.synchronize()

    def digest(self):
        return self.g.digest()

    def results(self):
        return self.g.results()

    def dump(self, a):
        from oracle_lib import ArenaDump
        return ArenaDump(*self.g.dump_raw(a))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 120, 160])
def test_large_pools_on_the_device(n):
    """k_step<4, false, true, true>, <4, true, true, true>, <4, true, false, true>: 48 arenas, launches of 100 steps."""
    o, g = run(_Device, n, 48, 5000, 100)
    # and one step per launch from there, the whole state after every step
    cmds, _ = config.bench_commands(48, 1, 40, seed0=99)
    for s in range(40):
        o.step(cmds[s]), g.g.step(cmds[s])
        for a in (0, 17, 47):
            d = diff_dumps(o.dump(a).as_dict(), g.dump(a).as_dict())
            assert d is None, "%d x %d step %d arena %d: %s" % (n, n, s, a, d)
