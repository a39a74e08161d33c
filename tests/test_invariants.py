"""Structural invariants of the world state that follow from the reference's rules, checked on full dumps of the
oracle, of the device core on the wave emulator and (gpu) of the HIP kernels at BASELINE size.  They are
size-independent: they hold for every arena at every step, whatever the seed."""
import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import ArenaDump, Oracle
from strikeforce_amd import abi, config


def check(d, cfg):
    flags = d.flags
    cols = cfg.cols
    cell = lambda e: (e.f * cfg.rows + e.r) * cols + e.c
    # `chest` counts the '?' cells: ++ in spawn_chest, -- in claim_chest (gameplay.hpp:507-515,532-543)
    assert d.hdr.chests == int(((flags & abi.CELL_CHEST) != 0).sum())
    # a cell holds at most one character, never on a wall: moves/spawns need '.', '?', '^', 'v', '*' (gameplay.hpp:750,535,547,562)
    occ = [cell(h) for h in d.humans if h.alive] + [cell(z) for z in d.zombies if z.alive]
    assert len(occ) == len(set(occ))
    assert all(not (flags[c] & abi.CELL_WALL) for c in occ)
    # one designated bullet per cell (the cell's single `bullet` pointer, gameplay.hpp:241)
    ref = [cell(b) for b in d.bullets if b.alive and b.ref]
    assert len(ref) == len(set(ref))
    for b in d.bullets:
        if b.alive:
            assert 1 <= b.way <= 4 and 0 <= b.traveled and b.traveled + 1 < max(b.range, 2) + 1
            assert 0 <= b.owner <= cfg.cap_humans
    # active exits sit on 'O' cells, and every 'O' cell is an active exit (gameplay.hpp:1265-1270,727-733,1355-1362)
    exits = sorted(cell(p) for p in d.portals if p.active)
    assert exits == sorted(np.nonzero(flags & abi.CELL_POUT)[0].tolist())
    # player-built objects: damage below the limit that removes them (gameplay.hpp:1357,1367), clean elsewhere
    temp = (flags & abi.CELL_TEMP) != 0
    assert (d.dmg[~temp] == 0).all()
    walls = temp & ((flags & abi.CELL_WALL) != 0)
    ups = temp & ((flags & abi.CELL_PIN_UP) != 0)
    assert (d.dmg[walls] < 1100).all() and (d.dmg[ups] < 1000).all()
    assert ((d.pidx >= 0) == ((flags & (abi.CELL_PIN_UP | abi.CELL_PIN_DN)) != 0)).all()
    for h in d.humans:
        if h.alive:
            assert 1 <= h.way <= 4 and h.hp > 0 and -1 <= h.vec <= 2 and h.blocks >= 0 and h.portals >= 0
            assert all(x >= 0 for x in h.cons) and all(x >= 0 for x in h.throw_cnt)
    # loot only comes from kills credited to the player's team: 50/75 per zombie + 450/675 if the player shot it,
    # 100 + 900 per rival human (gameplay.hpp:588-593,625-629)
    assert d.hdr.loot >= 0 and d.hdr.loot % 25 == 0 and d.hdr.kills <= d.hdr.teams_kills


@pytest.mark.parametrize("impl", [Oracle, Emu])
@pytest.mark.parametrize("name,steps", [("C3", 300), ("STRESS", 400), ("FLOORS", 300)])
def test_invariants_cpu(impl, name, steps):
    w = config.baseline_workload(name, arenas=3)
    sim = impl(w)
    tb, sr = w.seeds()
    sim.reset(tb, sr)
    cmds, _ = config.bench_commands(3, w.cfg.n_agents, steps)
    for s in range(steps):
        sim.step(cmds[s])
        if s % 7 == 0:
            for a in range(3):
                check(sim.dump(a), w.cfg)


@pytest.mark.gpu
def test_invariants_gpu_full_size():
    import torch
    from strikeforce_amd import env
    w = config.baseline_workload("C3", arenas=4096)
    g = env.ArenaBatch(w)
    tb, sr = w.seeds()
    g.reset(tb, sr)
    cmds, _ = config.bench_commands(4096, 1, 600)
    d = torch.from_numpy(cmds).cuda()
    for s in range(0, 600, 150):
        g.step_device(d.data_ptr() + s * 4096, 150)
        g.synchronize()
        for a in (0, 1, 777, 2048, 4095):
            check(ArenaDump(*g.dump_raw(a)), w.cfg)
