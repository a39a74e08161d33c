"""The N > 1 path on CPU: 2 and 8 gloo ranks (the driver's node has eight GPUs) each simulate their shard (device core
on the wave emulator), all-gather the result records, and the union must equal one oracle run over all arenas."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from strikeforce_amd import config, shard
from emu_lib import Emu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
A, steps = {arenas}, {steps}
w = config.baseline_workload("C2", arenas=A)
w.cfg.reseed_stride = world * A
sim = Emu(w)
tb, sr = shard.shard_seeds(w, rank)
sim.reset(tb, sr)
cmds, _ = config.bench_commands(A, 1, steps, seed0=shard.command_seed(w, rank))
sim.step_many(cmds)
local = torch.from_numpy(sim.results())
allr = shard.gather_results(local, world)
dig = torch.from_numpy(sim.digest().astype(np.int64))
alld = shard.gather_results(dig, world)
if rank == 0:
    np.save({out!r}, allr.numpy())
    np.save({out!r} + ".dig.npy", alld.numpy())
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,A,steps", [(2, 6, 900), (8, 3, 700)])
def test_sharded_ranks_match_a_single_oracle(tmp_path, world, A, steps):
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out, arenas=A, steps=steps))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    gathered = np.load(out)
    digests = np.load(out + ".dig.npy").astype(np.uint64)

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    from strikeforce_amd import config
    w = config.baseline_workload("C2", arenas=world * A)
    o = Oracle(w)
    tb, sr = w.seeds()
    o.reset(tb, sr)
    o.step_many(np.concatenate([config.bench_commands(A, 1, steps, seed0=12345 + r * A)[0] for r in range(world)], axis=1))
    # shards re-seed with the global arena count (reseed_stride), so sharded == unsharded exactly,
    # across episode boundaries too
    assert gathered.shape == (world * A, 1, 8)
    assert (gathered == o.results()).all()
    assert (digests == o.digest()).all()
    assert sum(o.dump(a).hdr.episodes for a in range(world * A)) > 0
