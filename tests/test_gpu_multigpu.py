"""The multi-GPU path on the one card a gpurun box has: the C-ABI side of the result all-gather with a one-rank RCCL
communicator, and `bench.py --gpus 2` starting its own two ranks (gloo rehearsal switch: RCCL refuses two ranks on one
device, so the 2-rank run exercises the launch / sharding / reduction control flow and the 1-rank test the RCCL calls).
The scaling curve itself is the driver's to measure on an 8-GPU node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_results_allgather_through_the_c_abi_one_rank():
    import torch
    from strikeforce_amd import config, env
    w = config.baseline_workload("C2", arenas=64, device=0)
    g = env.ArenaBatch(w)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.reset(*w.seeds())
    g.comm_init(env.ArenaBatch.comm_unique_id(), 0, 1)
    assert g.comm_ranks() == 1  # ncclCommCount of the library's communicator (sf_comm_ranks)
    cmds, _ = config.bench_commands(64, 1, 600)
    d = torch.from_numpy(cmds).cuda()
    out = [torch.full((64 * 8,), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
    want = []
    for i, s in enumerate(range(0, 600, 100)):
        g.step_device(d.data_ptr() + s * 64, 100)
        g.results_allgather(out[i & 1].data_ptr())
        want.append(g.results().reshape(-1).copy())  # (synchronises the launch stream; the gather runs on its own)
        g.comm_wait(host_too=True)
        assert (out[i & 1].cpu().numpy() == want[-1]).all()
    assert any((x != 0).any() for x in want)  # some episode ended and latched a record
    g.close()


def test_bench_starts_its_own_ranks():
    env_ = dict(os.environ, SF_BENCH_BACKEND="gloo", SF_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env_.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--arenas", "512", "--preroll", "100", "--no-interactive", "--no-cpu-baseline"],
                       env=env_, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert "configs[2]" in d["config"]["workload"]
    assert "rccl_ranks" in d and d["rccl_ranks"] is None  # (gloo rehearsal: no RCCL communicator; a real run reports N)
    assert d["repeats"]["n"] == 21                         # a 20-step region is short: median of 21 repeats
