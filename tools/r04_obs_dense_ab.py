"""sf_observe_device alone and behind a one-step launch, configs[2], 4096 arenas, 1500 steps in:
python tools/r04_obs_dense_ab.py <library>   (SF_OBS_DENSE_BLOCK=1 in the environment: round 3's 256-thread kernel for every agent)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_LIBRARY_PATH"] = os.path.join(ROOT, sys.argv[1])
import torch  # noqa: E402

from strikeforce_amd import config, env  # noqa: E402

A = 4096
torch.cuda.set_stream(torch.cuda.Stream())
w = config.baseline_workload("C3", arenas=A)
g = env.ArenaBatch(w)
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.reset(*w.seeds())
cmds, _ = config.bench_commands(A, 1, 1500 + 300)
d = torch.from_numpy(cmds).cuda()
for s0 in range(0, 1500, 100):
    g.step_device(d.data_ptr() + s0 * A, 100)
obs = torch.empty((A, 30752), dtype=torch.float32, device="cuda")
g.observe_device(obs.data_ptr())
torch.cuda.synchronize()
s, out = 1500, []
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(40):
        g.observe_device(obs.data_ptr())
    ev[1].record()
    for _ in range(40):
        g.step_device(d.data_ptr() + s * A, 1)
        g.observe_device(obs.data_ptr())
        s += 1
    ev[2].record()
    torch.cuda.synchronize()
    out.append((ev[0].elapsed_time(ev[1]) / 40 * 1e3, ev[1].elapsed_time(ev[2]) / 40 * 1e3))
print("%s%s: observe %s us; step + observe %s us" % (sys.argv[1], " (block form)" if os.environ.get("SF_OBS_DENSE_BLOCK") else "",
                                                   " ".join("%.1f" % a for a, _ in out), " ".join("%.1f" % b for _, b in out)), flush=True)
