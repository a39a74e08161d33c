"""Times the observation kernel alone: 4096 arenas of configs[1] after 400 steps, 50 observes (torch events);
plain and delta (one persistent buffer, the simulation advancing one step between calls)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strikeforce_amd import config, env
A = 4096
w = config.baseline_workload("C2", arenas=A)
g = env.ArenaBatch(w)
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.reset(*w.seeds())
cmds, _ = config.bench_commands(A, 1, 500)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 400)
obs = torch.empty(A * 30752, dtype=torch.float32, device="cuda")
for _ in range(5):
    g.observe_device(obs.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    g.observe_device(obs.data_ptr())
e1.record()
torch.cuda.synchronize()
print(os.environ.get("SF_LIBRARY_PATH", "default"), "k_observe %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
if hasattr(g, "observe_device_delta"):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    g.observe_device_delta(obs.data_ptr())
    ev[0].record()
    for s in range(50):
        g.step_device(d.data_ptr() + (400 + s) * A, 1)
        g.observe_device_delta(obs.data_ptr())
    ev[1].record()
    for s in range(50):
        g.step_device(d.data_ptr() + (450 + s) * A, 1)
    ev[2].record()
    torch.cuda.synchronize()
    print("step + delta observe %.1f us, step alone %.1f us -> delta observe %.1f us"
          % (ev[0].elapsed_time(ev[1]) / 50 * 1e3, ev[1].elapsed_time(ev[2]) / 50 * 1e3,
             (ev[0].elapsed_time(ev[1]) - ev[1].elapsed_time(ev[2])) / 50 * 1e3))
