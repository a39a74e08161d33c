"""Run the observation kernel alone (for rocprofv3): 4096 arenas of configs[1] after 400 steps, 20 observes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strikeforce_amd import config, env
A = 4096
w = config.baseline_workload("C2", arenas=A)
g = env.ArenaBatch(w)
g.set_stream(torch.cuda.current_stream().cuda_stream)
tb, sr = w.seeds(); g.reset(tb, sr)
cmds, _ = config.bench_commands(A, 1, 400)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 400)
obs = torch.empty(A * 30752, dtype=torch.float32, device="cuda")
for _ in range(20):
    g.observe_device(obs.data_ptr())
torch.cuda.synchronize()
print("nonzero fraction", float((obs != 0).float().mean()))
