"""Cost of an in-kernel restart: one K=1 launch in which every arena of configs[1] restarts ('_' kills the player)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strikeforce_amd import config, env
A = 4096
w = config.baseline_workload("C2", arenas=A, auto_reset=1)
g = env.ArenaBatch(w); g.set_stream(torch.cuda.current_stream().cuda_stream); g.reset(*w.seeds())
cmds, _ = config.bench_commands(A, 1, 500)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 300); torch.cuda.synchronize()   # warm-up of the next generator complete (64 steps)
kill = torch.full((A,), ord('_'), dtype=torch.uint8, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(); g.step_device(d.data_ptr() + 300 * A, 1); ev[1].record()
g.step_device(kill.data_ptr(), 1); ev[2].record()
g.step_device(d.data_ptr() + 301 * A, 1); ev[3].record()
torch.cuda.synchronize()
print("normal K=1 %.1f us; K=1 with every arena restarting %.1f us; the step after %.1f us; restarted: %d"
      % (ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3, int(g.done().sum())))
