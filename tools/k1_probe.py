"""Cost of short launches on configs[1] with auto-reset: K = 1, 2, 4, 8 per launch at the same point of the run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strikeforce_amd import config, env
A = 4096
for ar in (1, 0):
    w = config.baseline_workload("C2", arenas=A, auto_reset=ar)
    g = env.ArenaBatch(w); g.set_stream(torch.cuda.current_stream().cuda_stream); g.reset(*w.seeds())
    cmds, _ = config.bench_commands(A, 1, 1400)
    d = torch.from_numpy(cmds).cuda()
    g.step_device(d.data_ptr(), 400); torch.cuda.synchronize()
    pos = 400
    for K in (1, 2, 4, 8):
        n = 40
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n):
            g.step_device(d.data_ptr() + pos * A, K); pos += K
        b.record(); torch.cuda.synchronize()
        print("auto_reset=%d K=%d: %.1f us per launch" % (ar, K, a.elapsed_time(b) / n * 1e3))
# distribution of K=1 launch times with auto-reset (events per launch)
w = config.baseline_workload("C2", arenas=A, auto_reset=1)
g = env.ArenaBatch(w); g.set_stream(torch.cuda.current_stream().cuda_stream); g.reset(*w.seeds())
g.step_device(d.data_ptr(), 400); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
ev[0].record()
for i in range(200):
    g.step_device(d.data_ptr() + (400 + i) * A, 1); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(200))
print("K=1 launch us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (ts[0], ts[20], ts[100], ts[180], ts[-1]))
import numpy as np
ep = np.array([g.dump(a).hdr.episodes for a in range(0, A, 64)])
print("episodes per arena after 600 steps (sample): mean %.2f max %d" % (ep.mean(), ep.max()))
