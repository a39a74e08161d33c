import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from strikeforce_amd import config, env
w = config.baseline_workload("C2", arenas=4096, auto_reset=0)
g = env.ArenaBatch(w)
g.set_stream(torch.cuda.current_stream().cuda_stream)
tb, sr = w.seeds(); g.reset(tb, sr)
cmds, _ = config.bench_commands(4096, 1, 600)
d = torch.from_numpy(cmds).cuda()
def timeit(fn, n=40):
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1000
g.step_device(d.data_ptr(), 400); torch.cuda.synchronize()
print("K=1 live steps (us):", timeit(lambda i: g.step_device(d.data_ptr()+ (400+i)*4096, 1)))
print("K=2 (us per launch):", timeit(lambda i: g.step_device(d.data_ptr()+ (440+2*i)*4096, 2)))
print("K=10 (us per launch):", timeit(lambda i: g.step_device(d.data_ptr()+ (0+10*i)*4096, 10), 10))
kill = torch.full((4096,), ord('_'), dtype=torch.uint8, device='cuda')
g.step_device(kill.data_ptr(), 1); g.step_device(kill.data_ptr(), 1); torch.cuda.synchronize()
print("done arenas:", g.done().sum())
print("K=1 all-done no-op (us):", timeit(lambda i: g.step_device(d.data_ptr(), 1)))
print("K=50 all-done no-op (us):", timeit(lambda i: g.step_device(d.data_ptr(), 50)))
