import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strikeforce_amd import config, env
A = 4096
w = config.baseline_workload("C2", arenas=A, auto_reset=0)
g = env.ArenaBatch(w)
g.set_stream(torch.cuda.current_stream().cuda_stream)
tb, sr = w.seeds(); g.reset(tb, sr)
cmds, _ = config.bench_commands(A, 1, 300)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 300)
obs = torch.empty(A * 30752, dtype=torch.float32, device="cuda")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1000
print("observe live (us):", timeit(lambda: g.observe_device(obs.data_ptr())))
print("torch zero_ 504MB (us):", timeit(lambda: obs.zero_()))
print("torch fill_(1) (us):", timeit(lambda: obs.fill_(1.0)))
kill = torch.full((A,), ord('_'), dtype=torch.uint8, device='cuda')
g.step_device(kill.data_ptr(), 1); g.step_device(kill.data_ptr(), 1); torch.cuda.synchronize()
print("dead agents:", int(g.done().sum()))
print("observe all-dead = zero fill only (us):", timeit(lambda: g.observe_device(obs.data_ptr())))
