"""K = 1 launches of configs[2] (4096 arenas, 1500 steps in): python tools/r04_k1_ab.py <library> — ms per one-step launch
and per step + list observation, 3 x 100 each."""
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
cmds, _ = config.bench_commands(A, 1, 1500 + 700)
d = torch.from_numpy(cmds).cuda()
for s0 in range(0, 1500, 100):
    g.step_device(d.data_ptr() + s0 * A, 100)
keys = torch.zeros((A, 2048), dtype=torch.int32, device="cuda")
vals = torch.zeros((A, 2048), dtype=torch.float32, device="cuda")
cnt, pov = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros((A, 160), device="cuda")
torch.cuda.synchronize()
s = 1500
out = []
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(100):
        g.step_device(d.data_ptr() + s * A, 1)
        s += 1
    ev[1].record()
    for _ in range(100):
        g.step_device(d.data_ptr() + s * A, 1)
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048)
        s += 1
    ev[2].record()
    torch.cuda.synchronize()
    out.append((ev[0].elapsed_time(ev[1]) / 100 * 1e3, ev[1].elapsed_time(ev[2]) / 100 * 1e3))
print("%s: K=1 step %s us; step + list observation %s us" % (sys.argv[1], " ".join("%.1f" % a for a, _ in out), " ".join("%.1f" % b for _, b in out)), flush=True)
