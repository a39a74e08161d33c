"""Why does the one-launch-per-step loop run faster in tools/r04_two_streams.py than in bench.py?  Same loop, variations."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from strikeforce_amd import config, env  # noqa: E402

A, CAP = 4096, 2048
w = config.baseline_workload("C3", arenas=A)
g = env.ArenaBatch(w)
if "own" in sys.argv:
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.reset(*w.seeds())
n_pre = int(os.environ.get("PRE", "400"))
cmds, _ = config.bench_commands(A, 1, n_pre + 200)
d = torch.from_numpy(cmds).cuda()
for s0 in range(0, n_pre, 100):
    g.step_device(d.data_ptr() + s0 * A, 100)
keys = torch.zeros((A, CAP), dtype=torch.int32, device="cuda")
vals = torch.zeros((A, CAP), dtype=torch.float32, device="cuda")
cnt, pov = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros((A, 160), device="cuda")


def loop(n, first):
    for s in range(n):
        g.step_device(d.data_ptr() + (first + s) * A, 1)
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), CAP)


for rep in range(3):
    loop(5, n_pre)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(40, n_pre + 5)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 40
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    loop(40, n_pre + 50)
    e[1].record()
    torch.cuda.synchronize()
    print("pre-roll %d, %s stream: wall %.4f ms per step (%.1f M), events %.4f ms (%.1f M); mean non-zeros %.0f" % (
        n_pre, "own" if "own" in sys.argv else "null", wall * 1e3, A / wall / 1e6, e[0].elapsed_time(e[1]) / 40, A / (e[0].elapsed_time(e[1]) / 40) / 1e3,
        float(cnt.float().mean())), flush=True)
