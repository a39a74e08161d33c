"""Wave cycles per phase of k_observe_list (diagnostic build -DSF_DIAG_OBS, tools/ab/libsf_obsdiag.so), configs[2]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_LIBRARY_PATH"] = os.path.join(ROOT, "tools", "ab", "libsf_obsdiag.so")
import torch  # noqa: E402

from strikeforce_amd import config, env  # noqa: E402

w = config.baseline_workload(sys.argv[1] if len(sys.argv) > 1 else "C3", arenas=4096)
g = env.ArenaBatch(w)
g.reset(*w.seeds())
cmds, _ = config.bench_commands(4096, w.cfg.n_agents, 400)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 400)
n = 4096 * w.cfg.n_agents
keys = torch.zeros((n, 2048), dtype=torch.int32, device="cuda")
vals = torch.zeros((n, 2048), dtype=torch.float32, device="cuda")
cnt, pov = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros((n, 160), device="cuda")
import numpy as np  # noqa: E402
out = np.zeros((n, 8), dtype=np.uint32)
g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048)
ev[1].record()
torch.cuda.synchronize()
g.L.sf_diag_obs_read(g.h, out.ctypes.data_as(C.c_void_p), n)
names = ["0 flag loads issued", "1 scatter", "2a classify", "2b records", "pow pass", "pass masks", "3 list", "pov"]
tot = float(out.sum())
print("k_observe_list: %.1f us per launch; mean non-zeros %.0f" % (ev[0].elapsed_time(ev[1]) / 20 * 1e3, float(cnt.float().mean())))
for k in range(8):
    print("  %-22s %8.0f cycles per wave (max %6d)  %5.1f %%" % (names[k], out[:, k].mean(), out[:, k].max(), 100.0 * out[:, k].sum() / tot))
print("  wave life: mean %.0f, max %d cycles" % (out.sum(axis=1).mean(), out.sum(axis=1).max()))
