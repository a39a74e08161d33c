"""Same-call A/B of the policy forward: python tools/r04_forward_ab.py <library> [agents] — the forward (k_feat_list + k_tail)
on configs[2]'s lists 300 steps in, 5 x 20 forwards, and k_tail's share by the library's own events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_LIBRARY_PATH"] = os.path.join(ROOT, sys.argv[1])
import torch  # noqa: E402

from strikeforce_amd import config, env, policy  # noqa: E402

A = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
w = config.baseline_workload("C3", arenas=A)
g = env.ArenaBatch(w)
g.reset(*w.seeds())
cmds, _ = config.bench_commands(A, 1, 300)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 300)
pb = policy.PolicyBatch(policy.init_parameters(seed=0), A)
keys = torch.zeros((A, 2048), dtype=torch.int32, device="cuda")
vals = torch.zeros((A, 2048), dtype=torch.float32, device="cuda")
cnt, pov = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros((A, 160), device="cuda")
probs, value = torch.zeros((A, 9), device="cuda"), torch.zeros(A, device="cuda")
g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048)
fwd = lambda: pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048, A, probs.data_ptr(), value.data_ptr())
for _ in range(5):
    fwd()
torch.cuda.synchronize()
out = []
for _ in range(5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        fwd()
    ev[1].record()
    torch.cuda.synchronize()
    out.append(ev[0].elapsed_time(ev[1]) / 20 * 1e3)
pb.kernel_time(True)
for _ in range(20):
    fwd()
torch.cuda.synchronize()
by = pb.kernel_time_by_kernel(False)
print("%s: forward %s us; by kernel (ms, flop, launches) / 20: %s" % (sys.argv[1], " ".join("%.1f" % x for x in out),
                                                                       [round(m / 20 * 1e3, 1) for (m, f, n) in by]), flush=True)
