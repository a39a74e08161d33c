"""Cycles per phase of k_tail (diagnostic build -DSF_DIAG_TAIL, tools/ab/libsf_taildiag.so): closed loop on configs[2]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_LIBRARY_PATH"] = os.path.join(ROOT, "tools", "ab", os.environ.get("SF_TAIL_DIAG_LIB", "libsf_taildiag.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from strikeforce_amd import config, env, policy  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048, A, probs.data_ptr(), value.data_ptr())
torch.cuda.synchronize()
ev[0].record()
for _ in range(10):
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), pov.data_ptr(), 2048, A, probs.data_ptr(), value.data_ptr())
ev[1].record()
torch.cuda.synchronize()
out = np.zeros((A // 16, 16, 24), dtype=np.uint32)
assert g.L.sf_policy_diag_tail_read(out.ctypes.data_as(C.c_void_p), A // 16) == 0
names = ["prologue loads", "barrier", "feat -> feat_n", "barrier", "gru0 tiles (60)", "barrier", "gru cell + comb row", "barrier",
         "comb tiles (10, K=352)", "barrier", "gated_n row", "barrier", "gru1 tiles (60)", "barrier", "gru cell + out rows", "barrier",
         "ResB tiles (20, last layer)", "barrier", "ResB row (last layer)", "barrier", "head tiles (2)", "barrier", "softmax / store"]
print("forward: %.1f us" % (ev[0].elapsed_time(ev[1]) / 10 * 1e3))
tot = out[:, :, :23].sum(axis=2).astype(np.float64)
tot += 2 * out[:, :, 16:20].sum(axis=2)  # (the ResB loop runs three times; the stamps keep the last pass)
print("wave life (with the ResB loop x3): mean %.0f cycles, max %.0f" % (tot.mean(), tot.max()))
for k in range(23):
    x = out[:, :, k].astype(np.float64)
    print("  %2d %-28s mean %7.0f   wave 0 %7.0f   max over waves (mean over workgroups) %7.0f" % (k, names[k], x.mean(), x[:, 0].mean(), x.max(axis=1).mean()))
# per wave index: the tile phases, and where the wave ran (HW_ID, slot 23: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13)
hw = out[:, :, 23]
print("wave:        " + " ".join("%6d" % i for i in range(16)))
for k in (0, 4, 6, 8, 12, 16, 18):
    print("phase %2d:    " % k + " ".join("%6.0f" % v for v in out[:, :, k].astype(np.float64).mean(axis=0)))
print("SIMD (wg 0): " + " ".join("%6d" % ((int(v) >> 4) & 3) for v in hw[0]))
print("SIMD (wg 1): " + " ".join("%6d" % ((int(v) >> 4) & 3) for v in hw[1]))
print("SIMD (wg 9): " + " ".join("%6d" % ((int(v) >> 4) & 3) for v in hw[9]))
cu = ((hw[:, 0] >> 8) & 15) | (((hw[:, 0] >> 12) & 1) << 4) | (((hw[:, 0] >> 13) & 7) << 5)
print("workgroups per (SE, SH, CU) id seen by wave 0 [id: count], XCC not in HW_ID:", dict(zip(*np.unique(cu, return_counts=True))))
s4 = np.zeros(4)
n4 = np.zeros(4)
for wg in range(out.shape[0]):
    for wv in range(16):
        s = (int(hw[wg, wv]) >> 4) & 3
        s4[s] += out[wg, wv, 4]
        n4[s] += 1
print("gru0 tile phase by SIMD: mean cycles", (s4 / np.maximum(n4, 1)).round(0), "waves", n4)
