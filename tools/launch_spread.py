#!/usr/bin/env python3
"""How evenly do the 4096 wavefronts of a K-step launch finish?  (diagnostic build, -DSF_DIAG_STAMPS)
    SF_LIBRARY_PATH=$PWD/tools/ab/libsf_diag.so python3 tools/launch_spread.py C3
A wave's phase cycles add up to its lifetime inside the launch; the launch lasts as long as the longest one."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from strikeforce_amd import config, env  # noqa: E402

for wl in sys.argv[1:] or ["C3"]:
    A, K = 4096, 100
    w = config.baseline_workload(wl, arenas=A)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    cmds, _ = config.bench_commands(A, w.cfg.n_agents, 600)
    d = torch.from_numpy(cmds).cuda()
    stride = A * w.cfg.n_agents
    for s in range(0, 400, K):
        g.step_device(d.data_ptr() + s * stride, K)
    g.synchronize()
    L = env.load_library()
    L.sf_diag_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    for rep in range(2):
        g.step_device(d.data_ptr() + (400 + rep * K) * stride, K)
        out = np.zeros((A, 16), dtype=np.uint32)
        assert L.sf_diag_read(g.h, out.ctypes.data, A) == 0
        tot = out[:, :14].astype(np.float64).sum(axis=1)
        q = np.percentile(tot, [1, 25, 50, 75, 99])
        print("%s launch %d: lifetime cycles mean %.0f  p1 %.0f p25 %.0f p50 %.0f p75 %.0f p99 %.0f max %.0f  (max / mean = %.3f)"
              % (wl, rep, tot.mean(), q[0], q[1], q[2], q[3], q[4], tot.max(), tot.max() / tot.mean()))
        # waves that share a SIMD (consecutive arenas land on consecutive CUs; 4096 = 256 CUs x 16): spread of per-CU sums
        per_cu = tot.reshape(16, 256).sum(axis=0)
        print("   per-CU sum of its 16 waves: max / mean = %.3f" % (per_cu.max() / per_cu.mean()))
    g.close()
