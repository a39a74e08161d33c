#!/usr/bin/env python3
"""Wave cycles per tick phase from the diagnostic build (-DSF_DIAG_STAMPS, never the product library):

    SF_LIBRARY_PATH=$PWD/tools/ab/libsf_diag.so python3 tools/diag_stamps.py C2 C3

Prints, per workload, the average over arenas of the s_memtime cycles one K=100 launch spends in each phase, per step."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from strikeforce_amd import config, env  # noqa: E402

NAMES = ["(cmd load)", "zombie_action", "portal_damage", "human_action", "prewarm", "update_tmp", "hits", "update_bull",
         "loop_top", "h:get_command", "h:obey", "h:teleport", "h:claim_chest", "z:precompute"]
for wl in sys.argv[1:] or ["C2", "C3"]:
    A, K = 4096, 100
    w = config.baseline_workload(wl, arenas=A)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    cmds, _ = config.bench_commands(A, w.cfg.n_agents, 500)
    d = torch.from_numpy(cmds).cuda()
    stride = A * w.cfg.n_agents
    for s in range(0, 400, K):
        g.step_device(d.data_ptr() + s * stride, K)
    g.synchronize()
    g.step_device(d.data_ptr() + 400 * stride, K)
    out = np.zeros((A, 16), dtype=np.uint32)
    L = env.load_library()
    L.sf_diag_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    assert L.sf_diag_read(g.h, out.ctypes.data, A) == 0
    per = out.astype(np.float64).mean(axis=0) / K
    tot = per[:14].sum()
    print("%s: %.0f wave cycles per arena-step" % (wl, tot))
    life = out[:, :14].astype(np.float64).sum(axis=1) / K  # per arena: stamped cycles per step over this launch
    order = np.argsort(life)
    print("   per arena: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f (max / mean %.2f)"
          % (life.mean(), life[order[A // 2]], life[order[int(A * 0.9)]], life[order[int(A * 0.99)]], life.max(), life.max() / life.mean()))
    top = order[-41:]  # the busiest 1 %: where do THEIR cycles go?
    pt = out[top].astype(np.float64).mean(axis=0) / K
    print("   busiest 1 %% of the arenas: %.0f cycles per step: " % pt[:14].sum() +
          ", ".join("%s %.0f" % (NAMES[i], pt[i]) for i in np.argsort(-pt[:14])[:6]))
    for i, n in enumerate(NAMES):
        print("   %-14s %8.0f  %5.1f %%" % (n, per[i], 100 * per[i] / tot))
    g.close()
