"""When do the waves of a ONE-step k_step launch start, how long do they take to load, step and store?  (diagnostic build
-DSF_DIAG_STAMPS, tools/ab/libsf_diag.so: s_memrealtime stamps per wave)  configs[2], 4096 arenas."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_LIBRARY_PATH"] = os.path.join(ROOT, "tools", "ab", "libsf_diag.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from strikeforce_amd import config, env  # noqa: E402

A = 4096
w = config.baseline_workload("C3", arenas=A)
g = env.ArenaBatch(w)
g.reset(*w.seeds())
cmds, _ = config.bench_commands(A, 1, 460)
d = torch.from_numpy(cmds).cuda()
g.step_device(d.data_ptr(), 400)
for k in (1, 1, 1, 20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    g.step_device(d.data_ptr() + 400 * A, k)
    ev[1].record()
    torch.cuda.synchronize()
    t = np.zeros((A, 4), dtype=np.uint64)
    assert g.L.sf_diag_times_read(g.h, t.ctypes.data_as(C.c_void_p), A) == 0
    t = (t.astype(np.int64) - int(t[:, 0].min())) / 100.0  # us since the first wave's start
    q = lambda x: "min %.1f p50 %.1f p90 %.1f max %.1f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
    print("K=%d launch: %.1f us by HIP events" % (k, ev[0].elapsed_time(ev[1]) * 1e3))
    print("   wave starts (us after the first): " + q(t[:, 0]))
    print("   load  (tables + state):           " + q(t[:, 1] - t[:, 0]))
    print("   the step(s):                      " + q(t[:, 2] - t[:, 1]))
    print("   store:                            " + q(t[:, 3] - t[:, 2]))
    print("   wave ends (us after the first start): " + q(t[:, 3]))
    if k == 1:  # who are the slowest?  (population, and whether the arena's game has just restarted: hdr.steps == 0)
        st = t[:, 2] - t[:, 1]
        order = np.argsort(-st)
        steps_of = {}
        for a in list(order[:8]) + list(order[2040:2044]):
            dmp = g.dump(int(a))
            steps_of[int(a)] = (round(float(st[a]), 1), int(dmp.hdr.steps), sum(z.alive for z in dmp.zombies), sum(h.alive for h in dmp.humans),
                                sum(b.alive for b in dmp.bullets))
        print("   slowest eight, then four median ones: arena -> (step us, steps into its game, zombies, humans, bullets):", steps_of)
        print("   step time: p99 %.1f p99.9 %.1f; waves above 22 us: %d" % (np.percentile(st, 99), np.percentile(st, 99.9), int((st > 22).sum())))
