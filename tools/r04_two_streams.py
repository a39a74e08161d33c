"""The interactive loop (one step per launch + the list observation) and the closed policy loop driven as S independent
part-batches on S HIP streams instead of one batch on one stream: the launch ramps and tails of one part's kernels are
filled by the other parts' work.  configs[2], 4096 arenas in all."""
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from strikeforce_amd import config, env, policy  # noqa: E402

TOTAL, STEPS, CAP = 4096, 60, 2048


def parts(S, with_policy):
    out = []
    for j in range(S):
        n = TOTAL // S
        w = config.baseline_workload("C3", arenas=n)
        w.cfg.reseed_stride = TOTAL
        g = env.ArenaBatch(w)
        st = torch.cuda.Stream()
        g.set_stream(st.cuda_stream)
        g.reset(*w.seeds(first_arena=j * n))
        cmds, _ = config.bench_commands(n, 1, 400 + STEPS, seed0=12345 + j * n)
        with torch.cuda.stream(st):
            d = torch.from_numpy(cmds).cuda()
            buf = dict(keys=torch.zeros((n, CAP), dtype=torch.int32, device="cuda"), vals=torch.zeros((n, CAP), dtype=torch.float32, device="cuda"),
                       cnt=torch.zeros(n, dtype=torch.int32, device="cuda"), pov=torch.zeros((n, 160), device="cuda"),
                       probs=torch.zeros((n, 9), device="cuda"), value=torch.zeros(n, device="cuda"), cmd=torch.zeros(n, dtype=torch.uint8, device="cuda"),
                       new=torch.zeros(n, dtype=torch.uint8, device="cuda"), dense=torch.empty((n, 30752), device="cuda") if with_policy else None)
        st.synchronize()
        g.step_device(d.data_ptr(), 400)
        pb = None
        if with_policy:
            pb = policy.PolicyBatch(policy.init_parameters(seed=0), n)
            pb.set_stream(st.cuda_stream)
        out.append((g, n, d, buf, pb, st))
    torch.cuda.synchronize()
    return out


def interactive(P):
    def body(s):
        for g, n, d, b, _pb, _st in P:
            g.step_device(d.data_ptr() + (400 + s) * n, 1)
            g.observe_sparse_device(b["keys"].data_ptr(), b["vals"].data_ptr(), b["cnt"].data_ptr(), b["pov"].data_ptr(), CAP)
    for s in range(5):
        body(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(5, STEPS):
        body(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (STEPS - 5)


def closed(P):
    views = [g.done_view_device() for g, *_ in P]

    def body():
        for (g, n, _d, b, pb, _st), view in zip(P, views):
            g.observe_sparse_device(b["keys"].data_ptr(), b["vals"].data_ptr(), b["cnt"].data_ptr(), b["pov"].data_ptr(), CAP)
            g.observe_overflow_device(b["cnt"].data_ptr(), CAP, b["dense"].data_ptr(), b["pov"].data_ptr())
            pb.predict_sparse(b["keys"].data_ptr(), b["vals"].data_ptr(), b["cnt"].data_ptr(), b["pov"].data_ptr(), CAP, n, b["probs"].data_ptr(),
                              b["value"].data_ptr(), b["cmd"].data_ptr(), seed=0, d_dense_ptr=b["dense"].data_ptr(), reset_words=view)
            g.step_device(b["cmd"].data_ptr(), 1)
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20


for S in ((1, 2, 4, 8) if "--closed-only" not in sys.argv else ()):
    P = parts(S, False)
    dt = interactive(P)
    print("interactive, list observation, %d part(s): %.4f ms per step of %d arenas = %.1f M env-steps/s" % (S, dt * 1e3, TOTAL, TOTAL / dt / 1e6), flush=True)
    del P
for S in (1, 2, 4):
    P = parts(S, True)
    dt = closed(P)
    print("closed policy loop, %d part(s): %.4f ms per step = %.2f M agent-steps/s" % (S, dt * 1e3, TOTAL / dt / 1e6), flush=True)
    del P
