"""TEST INFRASTRUCTURE (uses the oracle / the reference build as the checker).  Soak: the HIP path against the oracle at the size bench.py reports, far longer than the test suite runs — every arena's
digest at several checkpoints of a long run (launches of 100 steps: the launch order by population is renewed on the way,
Timer episodes run out and restart on the seed the 4096-arena run gives them).  One JSON line per configuration.
Usage (GPU box): python tests/tools/soak_parity.py [C3:4000 C2:4000 C4:600 C5:300]"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

A, CHUNK, K = 4096, 128, 100


def oracle_chunk(job):
    name, first, steps, marks = job
    from oracle_lib import Oracle
    from strikeforce_amd import config
    wc = config.baseline_workload(name, arenas=CHUNK)
    wc.cfg.reseed_stride = A
    cmds, _ = config.bench_commands(A, wc.cfg.n_agents, steps)
    o = Oracle(wc)
    o.reset(*wc.seeds(first_arena=first))
    out, at = [], 0
    for m in marks:
        o.step_many(cmds[at:m, first:first + CHUNK])
        at = m
        out.append(o.digest().copy())
    ep = int(np.sum([o.dump(i).hdr.episodes for i in range(CHUNK)]))
    o.close()
    return first, out, ep


def main():
    import torch
    from strikeforce_amd import config, env
    specs = sys.argv[1:] or ["C3:4000", "C2:4000", "C4:600", "C5:300"]
    for spec in specs:
        name, steps = spec.split(":")
        steps = int(steps)
        marks = sorted({steps // 4, steps // 2, 3 * steps // 4, steps})
        marks = [m - m % K or K for m in marks]
        marks = sorted(set(marks))
        w = config.baseline_workload(name, arenas=A)
        n = w.cfg.n_agents
        g = env.ArenaBatch(w)
        g.reset(*w.seeds())
        cmds, _ = config.bench_commands(A, n, steps)
        d = torch.from_numpy(cmds).cuda()
        got, at = [], 0
        t0 = time.time()
        for m in marks:
            for s in range(at, m, K):
                g.step_device(d.data_ptr() + s * A * n, min(K, m - s))
            at = m
            g.synchronize()
            got.append(g.digest().copy())
        t_gpu = time.time() - t0
        g.close()
        t0 = time.time()
        with mp.Pool(min(12, os.cpu_count() or 1)) as pool:
            res = pool.map(oracle_chunk, [(name, first, marks[-1], marks) for first in range(0, A, CHUNK)])
        bad, episodes = 0, 0
        for first, outs, ep in res:
            episodes += ep
            for i, want in enumerate(outs):
                bad += int(np.count_nonzero(want != got[i][first:first + CHUNK]))
        print(json.dumps({"workload": name, "arenas": A, "steps": marks[-1], "checkpoints": marks, "steps_per_launch": K,
                          "arena_digests_compared": A * len(marks), "differing": bad, "episodes_ended_and_restarted": episodes,
                          "gpu_s": round(t_gpu, 2), "oracle_s": round(time.time() - t0, 1)}), flush=True)
        assert bad == 0


if __name__ == "__main__":
    mp.set_start_method("spawn")
    main()
