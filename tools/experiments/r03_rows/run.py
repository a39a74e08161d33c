#!/usr/bin/env python3
"""r03 experiment: (1) the row-form generator (rng_rows.hip: four arenas per wavefront, one per 16-lane DPP row) against
the generator's known answers; (2) timing: k_step with free draws (tools/ab/libsf_fake.so, -DSF_EXP_FAKE_DRAWS: a draw = a
v_readlane, what popping a ring would cost) alone and with the row producer making the launch's draws beside it on a
second stream — an upper bound for any design that moves the generator into producer waves.

    gpurun -- python tools/experiments/r03_rows/run.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
MOD = 65537


def tables():
    logt = np.zeros(1024 + MOD, dtype=np.uint16)
    exptab = np.zeros(512, dtype=np.uint32)
    v = 1
    for m in range(65536):
        logt[1024 + v] = m
        if v + 1024 >= MOD:
            logt[1024 + v - MOD] = m
        if m < 256:
            exptab[m] = v
        if m % 256 == 0:
            exptab[256 + m // 256] = v
        v = v * 3 % MOD
    return logt, exptab


def warmed_state(tb, serial, logt):
    """random.hpp:64-76 in plain Python: [18] log | us << 20 | seed << 24 after the 1024 warm-up draws"""
    us, seed, rnd = [], [], [0] * 18
    for _ in range(18):
        us.append(serial % 10 + 1), seed.append(tb % 10 + 1)
        serial //= 10
        tb //= 10
    jomle = 18
    for _ in range(1024):
        s = 1
        for i in range(18):
            s = (s + us[i] * pow(rnd[i], seed[i], MOD)) % MOD
        jomle += 1
        rnd = rnd[1:] + [pow(s + (s == 0), jomle % 65536, MOD)]
    return np.array([int(logt[1024 + rnd[i]]) | us[i] << 20 | seed[i] << 24 for i in range(18)], dtype=np.uint32)


def known_answers(tb, serial, n):
    """the first n results of rand() after _srand(tb, serial), random.hpp:54-76 in plain Python (value & 1023, :61)"""
    us, seed, rnd = [], [], [0] * 18
    for _ in range(18):
        us.append(serial % 10 + 1), seed.append(tb % 10 + 1)
        serial //= 10
        tb //= 10
    jomle, out = 18, []
    for d in range(1024 + n):
        s = 1
        for i in range(18):
            s = (s + us[i] * pow(rnd[i], seed[i], MOD)) % MOD
        jomle += 1
        rnd = rnd[1:] + [pow(s + (s == 0), jomle % 65536, MOD)]
        if d >= 1024:
            out.append(rnd[17] & 1023)
    return out


def main():
    import torch
    L = C.CDLL(os.path.join(HERE, "librngrows.so"))
    L.rr_run.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    logt, exptab = tables()
    d_logt, d_exp = torch.from_numpy(logt.view(np.int16)).cuda(), torch.from_numpy(exptab.view(np.int32)).cuda()
    # (1) correctness: 6 arenas (two wavefronts, one partly filled), 300 draws each
    seeds = [(1700000000 + i, 123456789) for i in range(5)] + [(10 ** 18 - 1, 10 ** 17)]
    st = np.stack([warmed_state(tb, sr, logt) for tb, sr in seeds])
    d_st = torch.from_numpy(st.view(np.int32)).cuda()
    n = 300
    d_out = torch.zeros((len(seeds), n), dtype=torch.int16, device="cuda")
    rc = L.rr_run(d_logt.data_ptr(), d_exp.data_ptr(), d_st.data_ptr(), 18 + 1024, d_out.data_ptr(), len(seeds), n, None)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().astype(np.int64) & 0xffff
    for i, (tb, sr) in enumerate(seeds):
        want = known_answers(tb, sr, n)
        assert list(got[i]) == want, (i, list(got[i][:8]), want[:8])
    print("row-form generator: %d arenas x %d draws equal the known answers (rc %d)" % (len(seeds), n, rc), flush=True)

    # (2) timing
    from strikeforce_amd import config, env
    A, K, LAUNCHES, PRE = 4096, 100, 10, 400
    draws_per_step = float(os.environ.get("DRAWS_PER_STEP", "51.5"))  # configs[2]: profiles/r02d
    rounds = int(draws_per_step * K)
    big = torch.from_numpy(np.tile(st[:1], (A, 1)).view(np.int32)).cuda()
    side = torch.cuda.Stream()

    def measure(lib, producer):
        env.LIB_PATH = os.path.join(ROOT, lib)  # another build of the same library
        env._LIB = None
        w = config.baseline_workload(os.environ.get("WL", "C3"), arenas=A)
        g = env.ArenaBatch(w)
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        g.reset(*w.seeds())
        cmds, _ = config.bench_commands(A, w.cfg.n_agents, PRE + K * LAUNCHES)
        d = torch.from_numpy(cmds).cuda()
        stride = A * w.cfg.n_agents
        for s0 in range(0, PRE, K):
            g.step_device(d.data_ptr() + s0 * stride, K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(LAUNCHES):
            if producer:
                L.rr_run(d_logt.data_ptr(), d_exp.data_ptr(), big.data_ptr(), 18 + 1024, None, A, rounds, C.c_void_p(side.cuda_stream))
            g.step_device(d.data_ptr() + (PRE + i * K) * stride, K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g.close()
        return A * K * LAUNCHES / dt / 1e6

    def producer_alone():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(LAUNCHES):
            L.rr_run(d_logt.data_ptr(), d_exp.data_ptr(), big.data_ptr(), 18 + 1024, None, A, rounds, C.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / LAUNCHES * 1e3

    for rep in range(2):
        print("producer alone: %.3f ms per %d-step launch's draws (%d rounds, 1024 waves)" % (producer_alone(), K, rounds), flush=True)
        print("product k_step                         %.1f M env-steps/s" % measure("tools/ab/libsf_base.so", False), flush=True)
        print("free draws                             %.1f M" % measure("tools/ab/libsf_fake.so", False), flush=True)
        print("free draws + row producer beside it    %.1f M" % measure("tools/ab/libsf_fake.so", True), flush=True)
        print("product k_step + row producer beside   %.1f M (sanity: what the producer's load costs)" % measure("tools/ab/libsf_base.so", True), flush=True)


if __name__ == "__main__":
    main()
