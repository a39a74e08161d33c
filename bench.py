#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched StrikeForce arena simulator on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch: every arena of every rank advances one iteration of
the reference loop (gameplay.hpp:1443-1472).  Workload = BASELINE.json configs[2], the configuration the metric is
quoted on ("64x64 map x32 entities"): 4096 arenas per GPU, 64x64 map, 8 humans + 24 zombies (+ 64 bullet slots),
Timer mode, random-action agent, NPC humans on human_rnpc_bot; arenas are sharded across ranks
with no data-path collective ("weak" scaling: 4096 arenas per GPU); the only collective is the RCCL all-gather of
the end-of-episode result records after each launch, issued through the library's C-ABI (sf_results_allgather) on
a side stream.  Commands are resident in HBM before the timed region.  Before --warmup, an untimed pre-roll
(PREROLL steps, full-length launches) brings zombie/NPC populations and clocks to steady state, so that the timed
region is representative whatever --steps is.  One JSON line is printed by rank 0.

With --gpus N > 1 and no torch.distributed environment, this process starts the N ranks itself (torch.distributed.run
as a child process, before anything here touches a GPU) and passes their output through.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREROLL = 400  # untimed steps before --warmup (steady-state populations: a zombie every 20 steps, an NPC every 25)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: f32-input MFMA (v_mfma_f32_32x32x2_f32), dense
L2_GATHER_GBS = 17800.0  # same guide, "Indexed rows": rows gathered out of the XCDs' L2s, measured 16.8-18.8 TB/s chip-wide
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: bf16 MFMA, dense (never the 2:1-sparsity figure)


def algorithmic_bytes_per_step(cfg, observe=False):
    """SURVEY.md §8(d) contract figure per arena-step (caps used for live counts)."""
    H, Z, B, A = cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.n_agents
    b = 2 * (128 * H + 16 * Z + 32 * B) + 2 * 80 + 16 * 8 * (H + Z) + 2 * 2 * 8 * 2 * B
    if observe:
        b += A * (123008 + 961 * 8)
    return b


def pmc_traffic(workload, arenas, kpl):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in
    their own runs and corrected as MI355X_MICROARCH.md prescribes; tools/summarize_prof.py), or None if this
    launch shape was not profiled."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        return d["%s/%d/%d" % (workload, arenas, kpl)]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def issue_bound(workload, arenas):
    """What actually bounds k_step — vector instruction issue — from the committed rocprofv3 PMC passes of this workload
    (profiles/instr_mix_per_arena_step.json, tools/pmc_report.py): instructions per arena-step and the fraction of the
    SIMDs' cycles in which a vector instruction was issued.  None when this shape was not profiled."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "instr_mix_per_arena_step.json")))
        m = d["%s/%d" % (workload, arenas)]
        return {"valu_per_arena_step": m["SQ_INSTS_VALU"], "salu_per_arena_step": m["SQ_INSTS_SALU"],
                "branch_per_arena_step": m["SQ_INSTS_BRANCH"], "lds_per_arena_step": m["SQ_INSTS_LDS"],
                "wave_cycles_per_arena_step": m["SQ_WAVE_CYCLES"] * 4,
                "simd_busy_frac": m["simd_valu_busy_frac"], "source": m["source"]}
    except Exception:  # noqa: BLE001
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_port(workload_name, seconds=2.0):
    """The oracle (a single-threaded CPU port of the reference loop) timed on this host, on a bounded sample:
    64 arenas of the same workload, as many 250-step rounds as fit in ~`seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from strikeforce_amd import config
    arenas = 64
    w = config.baseline_workload(workload_name, arenas=arenas)
    o = oracle_lib.Oracle(w)
    tb, sr = w.seeds()
    o.reset(tb, sr)
    lcg = (C.c_uint32 * (arenas * w.cfg.n_agents))(*[12345 + i for i in range(arenas * w.cfg.n_agents)])
    steps, t0 = 0, time.perf_counter()
    while True:
        steps += o.L.sfo_bench_run(o.h, 250, lcg)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d arenas of %s, %d arena-steps in %.1f s, oracle/sf_oracle.c single thread on %s"
                      % (arenas, workload_name, steps, dt, _cpu_model())}


def cpu_reference(workload_name, seconds=3.0):
    """The REFERENCE's own tick path timed on this host: oracle/_ref/sf_ref_tick_<dims> = the client's gameplay.hpp /
    Character.hpp / Item.hpp / random.hpp compiled head-less for this workload's dimensions (oracle/ref_tick.py; built by
    __graft_entry__.build() where the checkout exists, the binary travels), one arena, the same random-action agent,
    restarting when the player dies.  Its data files (Items/, character/human_enemy.txt, the map) are written here from
    the workload's own tables.  None where the binary is not there."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tick
    from strikeforce_amd import config
    w = config.baseline_workload(workload_name, arenas=1)
    cfg = w.cfg
    dims = (cfg.floors, cfg.rows, cfg.cols, cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.cap_chests)
    exe = ref_tick.binary_for(dims)
    if not os.path.exists(exe) or cfg.mode not in (0, 1, 2):
        return None
    d = tempfile.mkdtemp(prefix="sf_refbench_")
    try:
        for sub in ("map", "Items", "character"):
            os.makedirs(os.path.join(d, sub))
        it = cfg.items
        for i in range(4):  # name price vol lvl stamina Hp effect   (Item.hpp:41-45)
            open(os.path.join(d, "Items", "cons%d.txt" % i), "w").write("cons%d 0 1 1 %d %d %d\n" % ((i,) + tuple(it.cons[i])))
        for i in range(4):  # name price vol lvl stamina damage effect range   (Item.hpp:121-125)
            open(os.path.join(d, "Items", "throw%d.txt" % i), "w").write("throw%d 0 1 1 %d %d %d %d\n" % ((i,) + tuple(it.thr[i])))
        for i in range(8):  # Item.hpp:85-89, level 0 in the file
            open(os.path.join(d, "Items", "w%d.txt" % i), "w").write("w%d 0 1 0 %d %d %d %d\n" % ((i,) + tuple(it.weapon[i])))
        open(os.path.join(d, "character", "human_enemy.txt"), "w").write(" ".join(str(t) for t in config.HUMAN_ENEMY_TOKENS) + "\n")
        open(os.path.join(d, "profile.txt"), "w").write("player " + " ".join(str(t) for t in config.HUMAN_ENEMY_TOKENS) + "\n")
        cells = cfg.rows * cfg.cols
        chars, portal = bytes(w._map.raw).decode("ascii"), list(w._portal)
        for f in range(cfg.floors):
            open(os.path.join(d, "map", "floor%d.txt" % (f + 1)), "w").write(
                config.format_floor_text(chars[f * cells:(f + 1) * cells], portal[f * cells:(f + 1) * cells], cfg.rows, cfg.cols))
        mode = {0: "Solo", 1: "Timer", 2: "Squad"}[cfg.mode]
        script = "init profile.txt %s %d 0\nreset 1700000000 123456789\nbench 2000 12345\n" % (mode, cfg.level)
        # a first short run sizes the sample; then one run of about `seconds`
        r = subprocess.run([exe], input=script + "quit\n", cwd=d, capture_output=True, text=True, timeout=60)
        ok = [ln for ln in r.stdout.split("\n") if ln.startswith("ok ") and len(ln.split()) == 4]
        if r.returncode != 0 or not ok:
            return None
        n0, s0 = int(ok[-1].split()[1]), float(ok[-1].split()[2])
        n = max(2000, int(n0 / max(s0, 1e-6) * seconds))
        r = subprocess.run([exe], input=script.replace("bench 2000", "bench %d" % n) + "quit\n", cwd=d, capture_output=True,
                           text=True, timeout=120)
        ok = [ln for ln in r.stdout.split("\n") if ln.startswith("ok ") and len(ln.split()) == 4]
        if r.returncode != 0 or not ok:
            return None
        steps, sec, resets = int(ok[-1].split()[1]), float(ok[-1].split()[2]), int(ok[-1].split()[3])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"value": steps / sec, "unit": "env-steps/s", "cores": 1, "kind": "reference",
            "sample": "the reference's own gameplay.hpp tick path (%s: its headers compiled head-less for %dx%dx%d, pools "
                      "H%d Z%d B%d, oracle/ref_tick.py), 1 arena of %s, %d steps in %.2f s (%d games), single thread on %s "
                      "(host has %d logical cores; the reference loop is single-threaded, gameplay.hpp:1443)"
                      % (os.path.basename(exe), cfg.floors, cfg.rows, cfg.cols, cfg.cap_humans, cfg.cap_zombies,
                         cfg.cap_bullets, workload_name, steps, sec, resets + 1, _cpu_model(), os.cpu_count() or 0)}


def cpu_baseline(workload_name):
    """~5 s of CPU work: the reference itself where its head-less build is at hand (kind "reference"), with this repo's
    CPU port of the same loop (the oracle) timed beside it; the port alone otherwise."""
    port = cpu_port(workload_name)
    ref = None
    try:
        ref = cpu_reference(workload_name)
    except Exception as e:  # noqa: BLE001  (a baseline that cannot run must not take the bench line with it)
        port["reference_error"] = repr(e)[:200]
    if ref is None:
        return port
    ref["port"] = port
    return ref


WORKLOAD_TEXT = {
    "C2": "BASELINE configs[1] (C2)", "C3": "BASELINE configs[2] (C3)", "C4": "BASELINE configs[3] (C4)",
    "C5": "BASELINE configs[4] (C5)",
    "NATIVEGAME": "the reference's own world (not a BASELINE config: gameplay.hpp:37's 3 x 30 x 100 map, a whole Timer game, "
                  "pools of 1024 zombies / 512 exits — the large-pool kernel, tables in LDS)"}


def describe_workload(name, arenas, cfg):
    mode = {0: "Solo", 1: "Timer", 2: "Squad", 3: "Battle"}.get(cfg.mode, str(cfg.mode))
    return ("%s: %d arenas/GPU, %dx%d map, %d human + %d zombie + %d bullet slots, %s level %d, %d commanded "
            "agent(s) per arena on the random-action agent" % (WORKLOAD_TEXT.get(name, name), arenas, cfg.rows, cfg.cols,
                                                              cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, mode,
                                                              cfg.level, cfg.n_agents))


def other_configs(args, local, torch, config, env, skip):
    """The same throughput measurement (pre-roll, commands resident, K steps per launch, one GPU, 4096 arenas) on the
    other BASELINE configurations that run on a GPU; parity for all of them is in tests/."""
    out = {}
    for name in ("C2", "C3", "C4", "C5", "NATIVEGAME"):
        if name == skip:
            continue
        w = config.baseline_workload(name, arenas=args.arenas, device=local)
        g = env.ArenaBatch(w)
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        g.reset(*w.seeds())
        steps, pre = 300, PREROLL
        cmds, _ = config.bench_commands(args.arenas, w.cfg.n_agents, steps + pre)
        d = torch.from_numpy(cmds).cuda()
        stride = args.arenas * w.cfg.n_agents
        for s0 in range(0, pre, args.k_per_launch):
            g.step_device(d.data_ptr() + s0 * stride, min(args.k_per_launch, pre - s0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s0 in range(pre, pre + steps, args.k_per_launch):
            g.step_device(d.data_ptr() + s0 * stride, min(args.k_per_launch, pre + steps - s0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g.close()
        del d
        out[describe_workload(name, args.arenas, w.cfg)] = {
            "value": args.arenas * steps / dt, "unit": "env-steps/s", "steps": steps, "preroll": pre,
            "algorithmic_bytes_per_arena_step": algorithmic_bytes_per_step(w.cfg)}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as a child process
    (nothing in this process has touched a GPU) and pass its output and exit code through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env_ = dict(os.environ)
    env_.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--arenas", type=int, default=4096, help="arenas per GPU")
    ap.add_argument("--k-per-launch", type=int, default=100, help="loop iterations per kernel launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-interactive", action="store_true", help="skip the K=1 + observation measurement")
    ap.add_argument("--no-policy", action="store_true", help="skip the closed loop with the on-device policy network")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--preroll", type=int, default=PREROLL, help="untimed steady-state steps before --warmup")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from strikeforce_amd import config, env, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the simulator has no CPU path")
    # The one RCCL communicator of a multi-GPU run is the library's own (sf_comm_init, behind the C-ABI): torch.distributed
    # only carries rank 0's 128-byte id, the barriers and the MAX over ranks, and does that over gloo — a second RCCL
    # communicator (torch's "nccl" group) beside the library's would double the bootstrap and the device buffers for
    # nothing.  SF_BENCH_BACKEND=gloo + SF_BENCH_DEVICE=0 rehearse the N > 1 control flow on a one-GPU box (two ranks on
    # one card, which RCCL refuses): the records then go through host memory.  The driver's runs use neither.
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("SF_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    # Everything runs on a stream of its own, made torch's current stream (the HIP events below — torch's and the
    # library's — are recorded on it): launches on the NULL stream pay its implicit ordering against every other stream,
    # ~10 % on the one-launch-per-step loops (tools/r04_two_streams.py: 52 M against 44-48 M env-steps/s)
    torch.cuda.set_stream(torch.cuda.Stream())
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo")

    w = config.baseline_workload(args.workload, arenas=args.arenas, device=local)
    cfg = w.cfg
    cfg.reseed_stride = world * args.arenas  # shards never reuse a seed
    g = env.ArenaBatch(w)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    tb, sr = shard.shard_seeds(w, rank)
    g.reset(tb, sr)

    pre = args.preroll
    # a timed region shorter than ~50 ms (the driver's --steps 20 is one 0.4 ms launch) is one sample of one launch:
    # it is then repeated REPEATS times on fresh command windows, each repeat bracketed like the first, and the line
    # reports the median (min / max beside it).  Every repeat times EXACTLY --steps steps.
    REPEATS, SHORT_S = 21, 0.05
    # the repeats' command windows are generated (and uploaded) only if the first timed region turns out short: the
    # default --steps 1000 never repeats, and 21 windows of it would be 88 MB per agent for nothing
    total = pre + args.warmup + args.steps
    cmds, _ = config.bench_commands(args.arenas, cfg.n_agents, total, seed0=shard.command_seed(w, rank))
    d_cmds = torch.from_numpy(cmds).cuda()  # resident in HBM before the timed region
    stride = args.arenas * cfg.n_agents
    # end-of-episode result records, all-gathered over RCCL after every launch.  The gather of launch i
    # runs on RCCL's stream while launch i+1 computes: two buffer pairs, and a launch only waits for the gather that
    # used its pair two launches ago.  SF_BENCH_FORCE_GATHER=1 exercises this path with one rank.
    gather = world > 1 or os.environ.get("SF_BENCH_FORCE_GATHER") == "1"
    n_rec = args.arenas * cfg.n_agents * 8
    rccl = gather and backend == "nccl"
    if rccl:
        # the communicator is the library's own (sf_comm_init); torch.distributed (gloo) only carries rank 0's unique id.
        # Should the library's RCCL refuse to come up on this node, every rank falls back together to gathering the
        # records through host memory (gloo) and the line says so.
        res_all = [torch.zeros(world * n_rec, dtype=torch.int32, device="cuda") for _ in range(2)]
        ok = 1
        try:
            uid = [env.ArenaBatch.comm_unique_id() if rank == 0 else None]
        except env.StrikeForceError:
            uid, ok = [None], 0
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
            ok = 1 if uid[0] is not None else 0
        if ok:
            try:
                g.comm_init(uid[0], rank, world)
            except env.StrikeForceError:
                ok = 0
        if world > 1:
            t_ok = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok = int(t_ok.item())
        if not ok:
            rccl = False  # (`gather` stays on: the host path below)
    launches = [0]

    def run(first, count):
        s = first
        while s < first + count:
            k = min(args.k_per_launch, first + count - s)
            g.step_device(d_cmds.data_ptr() + s * stride, k)
            if rccl:  # sf_results_allgather: snapshot on the launch stream, ncclAllGather on the library's side stream
                g.results_allgather(res_all[launches[0] & 1].data_ptr())
                launches[0] += 1
            elif gather:  # rehearsal (ranks sharing one card), or the library's RCCL did not come up: through host memory
                res = torch.zeros(n_rec, dtype=torch.int32, device="cuda")
                g.results_device(res.data_ptr())
                torch.cuda.current_stream().synchronize()
                shard.gather_results(res.cpu(), world)
            s += k

    def drain():
        if rccl:
            g.comm_wait(host_too=True)

    run(0, pre)  # untimed pre-roll: full-length launches, populations and clocks at steady state
    run(pre, args.warmup)
    drain()
    torch.cuda.synchronize()
    g.kernel_time(True)  # start timing step launches with HIP events on the launch stream

    def timed(first):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(first, args.steps)
        drain()  # the gathers belong to the job: all of them are finished inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if world > 1:  # MAX over ranks
            tt = torch.tensor([t], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    dts = [timed(pre + args.warmup)]
    if dts[0] < SHORT_S:  # (the same decision on every rank: dts[0] is the maximum over ranks)
        total = pre + args.warmup + args.steps * REPEATS
        cmds, _ = config.bench_commands(args.arenas, cfg.n_agents, total, seed0=shard.command_seed(w, rank))  # (same stream, longer)
        d_cmds = torch.from_numpy(cmds).cuda()
        torch.cuda.synchronize()
        for r_ in range(1, REPEATS):
            dts.append(timed(pre + args.warmup + r_ * args.steps))
    dt = float(np.median(dts))
    k_ms, k_launches = g.kernel_time(False)
    rccl_ranks = None
    if rccl:
        try:
            rccl_ranks = g.comm_ranks()
        except env.StrikeForceError:
            rccl_ranks = None

    # the interactive loop an RL learner runs: one launch per step (K = 1) + the observation of every agent
    obs_n = 0 if args.no_interactive else 40
    loop_ms = obs_ms = k1_ms = host_ms = loop_delta_ms = loop_sparse_ms = loop_halves_ms = 0.0
    if obs_n:
        d_obs = torch.empty(args.arenas * cfg.n_agents * 30752, dtype=torch.float32, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        g.observe_device(d_obs.data_ptr())
        torch.cuda.synchronize()
        ev[0].record()
        for s in range(obs_n):
            g.step_device(d_cmds.data_ptr() + (s % total) * stride, 1)
            g.observe_device(d_obs.data_ptr())
        ev[1].record()
        for s in range(obs_n):
            g.observe_device(d_obs.data_ptr())
        ev[2].record()
        for s in range(obs_n):
            g.step_device(d_cmds.data_ptr() + (s % total) * stride, 1)
        ev[3].record()
        # the same loop with sf_observe_device_delta: one persistent observation buffer, only the changes are written
        g.observe_device_delta(d_obs.data_ptr())
        ev[4].record()
        for s in range(obs_n):
            g.step_device(d_cmds.data_ptr() + (s % total) * stride, 1)
            g.observe_device_delta(d_obs.data_ptr())
        ev[5].record()
        # the same loop with the observation as the list of its non-zero floats (what sf_policy_forward_sparse consumes)
        n_ag = args.arenas * cfg.n_agents
        s_keys = torch.zeros((n_ag, 2048), dtype=torch.int32, device="cuda")
        s_vals = torch.zeros((n_ag, 2048), dtype=torch.float32, device="cuda")
        s_cnt = torch.zeros(n_ag, dtype=torch.int32, device="cuda")
        s_pov = torch.zeros((n_ag, 160), dtype=torch.float32, device="cuda")
        g.observe_sparse_device(s_keys.data_ptr(), s_vals.data_ptr(), s_cnt.data_ptr(), s_pov.data_ptr(), 2048)
        ev[6].record()
        for s in range(obs_n):
            g.step_device(d_cmds.data_ptr() + (s % total) * stride, 1)
            g.observe_sparse_device(s_keys.data_ptr(), s_vals.data_ptr(), s_cnt.data_ptr(), s_pov.data_ptr(), 2048)
        ev[7].record()
        torch.cuda.synchronize()
        loop_sparse_ms = ev[6].elapsed_time(ev[7]) / obs_n
        del s_keys, s_vals, s_cnt, s_pov
        loop_ms = ev[0].elapsed_time(ev[1]) / obs_n
        obs_ms = ev[1].elapsed_time(ev[2]) / obs_n
        k1_ms = ev[2].elapsed_time(ev[3]) / obs_n
        loop_delta_ms = ev[4].elapsed_time(ev[5]) / obs_n
        # PCIe-inclusive: sf_step() with a host command array (H2D copy of arenas*agents bytes + K=1 launch)
        t1 = time.perf_counter()
        for s in range(obs_n):
            g.step(cmds[s % total])
        g.synchronize()
        host_ms = (time.perf_counter() - t1) * 1e3 / obs_n
        # The same list-observation loop driven as TWO half-batches on two streams: a one-step launch ends with the
        # youngest waves of its SIMDs (tools/experiments/README.md, the timeline of a one-step launch); the other half's
        # kernels fill that tail.  Same arenas, same seeds, same commands as the one batch, brought to the same point
        # of their games first.
        halves = []
        if args.arenas % 2 == 0:
            done_steps = pre + args.warmup + args.steps * len(dts)
            for j in range(2):
                n = args.arenas // 2
                wj = config.baseline_workload(args.workload, arenas=n, device=local)
                wj.cfg.reseed_stride = world * args.arenas
                gj = env.ArenaBatch(wj)
                sj = torch.cuda.Stream()
                gj.set_stream(sj.cuda_stream)
                gj.reset(*wj.seeds(first_arena=rank * args.arenas + j * n))
                cj, _ = config.bench_commands(n, cfg.n_agents, done_steps + 2 * obs_n, seed0=shard.command_seed(w, rank) + j * n * cfg.n_agents)
                with torch.cuda.stream(sj):
                    dj = torch.from_numpy(cj).cuda()
                    bj = (torch.zeros((n * cfg.n_agents, 2048), dtype=torch.int32, device="cuda"),
                          torch.zeros((n * cfg.n_agents, 2048), dtype=torch.float32, device="cuda"),
                          torch.zeros(n * cfg.n_agents, dtype=torch.int32, device="cuda"), torch.zeros((n * cfg.n_agents, 160), device="cuda"))
                sj.synchronize()
                for s0 in range(0, done_steps, args.k_per_launch):
                    gj.step_device(dj.data_ptr() + s0 * n * cfg.n_agents, min(args.k_per_launch, done_steps - s0))
                halves.append((gj, n * cfg.n_agents, dj, bj))

            def half_loop(first, count):
                for s in range(first, first + count):
                    for gj, sn, dj, bj in halves:
                        gj.step_device(dj.data_ptr() + s * sn, 1)
                        gj.observe_sparse_device(bj[0].data_ptr(), bj[1].data_ptr(), bj[2].data_ptr(), bj[3].data_ptr(), 2048)
            half_loop(done_steps, 5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            half_loop(done_steps + 5, obs_n)
            torch.cuda.synchronize()
            loop_halves_ms = (time.perf_counter() - t1) * 1e3 / obs_n
            for gj, _sn, _dj, _bj in halves:
                gj.close()
            del halves
    g.kernel_time(False)

    # the closed loop with the reference's bot network evaluated on the device:
    # observe -> sf_policy_forward -> sf_policy_act -> K=1 step, nothing leaves HBM
    pol = None
    if obs_n and not args.no_policy:
        from strikeforce_amd import policy
        agents = args.arenas * cfg.n_agents
        pb = policy.PolicyBatch(policy.init_parameters(seed=0), agents, device=local)
        pb.set_stream(torch.cuda.current_stream().cuda_stream)
        d_probs = torch.empty(agents * 9, dtype=torch.float32, device="cuda")
        d_value = torch.empty(agents, dtype=torch.float32, device="cuda")
        d_pcmd = torch.zeros(agents, dtype=torch.uint8, device="cuda")
        d_new = torch.zeros(agents, dtype=torch.uint8, device="cuda")
        pol_n = 10

        SP_CAP = 2048  # list entries per agent (an observation of the BASELINE configurations has ~250 non-zero floats)
        d_keys = torch.zeros((agents, SP_CAP), dtype=torch.int32, device="cuda")
        d_vals = torch.zeros((agents, SP_CAP), dtype=torch.float32, device="cuda")
        d_counts = torch.zeros(agents, dtype=torch.int32, device="cuda")
        d_pov = torch.zeros((agents, 160), dtype=torch.float32, device="cuda")
        # dense rows for the agents whose list does not fit (none in this workload; the two fallback launches then exit
        # at once): nobody is ever evaluated on a blank window, and nothing synchronises with the host
        d_fallback = torch.empty((agents, 30752), dtype=torch.float32, device="cuda")

        done_view = g.done_view_device()

        def closed_loop(n, one_call=True):
            for _ in range(n):
                # the observation goes to the network as the list of its non-zeros (bit-identical results to the dense
                # sf_observe_device + sf_policy_forward pair, tests/test_gpu_sparse_obs.py)
                g.observe_sparse_device(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), SP_CAP)
                g.observe_overflow_device(d_counts.data_ptr(), SP_CAP, d_fallback.data_ptr(), d_pov.data_ptr())
                if one_call:
                    # Agent::predict + update as the one call they are in the reference: the fresh memory of the agents
                    # whose game restarted (read from the environment's own flags), the forward and the draw in the
                    # forward's two launches — same results as the calls below, tests/test_gpu_sparse_obs.py
                    pb.predict_sparse(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), SP_CAP, agents,
                                      d_probs.data_ptr(), d_value.data_ptr(), d_pcmd.data_ptr(), seed=rank,
                                      d_dense_ptr=d_fallback.data_ptr(), reset_words=done_view)
                    g.step_device(d_pcmd.data_ptr(), 1)
                    continue
                pb.forward_sparse(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), SP_CAP, agents,
                                  d_probs.data_ptr(), d_value.data_ptr(), d_dense_ptr=d_fallback.data_ptr())
                pb.act(d_probs.data_ptr(), agents, d_pcmd.data_ptr(), seed=rank)
                g.step_device(d_pcmd.data_ptr(), 1)
                g.done_device(d_new.data_ptr())      # agents whose game restarted get a fresh memory,
                pb.reset_memory(d_new.data_ptr())    # like the reference's new Agent per game

        closed_loop(2, False)
        torch.cuda.synchronize()
        pe3 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        pe3[0].record()
        closed_loop(pol_n, False)
        pe3[1].record()
        closed_loop(2)
        torch.cuda.synchronize()
        # the loop's rate first, with nothing but the two events around it; then once more with the library's per-launch
        # events on (two hipEventRecord per matrix launch: ~8 % of a loop of this length) for the per-kernel rows
        pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        pe[0].record()
        closed_loop(pol_n)
        pe[1].record()
        torch.cuda.synchronize()
        pb.kernel_time(True)
        closed_loop(pol_n)
        torch.cuda.synchronize()
        by_k = pb.kernel_time_by_kernel(False)  # k_gemm (f32), k_gemm_b3 (bf16 split), conv0 on the non-zeros, k_tail
        sp_over = pb.sparse_overflows()
        n_cnt = d_counts.cpu().numpy().view(np.uint32)
        sp_fallback = int(((n_cnt > SP_CAP) | (n_cnt == 0xFFFFFFFF)).sum())  # agents redone from the dense fallback (last step)
        # useful work of conv0 on the last step's lists: every non-zero reaches the output pixels whose 3x3 / stride-2
        # window holds it (1 per axis for an odd coordinate, 2 for an even inner one), 160 channels, one fma each
        kk = d_keys.cpu().numpy().view(np.uint32)
        ok = np.arange(SP_CAP, dtype=np.uint32)[None, :] < np.minimum(n_cnt, SP_CAP)[:, None]
        yy, xx = (kk >> 9) & 31, (kk >> 14) & 31
        nwin = lambda v: np.where(v & 1, 1, np.where((v > 0) & (v < 30), 2, 1))
        conv0_fma = float((nwin(yy) * nwin(xx) * ok).sum()) * 160.0
        nonzeros = float(np.minimum(n_cnt, SP_CAP).sum())
        # the same forward on a second parameter set: weights 4x the default initialisation (saturating gates, peaked
        # softmax — the shape a trained checkpoint has, which a fresh initialisation lacks); operands change clocks —
        # and on the layered form of the convolution stack (SF_POLICY_LAYERED=1: conv0 on the non-zeros, conv1 / conv2 as
        # bf16-split products, conv3 in k_tail), the cross-check path, for the comparison
        pb4 = policy.PolicyBatch(policy.init_parameters(seed=3, gain=4.0), agents, device=local)
        pb4.set_stream(torch.cuda.current_stream().cuda_stream)
        os.environ["SF_POLICY_LAYERED"] = "1"
        try:
            pbl = policy.PolicyBatch(policy.init_parameters(seed=0), agents, device=local)
        finally:
            del os.environ["SF_POLICY_LAYERED"]
        pbl.set_stream(torch.cuda.current_stream().cuda_stream)
        ev4 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        fwd = lambda q: q.forward_sparse(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), SP_CAP, agents,
                                         d_probs.data_ptr(), d_value.data_ptr(), d_dense_ptr=d_fallback.data_ptr())
        for q in (pb, pb4, pbl):
            fwd(q), fwd(q)
        ev4[0].record()
        for _ in range(5):
            fwd(pb)
        ev4[1].record()
        for _ in range(5):
            fwd(pb4)
        ev4[2].record()
        for _ in range(5):
            fwd(pbl)
        ev4[3].record()
        torch.cuda.synchronize()
        pb4.close()
        pbl.close()
        pol = {"sparse_overflows": sp_over, "dense_fallback_agents_last_step": sp_fallback, "loop_ms": pe[0].elapsed_time(pe[1]) / pol_n,
               "loop_ms_separate_calls": pe3[0].elapsed_time(pe3[1]) / pol_n,
               "by_kernel": [(m / pol_n, f / pol_n, n // pol_n) for (m, f, n) in by_k], "conv0_fma": conv0_fma, "nonzeros": nonzeros,
               "forward_ms_default_init": ev4[0].elapsed_time(ev4[1]) / 5, "forward_ms_gain4": ev4[1].elapsed_time(ev4[2]) / 5,
               "forward_ms_layered": ev4[2].elapsed_time(ev4[3]) / 5,
               "steps": pol_n, "agents": agents}
        pb.close()

    if rank == 0:
        env_steps = world * args.arenas * args.steps
        bytes_step = algorithmic_bytes_per_step(cfg)
        avg_launch_s = (k_ms / 1e3) / max(1, k_launches)
        steps_per_launch = args.steps * len(dts) / max(1, k_launches)
        achieved = bytes_step * args.arenas * steps_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        kpl_actual = int(round(steps_per_launch))
        traffic = pmc_traffic(args.workload, args.arenas, kpl_actual) if abs(steps_per_launch - kpl_actual) < 1e-9 else None
        out = {
            "metric": "env-steps/sec (whole node), 64x64 map x32 entities",
            "value": env_steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "repeats": {"n": len(dts), "min_ms_per_step": min(dts) * 1e3 / args.steps, "max_ms_per_step": max(dts) * 1e3 / args.steps,
                        "what": "the timed region of exactly --steps steps, repeated on fresh command windows when it is "
                                "shorter than %d ms; value / ms_per_step are the median" % int(SHORT_S * 1e3)},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": describe_workload(args.workload, args.arenas, cfg),
                       "arenas_per_gpu": args.arenas, "steps_per_launch": steps_per_launch, "preroll_steps": pre,
                       "parallelism": ("arena-sharded x%d, no data-path collective; result records all-gathered over "
                                       "RCCL (%s) after each launch"
                                       % (world, "sf_results_allgather: the library's own communicator, the only one of the job"
                                          if rccl else "host path over gloo: the library's RCCL did not come up, or a rehearsal"))
                                      if world > 1 else
                                      "one GPU, arena-sharded by construction (no data-path collective)"},
            # contract figure: ALGORITHMIC bytes per launch / measured launch time against HBM peak.  `traffic` is the
            # HBM bytes the counters saw for a launch of this length (None when that shape was not profiled): the arena
            # state stays in registers/LDS across the K steps of a launch, so real HBM use is a few percent of the
            # algorithmic figure and the kernel's actual limiter is instruction issue (scalar + vector), see `limiter`.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_hbm_gbs": (traffic / avg_launch_s / 1e9) if (traffic and avg_launch_s > 0) else None,
                         "kernel": "k_step", "launches": k_launches, "avg_launch_ms": avg_launch_s * 1e3,
                         "steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_arena_step": bytes_step,
                         "limiter": "instruction issue of the slot-ordered (scalar) game loops, not HBM: see "
                                    "profiles/ (instruction mix, phase stamps) and DESIGN.md §6",
                         "issue": issue_bound(args.workload, args.arenas)},
        }
        if world > 1:
            # ncclCommCount of the library's communicator (sf_comm_ranks): what RCCL itself says took part; None on the
            # host path.  The multi-rank sf_results_allgather path is unmeasured on hardware until a SCALE record exists.
            out["rccl_ranks"] = rccl_ranks
            out["multi_rank_gather"] = "unmeasured on hardware before this run (no SCALE record of an earlier round exists)"
        obs_bytes = args.arenas * cfg.n_agents * (30752 * 4 + 961 * 8)
        if obs_n:
            out["interactive"] = {
              "what": "per rank: K=1 launch per step + sf_observe_device for every agent, %d steps" % obs_n,
              "env_steps_per_s": world * args.arenas / (loop_ms / 1e3),
              "ms_per_step": loop_ms, "k_step_K1_ms": k1_ms, "k_observe_ms": obs_ms,
              "delta_observation": {"what": "same loop with sf_observe_device_delta (one persistent buffer, only changed "
                                            "floats written; buffer bit-identical to the plain call's)",
                                    "env_steps_per_s": world * args.arenas / (loop_delta_ms / 1e3),
                                    "ms_per_step": loop_delta_ms},
              "sparse_observation": {"what": "same loop with sf_observe_sparse_device (the non-zero floats as a list, the form "
                                             "sf_policy_forward_sparse takes; no dense buffer)",
                                     "env_steps_per_s": world * args.arenas / (loop_sparse_ms / 1e3),
                                     "ms_per_step": loop_sparse_ms,
                                     "two_half_batches": None if not loop_halves_ms else {
                                         "what": "the same arenas as two half-batches (two sf_env, two streams): each half's one-step "
                                                 "launches fill the other's tails; wall clock over %d steps" % obs_n,
                                         "env_steps_per_s": world * args.arenas / (loop_halves_ms / 1e3), "ms_per_step": loop_halves_ms}},
            "sf_step_host_cmd_ms": host_ms,
              "k_observe_roofline": {"bound": "hbm", "achieved": obs_bytes / (obs_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": obs_bytes / (obs_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_agent_step": 30752 * 4 + 961 * 8},
          }
        if pol:
            (g_ms, g_fl, g_n), (b_ms, b_fl, b_n), (c_ms, _c_fl, c_n), (t_ms, t_fl, t_n) = pol["by_kernel"]
            tf = lambda fl, ms: fl / (ms / 1e3) / 1e12 if ms > 0 else 0.0
            nz_agent = pol["nonzeros"] / max(1, pol["agents"])
            # k_feat_list: per non-zero one 640-byte row of the composed matrix F + the 8-byte list entry; per agent the
            # 640-byte result.  F is 21 MB: the rows come out of the XCDs' L2s (88 % of the requests hit,
            # profiles/r03e_policy_l2_counters.json) and the infinity cache, not HBM, so the ceiling priced here is the
            # guide's measured row-gather rate out of L2 (MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide;
            # 8.6 TB/s out of the infinity cache); `vs_hbm_peak` is the same bytes against the HBM peak
            feat_bytes = pol["nonzeros"] * (640 + 8) + pol["agents"] * 640
            feat_gbs = feat_bytes / (c_ms / 1e3) / 1e9 if c_ms > 0 else 0.0
            feat_row = {"ms_per_forward": c_ms, "bound": "l2-gather", "achieved": feat_gbs, "peak": L2_GATHER_GBS, "unit": "GB/s",
                        "frac": feat_gbs / L2_GATHER_GBS, "vs_hbm_peak": feat_gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": feat_bytes, "useful_gflop": 2.0 * pol["nonzeros"] * 160 / 1e9,
                        "what": "the four bias-free convolutions (Modules.hpp:66-71: nothing between them) as ONE matrix, composed at "
                                "sf_policy_create, applied to the %.0f non-zeros per agent: one 640-byte row of it per non-zero "
                                "(f32 fmaf chain per pair of channels, the partial sums combined in f64); the matrix is 21 MB and "
                                "its rows come out of the L2s / infinity cache" % nz_agent}
            tail_row = {"ms_per_forward": t_ms, "bound": "mfma", "achieved": tf(t_fl, t_ms), "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": tf(t_fl, t_ms) / MFMA_F32_PEAK_TFLOPS,
                        "what": "both GRU cells + combined_processor + 6 ResB layers + heads, f32 MFMA"}
            # the forward's dominant kernel; the object keeps the contract's vocabulary ("mfma" for k_tail)
            dom = dict(feat_row if c_ms >= t_ms else tail_row)
            dom["kernel"] = "k_feat_list" if c_ms >= t_ms else "k_tail"
            dom.pop("what", None)
            out["policy"] = {
                "what": "per rank: observe (as the list of non-zero floats) -> bot-0.5 network (f32 results, random-init weights; "
                        "the convolution stack — four bias-free convolutions with nothing between them — as one composed matrix on "
                        "those non-zeros, everything behind it in one kernel on the f32 MFMA) -> sample -> K=1 step -> memory reset "
                        "of restarted games, all on device, %d steps; sf_policy_predict_sparse: reset + forward + sample in the "
                        "forward's two launches — six launches per step, two of them the fallbacks that are idle when every list fits "
                        "(separate_calls: the same loop through sf_policy_forward_sparse_or_dense / sf_policy_act / sf_done_device / "
                        "sf_policy_reset_memory, nine); "
                        "agents evaluated on a blank window: %d (lists that do not fit "
                        "are redone from a dense fallback on the device; %d agents took it in the last step)"
                        % (pol["steps"], pol["sparse_overflows"], pol["dense_fallback_agents_last_step"]),
                "agent_steps_per_s": world * pol["agents"] / (pol["loop_ms"] / 1e3), "ms_per_step": pol["loop_ms"],
                "separate_calls": {"agent_steps_per_s": world * pol["agents"] / (pol["loop_ms_separate_calls"] / 1e3),
                                   "ms_per_step": pol["loop_ms_separate_calls"]},
                "forward_ms": {"default_init": pol["forward_ms_default_init"], "weights_x4": pol["forward_ms_gain4"],
                               "layered": pol["forward_ms_layered"],
                               "what": "the forward alone (2 kernels) on the last step's lists: libtorch's default initialisation, "
                                       "weights 4x that (saturated gates: a trained checkpoint's operand shape), and the layered "
                                       "cross-check form of the convolution stack (SF_POLICY_LAYERED=1: 48.6 MFLOP per agent, "
                                       "conv1 / conv2 as bf16-split products on the bf16 MFMA — round 2's path)"},
                "roofline": dom,
                "kernels": {"k_feat_list": feat_row, "k_tail": tail_row},
                "flop_per_agent_forward": 2.0 * nz_agent * 160 + (g_fl + b_fl + t_fl) / pol["agents"],
            }
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(args, local, torch, config, env, args.workload)
        if not args.no_cpu_baseline:  # rank 0, after the timed region, at any N (the reference itself on one host core)
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
