"""Times sf_policy_forward at a given batch (default 4096 agents) on real observations of workload C2.
Usage: python tools/policy_bench.py [agents] [iters]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from strikeforce_amd import config, env, policy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = config.baseline_workload("C2", arenas=B)
g = env.ArenaBatch(w)
g.reset(*w.seeds())
pb = policy.PolicyBatch(policy.init_parameters(0), B)
d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
g.observe_device(d_obs.data_ptr())
g.synchronize()
for _ in range(2):
    pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
pb.synchronize()
pb.kernel_time(True)
t0 = time.perf_counter()
for _ in range(iters):
    pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
pb.synchronize()
wall = (time.perf_counter() - t0) / iters
ms, flop, n = pb.kernel_time(False)
# (the default form of the network — the convolution stack folded into one matrix — has one matrix launch, k_tail;
# SF_POLICY_LAYERED=1 times the layer-by-layer form with its GEMM launches)
print(json.dumps({"agents": B, "forward_ms": wall * 1e3, "gemm_ms_per_forward": ms / iters, "gemm_launches": n // iters,
                  "gemm_tflops": flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "flop_per_agent": flop / iters / B,
                  "agent_forwards_per_s": B / wall}))
# the closed loop: observe -> forward -> act -> step
t0 = time.perf_counter()
for _ in range(iters):
    g.observe_device(d_obs.data_ptr())
    pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
    pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=1)
    g.step_device(d_cmd.data_ptr(), 1)
g.synchronize()
pb.synchronize()
wall = (time.perf_counter() - t0) / iters
print(json.dumps({"closed_loop_ms_per_step": wall * 1e3, "closed_loop_agent_steps_per_s": B / wall}))
# the same with the observation handed over as the list of its non-zeros (no dense buffer written or read)
CAP = 2048
keys = torch.zeros((B, CAP), dtype=torch.int32, device="cuda")
vals = torch.zeros((B, CAP), dtype=torch.float32, device="cuda")
counts = torch.zeros(B, dtype=torch.int32, device="cuda")
pov = torch.zeros((B, 160), dtype=torch.float32, device="cuda")
for _ in range(2):
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, d_probs.data_ptr(), d_value.data_ptr())
pb.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, d_probs.data_ptr(), d_value.data_ptr())
pb.synchronize()
fwd = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
for _ in range(iters):
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, d_probs.data_ptr(), d_value.data_ptr())
    pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=1)
    g.step_device(d_cmd.data_ptr(), 1)
g.synchronize()
pb.synchronize()
wall = (time.perf_counter() - t0) / iters
print(json.dumps({"sparse_forward_ms": fwd * 1e3, "sparse_closed_loop_ms_per_step": wall * 1e3,
                  "sparse_closed_loop_agent_steps_per_s": B / wall, "overflows": pb.sparse_overflows()}))
