"""Closed loop on one MI355X with nothing leaving HBM: observe -> bot-0.5 network -> sample -> step.

    python examples/policy_loop.py [arenas] [steps]

The network's parameters are random here (same shapes and initialisers as the reference's AgentModel); pass a dict
of numpy arrays keyed by the reference's parameter names to `PolicyBatch` to run a trained model.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from strikeforce_amd import config, env, policy

arenas = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = config.baseline_workload("C2", arenas=arenas)           # 64x64 map, 1 player + 16 zombies, auto-reset
sim = env.ArenaBatch(w)
agents = arenas * w.cfg.n_agents
net = policy.PolicyBatch(policy.init_parameters(seed=0), agents)
# a stream of its own for both libraries: launches on the NULL stream pay its implicit ordering against every other stream
# (~10 % on a loop of one launch per step)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
sim.set_stream(stream.cuda_stream), net.set_stream(stream.cuda_stream)
sim.reset(*w.seeds())
CAP = 2048                                                   # list entries per agent (an observation has ~250 non-zero floats)
d_keys = torch.zeros((agents, CAP), dtype=torch.int32, device="cuda")
d_vals = torch.zeros((agents, CAP), dtype=torch.float32, device="cuda")
d_counts = torch.zeros(agents, dtype=torch.int32, device="cuda")
d_pov = torch.zeros((agents, 160), dtype=torch.float32, device="cuda")
d_dense = torch.empty((agents, 32, 31, 31), dtype=torch.float32, device="cuda")  # rows only for lists that do not fit
d_probs = torch.empty((agents, 9), dtype=torch.float32, device="cuda")
d_value = torch.empty(agents, dtype=torch.float32, device="cuda")
d_cmd = torch.zeros(agents, dtype=torch.uint8, device="cuda")
restarted = sim.done_view_device()                           # the games that just restarted, where the library keeps the flags
t0 = time.perf_counter()
for t in range(steps):
    # gameplay::bot()'s encoding as the list of its non-zero floats (the dense form: observe_device + net.forward)
    sim.observe_sparse_device(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), CAP)
    sim.observe_overflow_device(d_counts.data_ptr(), CAP, d_dense.data_ptr(), d_pov.data_ptr())
    # Agent::predict + update in one call: a new Agent's memory for restarted games, AgentModel::forward, the draw
    # (the same as net.reset_memory / net.forward_sparse / net.act one after the other)
    net.predict_sparse(d_keys.data_ptr(), d_vals.data_ptr(), d_counts.data_ptr(), d_pov.data_ptr(), CAP, agents, d_probs.data_ptr(),
                       d_value.data_ptr(), d_cmd.data_ptr(), seed=1234, d_dense_ptr=d_dense.data_ptr(), reset_words=restarted)
    sim.step_device(d_cmd.data_ptr(), 1)                     # one tick of every arena
sim.synchronize()
net.synchronize()
dt = time.perf_counter() - t0
res = sim.results()
print("%d arenas x %d steps in %.2f s = %.2f M agent-steps/s; kills so far: %d; mean state value %.3f"
      % (arenas, steps, dt, agents * steps / dt / 1e6, int(res[:, :, 0].sum()), float(d_value.mean())))
