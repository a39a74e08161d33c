"""TEST INFRASTRUCTURE (uses the oracle / the reference build as the checker).  Soak: the HIP network (both forms of its convolution stack) against the REFERENCE's own AgentModel
(oracle/_ref/libsf_refmodules.so: bots/bot-0.5/Modules.hpp:26-180 compiled unedited, one model object per agent as the
reference has one Agent per human) on the GPU simulator's own observations, over a recurrent closed loop: every step the
reference models and the HIP batch see the same observation and are fed the same (arg-max) action.  Prints one JSON line per
form with the worst differences.  Usage (GPU box; the .so travels with the snapshot): python tests/tools/soak_policy.py [agents] [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import refmodules  # noqa: E402
from strikeforce_amd import config, env, policy  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 48
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
assert refmodules.lib() is not None, "oracle/_ref/libsf_refmodules.so is not here"
for form in ("folded", "layered"):
    if form == "layered":
        os.environ["SF_POLICY_LAYERED"] = "1"
    else:
        os.environ.pop("SF_POLICY_LAYERED", None)
    params = policy.init_parameters(seed=17, gain=2.0)
    w = config.baseline_workload("C3", arenas=A)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    pb = policy.PolicyBatch(params, A)
    refs = [refmodules.RefAgentModel(params) for _ in range(A)]
    d_obs = torch.zeros((A, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_probs = torch.zeros((A, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(A, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(A, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(A, dtype=torch.int32, device="cuda")
    worst_p = worst_v = worst_h = 0.0
    agree = total = 0
    for s in range(STEPS):
        g.observe_device(d_obs.data_ptr())
        pb.forward(d_obs.data_ptr(), A, d_probs.data_ptr(), d_value.data_ptr())
        pb.act(d_probs.data_ptr(), A, d_cmd.data_ptr(), greedy=True, d_action_ptr=d_act.data_ptr())
        pb.synchronize()
        obs = d_obs.cpu().numpy()
        probs, value, acts = d_probs.cpu().numpy(), d_value.cpu().numpy(), d_act.cpu().numpy()
        for b in range(A):
            rp, rv, rh = refs[b].forward(obs[b])
            worst_p = max(worst_p, float(np.max(np.abs(probs[b] - rp) / (1e-6 + 5e-5 * np.abs(rp)))))
            worst_v = max(worst_v, float(abs(value[b] - rv) / (1e-6 + 5e-5 * abs(rv))))
            hb, _ = pb.get_memory(b)
            worst_h = max(worst_h, float(np.max(np.abs(hb - rh) / (1e-5 + 5e-5 * np.abs(rh)))))
            refs[b].update_actions(int(acts[b]))  # the same action into the reference's memory as into ours
            total += 1
        g.step_device(d_cmd.data_ptr(), 1)
        done = g.done()
        for b in np.nonzero(np.asarray(done).reshape(-1))[0]:
            refs[b].reset_memory()
        d_new = torch.from_numpy(np.asarray(done, dtype=np.uint8).reshape(-1)).cuda()
        pb.reset_memory(d_new.data_ptr())
    print(json.dumps({"form": form, "agents": A, "recurrent_steps": STEPS, "forwards_compared": total,
                      "worst_error_in_units_of_the_gate": {"probabilities": round(worst_p, 4), "value": round(worst_v, 4),
                                                            "recurrent_state": round(worst_h, 4)},
                      "gate": "|hip - reference| <= 1e-6 + 5e-5 |reference| (state: 1e-5 + 5e-5 |reference|); 1.0 = at the gate"}),
          flush=True)
    assert max(worst_p, worst_v, worst_h) <= 1.0
    for r in refs:
        r.close()
    pb.close()
    g.close()
