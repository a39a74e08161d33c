"""One line per bench JSON: headline, roofline fraction, interactive and policy figures."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    line = "%s value %.1f M frac %.3f" % (f, d["value"] / 1e6, d["roofline"]["frac"])
    if "interactive" in d:
        i = d["interactive"]
        hb = i["sparse_observation"].get("two_half_batches")
        line += " | interactive dense %.1f delta %.1f list %.1f M (two halves %s M; K=1 step %.4f ms, k_observe %.4f ms)" % (
            i["env_steps_per_s"] / 1e6, i["delta_observation"]["env_steps_per_s"] / 1e6, i["sparse_observation"]["env_steps_per_s"] / 1e6,
            ("%.1f" % (hb["env_steps_per_s"] / 1e6)) if hb else "-",
            i["k_step_K1_ms"], i["k_observe_ms"])
    if "policy" in d:
        p = d["policy"]
        line += " | policy %.2f M agent-steps/s (%.4f ms; forward %.4f; k_tail %.4f)" % (
            p["agent_steps_per_s"] / 1e6, p["ms_per_step"], p["forward_ms"]["default_init"], p["kernels"]["k_tail"]["ms_per_forward"])
        if "separate_calls" in p:
            line += " separate calls %.2f M" % (p["separate_calls"]["agent_steps_per_s"] / 1e6)
    print(line)
