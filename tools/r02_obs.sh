#!/bin/bash
# observation kernel: parity tests, then its time inside the interactive and policy loops
set -e
mkdir -p gpurun_out/obs
timeout -k 10 600 python -m pytest tests/test_gpu_sparse_obs.py tests/test_gpu_parity.py tests/test_describe_kat.py tests/test_gpu_policy.py -x -q -m gpu > gpurun_out/obs/tests.log 2>&1 || { tail -30 gpurun_out/obs/tests.log; exit 1; }
tail -1 gpurun_out/obs/tests.log
timeout -k 10 300 python tools/policy_bench.py 4096 30 | tail -1 | cut -c1-300
timeout -k 10 600 python bench.py --no-cpu-baseline --no-other-configs --no-policy > gpurun_out/obs/bench.json 2> gpurun_out/obs/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/obs/bench.json")); i = d["interactive"]
print("interactive", round(i["env_steps_per_s"] / 1e6, 1), "M; k_observe", round(i["k_observe_ms"], 4), "ms; delta", round(i["delta_observation"]["env_steps_per_s"] / 1e6, 1),
      "M; sparse", round(i["sparse_observation"]["env_steps_per_s"] / 1e6, 1), "M; k_step K1", round(i["k_step_K1_ms"], 4))
PY
