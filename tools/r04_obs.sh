#!/bin/bash
# round 4: the wave-per-agent list observation (k_observe_list): parity tests, then the bench's interactive / policy
# sections with the new kernel and with round 3's form (SF_OBS_LIST_BLOCK=1) in the same call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04d
timeout -k 10 600 python -m pytest tests/test_gpu_sparse_obs.py tests/test_gpu_policy.py -x -q -m gpu > gpurun_out/r04d/pytest_obs.log 2>&1
rc=$?
tail -15 gpurun_out/r04d/pytest_obs.log
[ $rc -eq 0 ] || exit $rc
for form in new old; do
  if [ $form = old ]; then export SF_OBS_LIST_BLOCK=1; else unset SF_OBS_LIST_BLOCK; fi
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r04d/bench_$form.json 2> gpurun_out/r04d/bench_$form.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r04d/bench_$form.json"))
i, p = d["interactive"], d["policy"]
so = i["sparse_observation"]
print("$form", "value", round(d["value"] / 1e6, 1), "| interactive dense", round(i["env_steps_per_s"] / 1e6, 1), "list, one launch", round(so["env_steps_per_s"] / 1e6, 1),
      "ms", round(so["ms_per_step"], 4), "two launches", round(so["two_launches"]["env_steps_per_s"] / 1e6, 1), "| policy", round(p["agent_steps_per_s"] / 1e6, 2), "ms", round(p["ms_per_step"], 4))
PY
done
