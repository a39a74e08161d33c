#!/bin/bash
# end-of-round check on one box: GPU tests, smoke(), the bench in its default and in the driver's form, N=2 on one card
set -e
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/final/pytest_gpu.log; exit 1; }
tail -1 gpurun_out/final/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_form.json 2> gpurun_out/final/bench_driver_form.err
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_form"):
    d = json.load(open("gpurun_out/final/%s.json" % f))
    print(f, round(d["value"] / 1e6, 1), "M env-steps/s, frac", round(d["roofline"]["frac"], 3), "| policy", round(d["policy"]["agent_steps_per_s"] / 1e6, 2) if "policy" in d else None,
          "| interactive", round(d["interactive"]["env_steps_per_s"] / 1e6, 1) if "interactive" in d else None)
PY
