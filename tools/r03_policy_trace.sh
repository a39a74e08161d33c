#!/bin/bash
# Kernel trace of the closed loop observe -> network -> sample -> step (bench.py's policy section, configs[2], 4096 agents):
# per-kernel durations of one loop iteration.  gpurun: bash tools/r03_policy_trace.sh [tag]
TAG=${1:-r03e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG
rm -rf gpurun_out/prof_pol && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pol -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/policy_bench_under_rocprof.json 2> $O/policy_rocprof.err
cp gpurun_out/prof_pol/*/*_kernel_stats.csv $O/policy_loop_kernel_stats.csv
cut -c1-150 $O/policy_loop_kernel_stats.csv | head -24
