#!/bin/bash
# Kernel trace of the policy network (run through gpurun): per-kernel durations of sf_policy_forward at 4096 agents.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_policy && mkdir -p gpurun_out/prof_policy
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_policy/stats -- python3 tools/policy_bench.py 4096 10 > gpurun_out/prof_policy/bench.log 2>&1
cat gpurun_out/prof_policy/bench.log | tail -3
cat gpurun_out/prof_policy/stats/*/*_kernel_stats.csv | cut -c1-200
