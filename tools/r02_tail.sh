#!/bin/bash
set -e
mkdir -p gpurun_out/tail
timeout -k 10 900 python -m pytest tests/test_gpu_policy.py -x -q > gpurun_out/tail/tests.log 2>&1 || { tail -40 gpurun_out/tail/tests.log; exit 1; }
tail -1 gpurun_out/tail/tests.log
for r in 1 2; do
SF_POLICY_FUSED_TAIL=0 timeout -k 10 200 python tools/policy_bench.py 4096 30 | tr '\n' ' '; echo " <- separate"
timeout -k 10 200 python tools/policy_bench.py 4096 30 | tr '\n' ' '; echo " <- fused"
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tail/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail/prof -- python3 tools/policy_bench.py 4096 20 > gpurun_out/tail/prof.log 2>&1
grep -h "k_tail\|b3\|conv0" gpurun_out/tail/prof/*/*_kernel_stats.csv | cut -c1-120
