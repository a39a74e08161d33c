#!/bin/bash
# round 4, first GPU pass: the GPU parity suite on the library with the large-pool (ZL) kernels, then a same-call A/B of
# the throughput kernels against round 3's library (tools/ab/r03_base.so), then a whole native game timed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04a
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04a/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/r04a/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
WL="C3 C2" tools/ab.sh tools/ab/r03_base.so strikeforce_amd/libstrikeforce_amd.so 2>&1 | tee gpurun_out/r04a/ab.txt
