#!/bin/bash
# round 4: (1) the bound of any restructuring of zombie_action's per-zombie decision logic (-DSF_EXP_ZLOOP_FREE: the draw
# loop keeps its draws and loses everything else), same-call A/B; (2) the GPU parity suite incl. the C++ host programs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04c
WL="C3 C2" tools/ab.sh strikeforce_amd/libstrikeforce_amd.so tools/ab/libsf_zloopfree.so 2>&1 | grep -v "^  \|Traceback\|json" | tee gpurun_out/r04c/ab_zloop.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04c/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/r04c/pytest_gpu.log
exit $rc
