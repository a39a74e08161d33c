#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04b
WL="C3 C2" tools/ab.sh tools/ab/r03_base.so strikeforce_amd/libstrikeforce_amd.so tools/ab/libsf_short4.so tools/ab/libsf_freewarm.so tools/ab/libsf_short4free.so 2>&1 | tee gpurun_out/r04b/ab_round_experiments.txt
