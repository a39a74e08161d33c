#!/bin/bash
set -e
mkdir -p gpurun_out/rng
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "parity or invariants or digest or lockstep" > gpurun_out/rng/tests.log 2>&1 || { tail -30 gpurun_out/rng/tests.log; exit 1; }
tail -1 gpurun_out/rng/tests.log
WL="C3 C2" bash tools/ab.sh tools/ab/libsf_head.so strikeforce_amd/libstrikeforce_amd.so | tee gpurun_out/rng/ab.txt
