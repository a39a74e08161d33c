#!/bin/bash
set -e
mkdir -p gpurun_out/sparse
timeout -k 10 600 python -m pytest tests/test_gpu_sparse_obs.py tests/test_gpu_policy.py -x -q > gpurun_out/sparse/tests.log 2>&1 || { tail -40 gpurun_out/sparse/tests.log; exit 1; }
tail -1 gpurun_out/sparse/tests.log
