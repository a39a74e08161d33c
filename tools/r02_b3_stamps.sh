#!/bin/bash
mkdir -p gpurun_out/b3
for d in zero rand; do
  echo "== $d"
  SF_LIBRARY_PATH=$PWD/tools/ab/libsf_b3exp9.so GEMM_ONLY=conv1-shape GEMM_SPLIT=1 GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 2>&1 | tail -6
done | tee gpurun_out/b3/stamps.txt
