#!/bin/bash
# timing experiments on k_gemm_b3 (tools/ab/libsf_b3exp<N>.so built with -DB3_EXP=N; results are wrong by design)
#  6 every tile reads the first tile's A rows (A from L2)   7 the loader waves stop after a few units (compute waves alone)
mkdir -p gpurun_out/b3
for d in zero rand; do
  for lib in strikeforce_amd/libstrikeforce_amd.so tools/ab/libsf_b3exp6.so tools/ab/libsf_b3exp7.so; do
    echo "$d $lib $(SF_LIBRARY_PATH=$PWD/$lib GEMM_SPLIT=1 GEMM_DATA=$d GEMM_ONLY=conv1-shape timeout -k 10 120 python tools/gemm_shapes.py 4096)"
  done
done | tee gpurun_out/b3/exp.txt
