#!/bin/bash
# ablations of k_gemm_b3 (tools/ab/libsf_b3exp<N>.so built with -DB3_EXP=N): timing only, results are wrong by design
#  1 no lstore, no barrier   2 MFMAs only   3 no split arithmetic   4 no global loads in the loop   5 no barrier
mkdir -p gpurun_out/b3
for d in zero rand; do
  for lib in strikeforce_amd/libstrikeforce_amd.so tools/ab/libsf_b3exp1.so tools/ab/libsf_b3exp2.so tools/ab/libsf_b3exp4.so tools/ab/libsf_b3exp5.so tools/ab/libsf_b3exp6.so; do
    echo "$d $lib $(SF_LIBRARY_PATH=$PWD/$lib GEMM_SPLIT=1 GEMM_DATA=$d GEMM_ONLY=conv1-shape timeout -k 10 120 python tools/gemm_shapes.py 4096)"
  done
done | tee gpurun_out/b3/exp.txt
