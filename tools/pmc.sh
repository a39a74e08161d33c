#!/bin/bash
# Collect SQ instruction-mix counters for the bench's step kernel (run on the GPU box via gpurun).
# PMC passes run in their own processes, with no trace domain besides counter collection.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
ARGS="${BENCH_ARGS:---steps 200 --warmup 400 --k-per-launch 50 --no-cpu-baseline --no-interactive}"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH \
  --output-format csv -d gpurun_out/pmc/sq1 -- python3 bench.py $ARGS > gpurun_out/pmc/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d gpurun_out/pmc/sq2 -- python3 bench.py $ARGS > gpurun_out/pmc/sq2.log 2>&1
python3 tools/pmc_report.py
