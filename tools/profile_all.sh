#!/bin/bash
# Full profiling pass of the bench on the GPU box (run through gpurun).  Kernel trace/stats and each PMC group run in
# their own processes (no trace domains are combined with --pmc).  Results land in gpurun_out/prof and gpurun_out/pmc;
# tools/summarize_prof.py copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
ARGS="--steps 1000 --warmup 100 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py $ARGS > gpurun_out/prof/bench_stats.log 2>&1
grep -h "sf::" gpurun_out/prof/stats/*/*_kernel_stats.csv
PARGS="--steps 200 --warmup 400 --no-cpu-baseline --no-interactive"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py $PARGS > gpurun_out/prof/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py $PARGS > gpurun_out/prof/bench_write.log 2>&1
bash tools/pmc.sh
