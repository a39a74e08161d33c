#!/bin/bash
# Round-2 profiling pass of the bench (gpurun).  Kernel trace/stats and each PMC group run in their own processes (no
# trace domain is combined with --pmc).  Results: gpurun_out/prof (kernel stats, FETCH/WRITE passes per launch length),
# gpurun_out/pmc (SQ instruction mix), gpurun_out/r02p (bench lines, phase stamps).  tools/summarize_prof.py copies the
# summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof gpurun_out/r02p && mkdir -p gpurun_out/prof gpurun_out/r02p
python3 bench.py --steps 1000 --warmup 100 > gpurun_out/r02p/bench_1000.json 2> gpurun_out/r02p/bench_1000.err
python3 bench.py --steps 20 --warmup 5 --no-policy --no-other-configs > gpurun_out/r02p/bench_driver_form.json 2> gpurun_out/r02p/bench_driver_form.err
ARGS="--steps 1000 --warmup 100 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py $ARGS > gpurun_out/prof/bench_stats.log 2>&1
grep -h "sf::" gpurun_out/prof/stats/*/*_kernel_stats.csv | head -12
for K in 100 20; do
  PARGS="--steps 400 --warmup 0 --k-per-launch $K --no-cpu-baseline --no-interactive --no-other-configs --no-policy"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch_$K -- python3 bench.py $PARGS > gpurun_out/prof/bench_fetch_$K.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write_$K -- python3 bench.py $PARGS > gpurun_out/prof/bench_write_$K.log 2>&1
done
BENCH_ARGS="--steps 200 --warmup 100 --k-per-launch 50 --no-cpu-baseline --no-interactive --no-other-configs --no-policy" bash tools/pmc.sh > gpurun_out/r02p/pmc_mix.txt 2>&1
cp gpurun_out/pmc/per_arena_step.json gpurun_out/r02p/instr_mix_per_arena_step.json
[ -f tools/ab/libsf_diag.so ] && SF_LIBRARY_PATH=$PWD/tools/ab/libsf_diag.so python3 tools/diag_stamps.py C3 C2 2>&1 | grep -v amdgpu.ids > gpurun_out/r02p/phase_stamps.txt
echo profile pass done
