#!/bin/bash
# Round-3 profiling pass (gpurun).  The kernel trace / stats and every PMC group run in their own processes (no trace
# domain is ever combined with --pmc).  The traced command is the DRIVER's bench command (`--steps 20 --warmup 5`; the
# side measurements switched off so that the last 21 k_step dispatches are exactly the timed region's: 21 repeats of
# one 20-step launch) and the default one (`--steps 1000 --warmup 100`: ten 100-step launches).
# tools/summarize_prof.py then copies the summaries into profiles/ under the tag given as $1 (default r03a).
TAG=${1:-r03a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG
rm -rf gpurun_out/prof $O && mkdir -p gpurun_out/prof $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
SIDE="--no-cpu-baseline --no-other-configs --no-interactive --no-policy"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats20 -- python3 bench.py --steps 20 --warmup 5 $SIDE > $O/bench_driver_form_under_rocprof.json 2> $O/rocprof20.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --steps 1000 --warmup 100 $SIDE > $O/bench_under_rocprof.json 2> $O/rocprof.err
grep -h "sf::" gpurun_out/prof/stats20/*/*_kernel_stats.csv | cut -c1-160 | head -6
for K in 100 20; do
  PARGS="--steps 400 --warmup 0 --k-per-launch $K $SIDE"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch_$K -- python3 bench.py $PARGS > gpurun_out/prof/bench_fetch_$K.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write_$K -- python3 bench.py $PARGS > gpurun_out/prof/bench_write_$K.log 2>&1
done
BENCH_ARGS="--steps 200 --warmup 100 --k-per-launch 50 $SIDE" bash tools/pmc.sh > $O/pmc_mix.txt 2>&1
cp gpurun_out/pmc/per_arena_step.json $O/instr_mix_per_arena_step.json
for l in libsf_diag libsf_fakediag; do
  [ -f tools/ab/$l.so ] && { echo "== $l"; SF_LIBRARY_PATH=$PWD/tools/ab/$l.so python3 tools/diag_stamps.py C3 C2 2>&1 | grep -v amdgpu.ids; } >> $O/phase_stamps.txt
done
echo profile pass done
