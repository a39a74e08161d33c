#!/bin/bash
# round-2 opening probe (gpurun): A/B of the product library against an experimental build whose draws are free
# (upper bound of any RNG restructuring), on C2 and C3; then kernel stats and the SQ instruction mix of C3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
B="--steps 1000 --warmup 200 --no-cpu-baseline --no-interactive --no-other-configs --no-policy"
for wl in C2 C3 C4 C5; do
  for lib in strikeforce_amd/libstrikeforce_amd.so tools/ab/libsf_fakerng.so; do
    SF_LIBRARY_PATH=$PWD/$lib python3 bench.py $B --workload $wl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $lib', round(d['value']/1e6,1), 'M steps/s, launch ms', round(d['roofline']['avg_launch_ms'],3))" | tee -a gpurun_out/r02a/ab.txt
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/stats -- python3 bench.py $B --workload C3 > gpurun_out/r02a/bench_stats.log 2>&1
rm -rf gpurun_out/pmc
BENCH_ARGS="--steps 200 --warmup 400 --k-per-launch 50 --no-cpu-baseline --no-interactive --workload C3" bash tools/pmc.sh > gpurun_out/r02a/pmc_c3.txt 2>&1
cp gpurun_out/pmc/per_arena_step.json gpurun_out/r02a/c3_per_arena_step.json
