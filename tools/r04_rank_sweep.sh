#!/bin/bash
# k_rank (launch order by population) on or off, same call: arena counts 2048..16384 on configs[2], and the HBM-plane
# configurations (SF_BALANCE=2 forces the order there).  Output: gpurun_out/r04b/rank_sweep.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04b
ARGS="--steps 600 --warmup 100 --no-cpu-baseline --no-interactive --no-other-configs --no-policy"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), 'M env-steps/s, launch ms', round(d['roofline']['avg_launch_ms'],3))"; }
for round in 1 2; do
  for A in 2048 4096 8192 16384 5000; do
    for b in 0 1; do
      SF_BALANCE=$b python3 bench.py $ARGS --workload C3 --arenas $A 2>/dev/null | line "C3 arenas=$A SF_BALANCE=$b"
    done
  done
  for wl in C4 C5; do
    for b in 0 2; do
      SF_BALANCE=$b python3 bench.py $ARGS --workload $wl --arenas 4096 2>/dev/null | line "$wl arenas=4096 SF_BALANCE=$b"
    done
  done
done | tee gpurun_out/r04b/rank_sweep.txt
