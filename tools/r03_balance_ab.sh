#!/bin/bash
# same-call A/B of the population-ordered launch (k_rank): SF_BALANCE=0 vs 1, default and driver forms, configs[2] and [1]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="--no-cpu-baseline --no-interactive --no-other-configs --no-policy"
for round in 1 2; do
  for wl in C3; do
    for b in 0 1 3 4; do
      for form in "--steps 1000 --warmup 100" "--steps 20 --warmup 5"; do
        SF_BALANCE=$b python3 bench.py $ARGS $form --workload $wl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl balance=$b [$form]', round(d['value']/1e6,1), 'M, launch ms', round(d['roofline']['avg_launch_ms'],3))"
      done
    done
  done
done
