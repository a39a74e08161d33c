#!/bin/bash
# how often k_rank should renew the launch order when launches are short: SF_RANK_EVERY = 20 / 40 / 60 / 100 (default) steps,
# the driver's bench form (twenty-step launches) and the default one, same call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04g
SIDE="--no-cpu-baseline --no-interactive --no-other-configs --no-policy"
for round in 1 2; do
  for ev in 100 60 40 20; do
    for form in "--steps 20 --warmup 5" "--steps 1000 --warmup 100"; do
      SF_RANK_EVERY=$ev python3 bench.py $form $SIDE 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('every $ev | $form |', round(d['value']/1e6,1), 'M steps/s, launch ms', round(d['roofline']['avg_launch_ms'],4))"
    done
  done
done | tee gpurun_out/r04g/rank_every.txt
