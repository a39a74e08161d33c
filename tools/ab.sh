#!/bin/bash
# A/B two (or more) builds of the library in ONE gpurun call (boxes differ by ~10 % in clocks, so numbers from
# different calls do not compare).  usage: WL="C2 C3" tools/ab.sh A.so B.so [...]   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ARGS="${BENCH_ARGS:---steps 600 --warmup 100 --no-cpu-baseline --no-interactive --no-other-configs --no-policy}"
for round in 1 2; do
  for wl in ${WL:-C3 C2}; do
    for lib in "$@"; do
      SF_LIBRARY_PATH=$PWD/$lib python3 bench.py $ARGS --workload $wl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $lib', round(d['value']/1e6,1), 'M steps/s, launch ms', round(d['roofline']['avg_launch_ms'],3))"
    done
  done
done
