#!/bin/bash
# A/B builds of the library on the policy section of bench.py (configs[2], 4096 agents) in ONE gpurun call.
# usage: tools/ab_policy.sh A.so B.so [...]   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in "$@"; do
    SF_LIBRARY_PATH=$PWD/$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys,json; p=json.loads(sys.stdin.read())['policy']; k=p['kernels']
print('$lib', 'feat %.4f tail %.4f forward %.4f / x4 %.4f loop %.4f ms' % (k['k_feat_list']['ms_per_forward'], k['k_tail']['ms_per_forward'], p['forward_ms']['default_init'], p['forward_ms']['weights_x4'], p['ms_per_step']))"
  done
done
