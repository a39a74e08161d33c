#!/bin/bash
# Round 3, after the convolution stack was folded: both bench forms, the kernel trace of the closed loop
# observe -> network -> sample -> step (bench.py's policy section, configs[2], 4096 agents), and the L2 counters of its two
# kernels (own processes: no trace domain is combined with --pmc).  gpurun: bash tools/r03_policy_trace.sh [tag]
TAG=${1:-r03e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG
rm -rf gpurun_out/prof_pol $O && mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
SIDE="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pol/trace -- python3 bench.py $SIDE > $O/policy_bench_under_rocprof.json 2> $O/policy_rocprof.err
cp gpurun_out/prof_pol/trace/*/*_kernel_stats.csv $O/policy_loop_kernel_stats.csv
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/prof_pol/pmc -- python3 bench.py $SIDE > $O/policy_pmc.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, collections, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/prof_pol/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_feat_list' in k or 'k_tail' in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
out = {k: {c: acc[k][c] / n[(k, c)] for c in acc[k]} for k in acc}
for k in out:
    out[k]['l2_request_bytes'] = out[k].get('TCC_REQ_sum', 0) * 128
json.dump({"what": "per launch, 4096 agents of configs[2]; TCC requests are 128 bytes", "kernels": out}, open(sys.argv[1] + '/policy_l2_counters.json', 'w'), indent=1)
print(json.dumps(out))
PY
cut -c1-150 $O/policy_loop_kernel_stats.csv | head -16
