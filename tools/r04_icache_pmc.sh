#!/bin/bash
# Instruction-cache requests / misses of k_step per launch, one-step launches against 100-step ones (own PMC passes, no trace domain).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; rm -rf gpurun_out/prof_ic; mkdir -p $O gpurun_out/prof_ic
SIDE="--no-cpu-baseline --no-other-configs --no-interactive --no-policy"
for K in 1 100; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS --output-format csv -d gpurun_out/prof_ic/k$K -- python3 bench.py --steps $((K*30)) --warmup 0 --k-per-launch $K $SIDE > gpurun_out/prof_ic/bench_$K.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d gpurun_out/prof_ic/w$K -- python3 bench.py --steps $((K*30)) --warmup 0 --k-per-launch $K $SIDE > gpurun_out/prof_ic/benchw_$K.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/r04g/icache_pmc.txt
import csv, glob, collections
for K in (1, 100):
    acc = collections.defaultdict(list)
    for pat in ("gpurun_out/prof_ic/k%d/*/*counter_collection.csv" % K, "gpurun_out/prof_ic/w%d/*/*counter_collection.csv" % K):
        for f in glob.glob(pat):
            for r in csv.DictReader(open(f)):
                if "k_step" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the last 30 dispatches are the timed launches of this K (the pre-roll runs 100-step launches)
    print("K=%d" % K, {c: round(sum(v[-30:]) / len(v[-30:])) for c, v in acc.items()}, "dispatches", {c: len(v) for c, v in acc.items()})
PY
