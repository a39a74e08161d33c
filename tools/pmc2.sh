#!/bin/bash
# wait/ifetch/LDS breakdown of the step kernel (separate PMC passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc2 && mkdir -p gpurun_out/pmc2
ARGS="--steps 200 --warmup 400 --k-per-launch 50 --no-cpu-baseline --no-interactive"
rocprofv3 -L > gpurun_out/pmc2/counters.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_INSTS_EXP_GDS\|SQC_DCACHE[A-Z_]*" gpurun_out/pmc2/counters.txt | sort -u | tr '\n' ' '
echo
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmc2/a -- python3 bench.py $ARGS > gpurun_out/pmc2/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc2/b -- python3 bench.py $ARGS > gpurun_out/pmc2/b.log 2>&1
python3 - <<'PY'
import csv, glob
for run in ("a","b"):
    for f in sorted(glob.glob("gpurun_out/pmc2/%s/*/*_counter_collection.csv" % run))[-1:]:
        rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                print(run, r["Kernel_Name"][:28], r["Counter_Name"], "%.1f per arena-step" % (float(r["Counter_Value"])/4096/50))
PY
