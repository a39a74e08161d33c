#!/bin/bash
set -e
mkdir -p gpurun_out/b3
for d in rand zero; do
  GEMM_SPLIT=1 GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 > gpurun_out/b3/shapes_split_$d.json
  GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 > gpurun_out/b3/shapes_f32_$d.json
done
head -3 gpurun_out/b3/shapes_*.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/b3/pmc_a gpurun_out/b3/pmc_b
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d gpurun_out/b3/pmc_a -- python3 tools/policy_bench.py 4096 4 > gpurun_out/b3/pmc_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
  --output-format csv -d gpurun_out/b3/pmc_b -- python3 tools/policy_bench.py 4096 4 > gpurun_out/b3/pmc_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_a","pmc_b"):
    fs = glob.glob("gpurun_out/b3/%s/*/*counter_collection.csv" % d)
    if not fs: print(d,"no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "b3" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ids = sorted(acc, key=int)[-2:]
    for i in ids: print(d, i, {c: sum(v) for c, v in acc[i].items()})
PY
