#!/bin/bash
set -e
mkdir -p gpurun_out/b3
timeout -k 10 300 python -m pytest tests/test_gpu_policy.py -x -q -k "split" > gpurun_out/b3/tests.log 2>&1 || { tail -30 gpurun_out/b3/tests.log; exit 1; }
tail -1 gpurun_out/b3/tests.log
for lib in strikeforce_amd/libstrikeforce_amd.so $EXTRA_LIBS; do
for d in zero rand; do
  for sh in conv1-shape conv2-shape; do
  echo "$lib $d $(SF_LIBRARY_PATH=$PWD/$lib GEMM_ONLY=$sh GEMM_SPLIT=1 GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096)"
  done
done
echo "$lib $(SF_LIBRARY_PATH=$PWD/$lib timeout -k 10 200 python tools/policy_bench.py 4096 20 | head -1)"
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/b3/pmc_b
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  --output-format csv -d gpurun_out/b3/pmc_b -- python3 tools/policy_bench.py 4096 4 > gpurun_out/b3/pmc_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_b",):
    fs = glob.glob("gpurun_out/b3/%s/*/*counter_collection.csv" % d)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "b3" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ids = sorted(acc, key=int)[-2:]
    for i in ids: print(d, i, {c: sum(v) for c, v in acc[i].items()})
PY
