#!/bin/bash
# bf16-split conv kernel: parity tests, then A/B of the forward with conv1/conv2 on the f32 and on the split kernel
set -e
mkdir -p gpurun_out/b3
timeout -k 10 500 python -m pytest tests/test_gpu_policy.py -x -q -k "split or forward_matches or large_batch" > gpurun_out/b3/tests.log 2>&1 || { tail -30 gpurun_out/b3/tests.log; exit 1; }
tail -3 gpurun_out/b3/tests.log
SF_POLICY_F32_CONV=1 timeout -k 10 200 python tools/policy_bench.py 4096 20 > gpurun_out/b3/f32.json
timeout -k 10 200 python tools/policy_bench.py 4096 20 > gpurun_out/b3/split.json
cat gpurun_out/b3/f32.json gpurun_out/b3/split.json
for d in rand zero; do
  GEMM_SPLIT=1 GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 > gpurun_out/b3/shapes_split_$d.json
done
head -3 gpurun_out/b3/shapes_split_*.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/b3/prof gpurun_out/b3/pmc_a gpurun_out/b3/pmc_f
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b3/prof -- python3 tools/policy_bench.py 4096 20 > gpurun_out/b3/prof.log 2>&1
grep -h "b3\|conv0\|fixup" gpurun_out/b3/prof/*/*_kernel_stats.csv | cut -c1-160
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \
  --output-format csv -d gpurun_out/b3/pmc_a -- python3 tools/policy_bench.py 4096 4 > gpurun_out/b3/pmc_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/b3/pmc_f -- python3 tools/policy_bench.py 4096 4 > gpurun_out/b3/pmc_f.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_a","pmc_f"):
    fs = glob.glob("gpurun_out/b3/%s/*/*counter_collection.csv" % d)
    if not fs: print(d,"no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "b3" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ids = sorted(acc, key=int)[-2:]
    for i in ids: print(d, i, {c: sum(v) for c, v in acc[i].items()})
PY
