#!/bin/bash
# round-2 record of the policy row: all GPU policy tests, forward / closed-loop timing, kernel trace, MFMA-busy and HBM counters
set -e
T=${TAG:-r02c}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests/test_gpu_policy.py -x -q > gpurun_out/$T/policy_tests.log 2>&1 || { tail -30 gpurun_out/$T/policy_tests.log; exit 1; }
tail -1 gpurun_out/$T/policy_tests.log
SF_POLICY_F32_CONV=1 timeout -k 10 200 python tools/policy_bench.py 4096 20 > gpurun_out/$T/policy_bench_f32conv.json
timeout -k 10 200 python tools/policy_bench.py 4096 20 > gpurun_out/$T/policy_bench.json
cat gpurun_out/$T/policy_bench_f32conv.json gpurun_out/$T/policy_bench.json
for d in zero rand; do
  GEMM_SPLIT=1 GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 > gpurun_out/$T/gemm_shapes_split_$d.json
  GEMM_DATA=$d timeout -k 10 200 python tools/gemm_shapes.py 4096 > gpurun_out/$T/gemm_shapes_f32_$d.json
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/$T/prof gpurun_out/$T/pmc_a gpurun_out/$T/pmc_f gpurun_out/$T/pmc_w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof -- python3 tools/policy_bench.py 4096 20 > gpurun_out/$T/prof.log 2>&1
cp gpurun_out/$T/prof/*/*_kernel_stats.csv gpurun_out/$T/policy_kernel_stats.csv
cut -c1-150 gpurun_out/$T/policy_kernel_stats.csv | head -20
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU \
  --output-format csv -d gpurun_out/$T/pmc_a -- python3 tools/policy_bench.py 4096 4 > gpurun_out/$T/pmc_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$T/pmc_f -- python3 tools/policy_bench.py 4096 4 > gpurun_out/$T/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$T/pmc_w -- python3 tools/policy_bench.py 4096 4 > gpurun_out/$T/pmc_w.log 2>&1
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in ("pmc_a", "pmc_f", "pmc_w"):
    fs = glob.glob("gpurun_out/$T/%s/*/*counter_collection.csv" % d)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    for k in acc:
        if "gemm" in k or "conv0" in k:
            out.setdefault(k, {})["dispatches"] = len(seen[k])
            out[k].update({c: v / len(seen[k]) for c, v in acc[k].items()})
json.dump(out, open("gpurun_out/$T/policy_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
PY
