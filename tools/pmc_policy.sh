#!/bin/bash
# PMC passes over the policy network's kernels (run through gpurun).  Counter collection only, one group per process.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_policy && mkdir -p gpurun_out/pmc_policy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d gpurun_out/pmc_policy/a -- python3 tools/policy_bench.py 4096 4 > gpurun_out/pmc_policy/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --output-format csv -d gpurun_out/pmc_policy/b -- python3 tools/policy_bench.py 4096 4 > gpurun_out/pmc_policy/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_policy/f -- python3 tools/policy_bench.py 4096 4 > gpurun_out/pmc_policy/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_policy/w -- python3 tools/policy_bench.py 4096 4 > gpurun_out/pmc_policy/w.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
summary = {}
for d in "abfw":
    fs = glob.glob("gpurun_out/pmc_policy/%s/*/*counter_collection.csv" % d)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    seen = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        seen[(r["Kernel_Name"][:60], r["Dispatch_Id"])] += 1
    for k, _ in seen: cnt[k] += 1
    for k in acc:
        if "gemm" in k or "observe" in k:
            print(k, "dispatches", cnt[k], {c: v / cnt[k] for c, v in acc[k].items()})
            summary.setdefault(k, {"dispatches": cnt[k]}).update({c: v / cnt[k] for c, v in acc[k].items()})
json.dump(summary, open('gpurun_out/pmc_policy/summary.json', 'w'), indent=1)
PY
