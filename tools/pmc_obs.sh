#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmco && mkdir -p gpurun_out/pmco
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmco/stats -- python3 tools/obs_only.py > gpurun_out/pmco/stats.log 2>&1
grep -h "k_observe" gpurun_out/pmco/stats/*/*_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmco/sq -- python3 tools/obs_only.py > gpurun_out/pmco/sq.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmco/wr -- python3 tools/obs_only.py > gpurun_out/pmco/wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmco/rd -- python3 tools/obs_only.py > gpurun_out/pmco/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmco/tcc -- python3 tools/obs_only.py > gpurun_out/pmco/tcc.log 2>&1
python3 - <<'PY'
import csv, glob
for run in ("sq","wr","rd","tcc"):
    for f in glob.glob("gpurun_out/pmco/%s/*/*_counter_collection.csv" % run):
        rows = [r for r in csv.DictReader(open(f)) if "k_observe" in r["Kernel_Name"]]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                print(run, r["Counter_Name"], "%.4g" % float(r["Counter_Value"]))
PY
