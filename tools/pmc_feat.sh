cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_feat && mkdir -p gpurun_out/pmc_feat
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmc_feat/a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_feat/a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmc_feat/b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_feat/b.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ('a','b'):
    for f in glob.glob('gpurun_out/pmc_feat/%s/*/*counter_collection.csv'%d):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            if 'k_feat_list' in k or 'k_tail' in k:
                acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k in acc:
            print(k, {c: acc[k][c]/n[(k,c)] for c in acc[k]})
PY
