#!/usr/bin/env python3
"""Per arena-step instruction mix of the last k_step dispatch from gpurun_out/pmc (see tools/pmc.sh)."""
import csv
import glob
import json
import sys

arenas, kpl = 4096, 50
out = {}
for run in ("sq1", "sq2"):
    for f in glob.glob("gpurun_out/pmc/%s/*/*_counter_collection.csv" % run):
        rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                out[r["Counter_Name"]] = float(r["Counter_Value"]) / arenas / kpl
for k in sorted(out):
    print("%-24s %10.1f per arena-step" % (k, out[k]))
json.dump(out, open("gpurun_out/pmc/per_arena_step.json", "w"), indent=1, sort_keys=True)
