#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a gpurun profiling call from gpurun_out/prof into profiles/ (tracked).

    python tools/summarize_prof.py r01a C2 4096 50

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_kernel_trace_ksteps.csv
(the k_step / k_reset dispatch rows) and merges the PMC passes (FETCH_SIZE / WRITE_SIZE, collected in
their own runs) into profiles/pmc_summary.json, keyed by workload/arenas/steps-per-launch.
"""
import csv
import glob as _glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class glob:  # gpurun merges every call's outputs into gpurun_out/: only the newest file of a pattern is current
    @staticmethod
    def glob(pattern):
        files = sorted(_glob.glob(pattern), key=os.path.getmtime)
        return files[-1:]

tag, workload, arenas, kpl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
timed = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # launches of the bench's timed region (the last ones of that length)
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
# round 3: the driver's own command (--steps 20 --warmup 5: 21 repeats of one 20-step launch, side measurements off, so
# the LAST 21 k_step dispatches are the timed region's) traced in its own run
for f in glob.glob(os.path.join(src, "stats20", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, tag + "_driver_form_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats20", "*", "*_kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    last = d[-21:]
    json.dump({"command": "python bench.py --steps 20 --warmup 5 (side measurements off)", "k_step_dispatches": len(d),
               "timed_region_dispatches": len(last), "k_step_K20_timed_avg_ms": sum(last) / len(last) / 1e6,
               "k_step_K20_timed_min_ms": min(last) / 1e6, "k_step_K20_timed_max_ms": max(last) / 1e6,
               "all_dispatch_ms": [x / 1e6 for x in d]},
              open(os.path.join(dst, tag + "_driver_form_kernel_summary.json"), "w"), indent=1)
    print("driver form: last 21 k_step dispatches avg %.4f ms" % (sum(last) / len(last) / 1e6))
for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "sf::" in r["Kernel_Name"] or "sfp::" in r["Kernel_Name"]]
    with open(os.path.join(dst, tag + "_kernel_trace_sf.csv"), "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=list(rows[0].keys()) + ["Duration_Ns"])
        w.writeheader()
        for r in keep:
            r["Duration_Ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            w.writerow(r)
    # the bench's timed region = the last `steps / kpl` k_step dispatches before the K = 1 interactive section:
    # take the k_step dispatches longer than half the longest one (K = kpl launches; K = 1 launches are ~20x shorter)
    d = [r["Duration_Ns"] for r in keep if "k_step" in r["Kernel_Name"]]
    big = [x for x in d if x > 0.5 * max(d)] if d else []
    summary = {"k_step_dispatches": len(d), "k_step_K%d_dispatches" % kpl: len(big),
               "k_step_K%d_avg_ms" % kpl: sum(big) / len(big) / 1e6 if big else None,
               # the bench's timed region = the last `timed` launches of that length (the earlier ones are the untimed
               # pre-roll and warm-up, with populations still growing): this is the figure bench.py's HIP events give
               "k_step_K%d_timed_launches" % kpl: timed or None,
               "k_step_K%d_timed_avg_ms" % kpl: (sum(big[-timed:]) / timed / 1e6) if (timed and len(big) >= timed) else None,
               "k_observe_avg_ms": (lambda o: sum(o) / len(o) / 1e6 if o else None)(
                   [r["Duration_Ns"] for r in keep if "k_observe" in r["Kernel_Name"]])}
    json.dump(summary, open(os.path.join(dst, tag + "_kernel_summary.json"), "w"), indent=1)
    print(summary)

pmc = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    pats = [os.path.join(src, "pmc_%s_%d" % (kind, kpl), "*", "*_counter_collection.csv"),
            os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv")]
    files = glob.glob(pats[0]) or glob.glob(pats[1])
    for f in files:
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if "k_step" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        if vals:
            # skip the warm-up launch (first): it runs the lighter early-episode steps
            v = vals[1:] if len(vals) > 1 else vals
            pmc[counter + "_KB_per_launch"] = sum(v) / len(v)
            pmc[counter + "_launches"] = len(v)
if pmc:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads:
    # doubled for the 16-B/lane flag-plane stream; WRITE_SIZE is exact.  KB = 1024 B.
    fetch = pmc.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024
    write = pmc.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
    pmc["hbm_bytes_per_launch_corrected"] = 2 * fetch + write
    pmc["hbm_bytes_per_launch_uncorrected"] = fetch + write
    pmc["tag"] = tag
    path = os.path.join(dst, "pmc_summary.json")
    allp = json.load(open(path)) if os.path.exists(path) else {}
    allp["%s/%d/%d" % (workload, arenas, kpl)] = pmc
    json.dump(allp, open(path, "w"), indent=1, sort_keys=True)
print("profiles/ updated:", sorted(os.listdir(dst)))
