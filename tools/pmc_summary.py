#!/usr/bin/env python3
"""Summarise tools/pmc_ablate.sh output: per-wave means of every counter for kernels matching a name."""
import csv, collections, glob, sys
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "pip_tile")
for vdir in sorted(glob.glob(root + "/*/")):
    agg, waves, dur = collections.defaultdict(list), None, []
    for f in glob.glob(vdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                waves = int(r["Grid_Size"]) // 64
    for f in glob.glob(vdir + "/**/*kernel_trace.csv", recursive=True):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if not waves: continue
    print(vdir.rstrip("/").split("/")[-1], "waves=%d us=%.1f" % (waves, sum(dur) / max(len(dur), 1)))
    for k in sorted(agg): print("   %-26s %10.1f /wave" % (k, sum(agg[k]) / len(agg[k]) / waves))
