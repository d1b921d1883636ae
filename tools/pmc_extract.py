#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection.csv files (one PMC pass per directory) to the rows of one kernel:
    python tools/pmc_extract.py <prof dir> <kernel substring> <out csv>
Columns: pass, dispatch, kernel (short), counter, value, duration_us."""
import csv, glob, os, sys

root, pat, out = sys.argv[1:4]
rows = []
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    pas = os.path.relpath(f, root).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if r.get("End_Timestamp") else ""
            rows.append([pas, r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"], dur])
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["pass", "dispatch", "kernel", "counter", "value", "duration_us"])
    w.writerows(rows)
print(len(rows), "rows ->", out)
