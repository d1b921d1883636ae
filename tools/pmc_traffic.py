#!/usr/bin/env python3
"""Fold the PMC passes of tools/profile_headline.sh into profiles/<tag>_pmc_traffic.json (what bench.py's
`roofline.traffic` reads).  The record names the hash of the kernel sources it was measured at; bench.py reports no
traffic for other sources.

    python tools/pmc_traffic.py <prof dir> <out json>

Calibration: tools/micro/tile_probe runs kernels with KNOWN byte counts in this kernel's own access pattern (10M points:
16-byte coalesced loads = 160 MB in, 4-byte stores = 40 MB per output array, plus 4-byte gathers that hit the XCD L2).
factor_read = 160e6 / FETCH_SIZE(probe mode 0), factor_write = 40e6 / WRITE_SIZE(probe mode 0); the bench kernel's counters
are multiplied by these factors (MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-byte requests at 64 bytes for wide
coalesced reads; other widths must be calibrated)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def counters(root, pat):
    """{counter: mean value per dispatch} and dispatch count for kernels whose name contains `pat` under root/**."""
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, max((len(v) for v in agg.values()), default=0)


def per_dispatch(root, pat):
    """[(kernel name, dispatch id, {counter: value})] in dispatch order."""
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                rows.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [(k[1], k[0], v) for k, v in sorted(rows.items())]


def durations_us(root, pat):
    d = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        d += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    return d


def main():
    root, out = sys.argv[1], sys.argv[2]
    import bench  # source_hash()

    KERNEL = "pip_tile"
    rec = {"kernel": "gpk_pip_tile", "source_hash": bench.source_hash(), "command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-join-stats --no-default-shape (cold inputs: 3 rotating 10M-point sets)"}
    vals = {}
    for d in glob.glob(os.path.join(root, "pmc_*")):
        if os.path.isdir(d):
            c, n = counters(d, KERNEL)
            vals.update(c)
            rec.setdefault("dispatches", n)
    # probe calibration (mode 0 = the first instantiation probe<0, 2>: 160 MB in, one 40 MB output)
    cal = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = per_dispatch(os.path.join(root, "probe_" + name), "probe")
        mode0 = [v[name] for k, _, v in rows if "probe<0" in k or "Li0ELi2" in k and name in v]
        if not mode0:
            mode0 = [v[name] for _, _, v in rows[:12] if name in v]
        if mode0:
            cal[name] = sum(mode0) / len(mode0)  # KB
    f_read = 160e6 / (cal["FETCH_SIZE"] * 1024.0) if cal.get("FETCH_SIZE") else None
    f_write = 40e6 / (cal["WRITE_SIZE"] * 1024.0) if cal.get("WRITE_SIZE") else None
    rec["counters_per_launch"] = vals
    rec["probe_counters_KB"] = cal
    rec["factor_read"], rec["factor_write"] = f_read, f_write
    fetch = vals.get("FETCH_SIZE", 0.0) * 1024.0
    write = vals.get("WRITE_SIZE", 0.0) * 1024.0
    rec["fetch_bytes_raw"], rec["write_bytes_raw"] = fetch, write
    rec["traffic_bytes_per_launch"] = int(fetch * (f_read or 1.0) + write * (f_write or 1.0))
    # The read factor is exact only for the coalesced 16-byte point stream (what the probe reads).  Split estimate: the 160 MB
    # of points account for 160e6 / factor of the raw counter; whatever else was fetched — index gathers that missed the
    # XCD L2 — is taken at the counter's own 64 bytes per request.  The truth lies between this and the figure above.
    if f_read:
        pts_raw = 160e6 / f_read
        rec["traffic_bytes_split_estimate"] = int(160e6 + max(fetch - pts_raw, 0.0) + write * (f_write or 1.0))
        rec["traffic_note"] = "traffic_bytes_per_launch applies the probe's read factor to every fetched byte (upper bound); traffic_bytes_split_estimate applies it to the point stream only and counts the remaining requests at 64 bytes"
    rec["calibration"] = (
        f"FETCH_SIZE x {f_read:.3f}, WRITE_SIZE x {f_write:.3f}: factors that make tools/micro/tile_probe mode 0 (160 MB of 16-byte coalesced loads in, 40 MB of 4-byte stores out, known byte counts) read its true size"
        if f_read and f_write
        else "uncalibrated: probe passes missing; raw FETCH_SIZE + WRITE_SIZE"
    )
    if vals.get("TCC_HIT_sum") is not None and vals.get("TCC_MISS_sum") is not None:
        rec["l2_hit_rate"] = vals["TCC_HIT_sum"] / max(vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"], 1.0)
    dur = durations_us(os.path.join(root, "stats"), KERNEL) or [x for d in glob.glob(os.path.join(root, "pmc_*")) for x in durations_us(d, KERNEL)]
    if dur:
        rec["kernel_us_rocprof_mean"] = sum(dur) / len(dur)
        rec["kernel_us_rocprof_median"] = sorted(dur)[len(dur) // 2]  # (the first launch of a process runs several times longer)
    if vals.get("SQ_INSTS_VALU") and dur:
        # every vector instruction occupies its SIMD for 4 cycles (wave64 on 16 lanes); 256 CUs x 4 SIMDs at 2.4 GHz
        cycles = sorted(dur)[len(dur) // 2] * 1e-6 * 2.4e9
        rec["valu_busy"] = vals["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles)
        rec["valu_busy_note"] = "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz, median launch)"
        if vals.get("SQ_WAVES"):
            rec["instr_per_wave"] = {k: vals[k] / vals["SQ_WAVES"] for k in vals if k.startswith("SQ_INSTS_")}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec)[:600])


if __name__ == "__main__":
    main()
