#!/usr/bin/env python3
"""Fold the PMC passes of tools/profile_configs.sh into profiles/<tag>_pmc_traffic_configs.json: per configuration the
dominant kernel's FETCH_SIZE / WRITE_SIZE per launch (raw counter bytes) and the figure with the 16-byte-stream read factor
of the headline calibration (tools/pmc_traffic.py) applied — these kernels mix coalesced streams with gathers, so the truth
lies between the two.  The record names the source hash; bench.py reports it only while the hash matches.
    python tools/pmc_configs.py <prof dir> <out json>"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DOMINANT = {"c3": "distance_grouped", "c4": "pair_refine", "c5": "pip_tile"}
READ_FACTOR = 1.9


def mean_counter(root, pat, name):
    v = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main():
    root, out = sys.argv[1], sys.argv[2]
    import bench

    rec = {"source_hash": bench.source_hash(), "read_factor_of_the_16_byte_stream": READ_FACTOR, "configs": {}}
    for c, pat in DOMINANT.items():
        f, nf = mean_counter(os.path.join(root, f"{c}_FETCH_SIZE"), pat, "FETCH_SIZE")
        w, nw = mean_counter(os.path.join(root, f"{c}_WRITE_SIZE"), pat, "WRITE_SIZE")
        if f is None or w is None:
            continue
        rec["configs"][c] = {
            "kernel": pat,
            "launches": min(nf, nw),
            "fetch_bytes_raw": f * 1024.0,
            "write_bytes_raw": w * 1024.0,
            "traffic_bytes_raw": int((f + w) * 1024.0),
            "traffic_bytes_with_read_factor": int((f * READ_FACTOR + w) * 1024.0),
        }
        v, nv = mean_counter(os.path.join(root, f"{c}_SQ_INSTS_VALU"), pat, "SQ_INSTS_VALU")
        sa, _ = mean_counter(os.path.join(root, f"{c}_SQ_INSTS_VALU"), pat, "SQ_INSTS_SALU")
        if v is not None:
            rec["configs"][c]["valu_wave_instructions_per_launch"] = v
            rec["configs"][c]["salu_wave_instructions_per_launch"] = sa
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec)[:900])


if __name__ == "__main__":
    main()
