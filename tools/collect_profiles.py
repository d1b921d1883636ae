#!/usr/bin/env python3
"""Copy one evidence run (tools/gpu_round.sh <tag>, merged back under gpurun_out/) into profiles/ under stable names:
    python tools/collect_profiles.py <tag>
Only summaries are copied (bench lines, test summary, kernel stats, the headline kernel's PMC rows, the traffic records)."""
import json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def json_lines(path):
    out = []
    if os.path.exists(path):
        for l in open(path, errors="replace"):
            l = l.strip()
            if l.startswith("{") and '"metric"' in l:
                try:
                    json.loads(l)
                    out.append(l)
                except Exception:
                    pass
    return out


lines = []
for c in ("c2", "c3", "c4", "c5"):
    lines += json_lines(os.path.join(G, f"{tag}_bench_{c}.log"))[-1:]
with open(os.path.join(P, f"{tag}_bench_lines.jsonl"), "w") as f:
    f.write("\n".join(lines) + "\n")
print(len(lines), "bench lines")
t = os.path.join(G, f"{tag}_gputests.log")
if os.path.exists(t):
    keep = [l for l in open(t, errors="replace") if ("passed" in l or "failed" in l or l.startswith(("real", "user", "sys")))]
    open(os.path.join(P, f"{tag}_gputests_summary.txt"), "w").write("".join(keep))
for src, dst in (
    (f"{tag}_kernel_stats.csv", f"{tag}_c2_kernel_stats.csv"),
    (f"{tag}_c3_kernel_stats.csv",) * 2,
    (f"{tag}_c4_kernel_stats.csv",) * 2,
    (f"{tag}_c5_kernel_stats.csv",) * 2,
    (f"{tag}_ops.jsonl",) * 2,
    (f"{tag}_ops_kernel_stats.csv",) * 2,
    (f"{tag}_reference_benches.jsonl",) * 2,
    (f"{tag}_sizes.txt",) * 2,
    (f"{tag}_timeline_trace1.txt", f"{tag}_timeline_flow.txt"),
    (f"{tag}_timeline_trace2.txt", f"{tag}_timeline_flow_phases.txt"),
    (f"{tag}_c2_pmc_counters.txt",) * 2,
    (f"prof_{tag}/{tag}_pmc_traffic.json", f"{tag}_pmc_traffic.json"),
    (f"prof_{tag}_cfg/{tag}_pmc_traffic_configs.json", f"{tag}_pmc_traffic_configs.json"),
):
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
    else:
        print("missing:", src)
under = json_lines(os.path.join(G, f"prof_{tag}", "stats.log"))
if under:
    open(os.path.join(P, f"{tag}_c2_bench_line_under_rocprof.jsonl"), "w").write(under[-1] + "\n")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_extract.py"), os.path.join(G, f"prof_{tag}"), "pip_tile", os.path.join(P, f"{tag}_c2_pmc_passes.csv")], check=False)
