#!/usr/bin/env python3
"""Condense rocprofv3 CSV outputs (gpurun_out/prof_*) into small tracked summaries under profiles/.

  python tools/summarize_profiles.py r01 [outdir]
(on the GPU box: outdir = gpurun_out/summary, then delete the raw traces -- gpurun only copies back 64 MiB)
writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim top rows) and
profiles/<tag>_summary.json (per-kernel durations, PMC counters per launch, derived HBM traffic).
"""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def first(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # gpurun merges into gpurun_out: newest run wins


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


def main(tag, outdir=None):
    outdir = outdir or os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    out = {"tag": tag, "source": "rocprofv3 on python3 bench.py (binary-narrow 16x16, 4096 envs, 1 MI355X)"}
    ks = first("prof_kt/**/*kernel_stats.csv")
    if ks:
        rows = list(csv.DictReader(open(ks)))
        keep = [r for r in rows if "pcgrl" in r["Name"]]
        with open(os.path.join(outdir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            for r in keep:
                w.writerow(r)
        out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                "pct": float(r["Percentage"])} for r in keep]
    kt = first("prof_kt/**/*kernel_trace.csv")
    if kt:
        rows = [r for r in csv.DictReader(open(kt)) if "step_kernel" in r["Kernel_Name"]]
        gaps = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
        r0 = rows[0]
        durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
        periods = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(len(rows) - 1)]
        q = lambda f: durs[min(len(durs) - 1, int(f * len(durs)))]
        out["step_kernel_launch"] = {"grid": int(r0["Grid_Size_X"]), "workgroup": int(r0["Workgroup_Size_X"]),
                                     "median_gap_between_launches_ns": statistics.median(gaps),
                                     "duration_ns": {"p10": q(0.1), "median": q(0.5), "p90": q(0.9), "p99": q(0.99),
                                                     "mean": statistics.mean(durs)},
                                     "median_start_to_start_ns": statistics.median(periods)}
    counters = defaultdict(dict)
    for d in ("prof_fetch", "prof_write", "prof_sq"):
        f = first(f"{d}/**/*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pcgrl" in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            counters[k][c] = {"mean_per_launch": statistics.mean(v), "launches": len(v)}
    out["pmc"] = counters
    # HBM traffic of the step kernel per launch.  FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: FETCH_SIZE
    # under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE is calibrated here on observe_kernel, which writes
    # exactly 4096 * 3072 B with the same store pattern as the step kernel.
    step = next((k for k in counters if "step_kernel" in k), None)
    obs = next((k for k in counters if "observe_kernel" in k), None)
    if step and "WRITE_SIZE" in counters[step]:
        wcal = None
        if obs and "WRITE_SIZE" in counters[obs]:
            wcal = (4096 * 3072) / (counters[obs]["WRITE_SIZE"]["mean_per_launch"] * 1024)
        wr = counters[step]["WRITE_SIZE"]["mean_per_launch"] * 1024
        rd = counters[step].get("FETCH_SIZE", {"mean_per_launch": 0})["mean_per_launch"] * 1024
        out["hbm_traffic_per_launch"] = {
            "write_bytes_raw": wr, "write_calibration_factor": wcal, "fetch_bytes_raw": rd,
            "fetch_bytes_x2_correction": 2 * rd, "traffic_bytes": (wr * (wcal or 1.0)) + 2 * rd,
            "algorithmic_bytes": 4096 * 3348}
    with open(os.path.join(outdir, f"{tag}_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", sys.argv[2] if len(sys.argv) > 2 else None)
