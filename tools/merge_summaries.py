#!/usr/bin/env python3
"""profiles/<tag>_summary.json + <tag>_kernel_stats.csv from the partial summaries of several tools/profile_all.sh calls
(gpurun limits a call's duration; the passes of a round run in batches):  python tools/merge_summaries.py r05 r05a r05b ..."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "gpurun_out", "summary")
tag, parts = sys.argv[1], sys.argv[2:]
out, rows, fields = None, [], None
for p in parts:
    d = json.load(open(os.path.join(S, f"{p}_summary.json")))
    if out is None:
        out = dict(d, tag=tag, batches=parts)
    else:
        if d.get("kernel_sources_sha16") != out.get("kernel_sources_sha16"):
            sys.exit(f"{p}: profiled on other kernels ({d.get('kernel_sources_sha16')} vs {out.get('kernel_sources_sha16')}): not merged")
        out["workloads"].update(d["workloads"])
        out["hbm_traffic_per_launch_by_workload"].update(d["hbm_traffic_per_launch_by_workload"])
    f = os.path.join(S, f"{p}_kernel_stats.csv")
    if os.path.exists(f):
        r = list(csv.DictReader(open(f)))
        if r:
            fields = fields or list(r[0].keys())
            rows += r
if "binary-narrow" in out["workloads"] and "hbm_traffic_per_launch" in out["workloads"]["binary-narrow"]:
    out["hbm_traffic_per_launch"] = out["workloads"]["binary-narrow"]["hbm_traffic_per_launch"]  # (bench.py's headline lookup)
json.dump(out, open(os.path.join(S, f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(S, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=fields)
    w.writeheader()
    for r in rows:
        w.writerow(r)
print(tag, len(out["workloads"]), "entries,", len(rows), "kernel-stat rows")
