#!/usr/bin/env python3
"""Condense rocprofv3 CSV outputs (gpurun_out/prof_<workload>_{kt,fetch,write,sq}) into small tracked summaries.

  python tools/summarize_profiles.py r02 [outdir]
(on the GPU box: outdir = gpurun_out/summary, then delete the raw traces -- gpurun only copies back 64 MiB)
writes <outdir>/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats rows of the engine's kernels, verbatim, one block
per workload) and <outdir>/<tag>_summary.json (per workload: per-kernel durations, distribution of the dominant kernel's
begin-end times, PMC counters per launch, derived HBM traffic, LDS bank-conflict rate, waves per CU).
Produced by tools/profile_all.sh.
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
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ENVS = {w: v[3] for w, v in bench.WORKLOADS.items()}  # envs per GPU of every bench workload


def first(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # gpurun merges into gpurun_out: newest run wins


def short(name):
    return name.split("(")[0].replace("void ", "")[:72]


def is_m3_step(name):
    return "m3_kernel<0" in name or "m3_kernel<(pcgrl::M3Mode)0" in name  # (<0, SC>, <0, 0, true>: the 7x7x7 variant)


def parse_entry(ent):
    """'workload[@envs][+rollout]' (tools/profile_all.sh) -> (workload, envs, rollout?)"""
    ro = ent.endswith("+rollout")
    we = ent[:-len("+rollout")] if ro else ent
    w, _, e = we.partition("@")
    return w, (int(e) if e else ENVS.get(w, 0)), ro


def compiler_resources():
    """kernel name (as tools/kernel_resources.py prints it) -> the code object's own figures; rocprofv3's trace columns give
    the architectural VGPRs and the STATIC LDS only"""
    res = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_resources.txt"))):
        for line in open(f):
            t = line.split()
            if len(t) >= 8 and t[1].isdigit():
                res[" ".join(t[7:])] = {"vgpr": int(t[1]), "agpr": int(t[2]), "sgpr": int(t[3]), "static_lds_bytes": int(t[4]),
                                        "scratch_bytes": int(t[5]), "source": os.path.basename(f)}
    return res


def launch_class(r):
    return (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0))


def summarize(ent, stats_rows, resources):
    out = {}
    w0, n_envs, rollout = parse_entry(ent)
    w = ent
    dom = ("rollout_kernel" if rollout and "3D" not in w0 else "m3_kernel" if "3D" in w0
           else "stats_for_grids_kernel" if "stats-for-grids" in w0 else "step_kernel")
    is_dom = (lambda name: dom in name and ((("m3_kernel<5" in name or "M3Mode)5" in name) if rollout else is_m3_step(name)) or dom != "m3_kernel"))
    cls = None
    out["entry"] = {"workload": w0, "envs": n_envs, "kernel_profiled": "open-loop rollout (pcgrl_rollout, 64 steps per launch)" if rollout else "step"}
    # three kernel traces per step entry (tools/profile_all.sh): the least disturbed one is reported, all three means are kept
    kt_dirs, run_means = [f"prof_{w}_kt"], {}
    for extra in (f"prof_{w}_ktrep2", f"prof_{w}_ktrep3"):
        if glob.glob(os.path.join(G, extra)):
            kt_dirs.append(extra)
    for d_ in kt_dirs:
        f_ = first(f"{d_}/**/*kernel_trace.csv")
        if f_:
            durs_ = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f_)) if is_dom(r["Kernel_Name"])]
            if durs_:
                run_means[d_] = statistics.mean(durs_)
    kt_dir = f"prof_{w}_kt"
    if len(run_means) >= 2:  # the run the profiler disturbed least: its per-dispatch overhead only ever adds
        kt_dir = min(run_means, key=run_means.get)
    out["kernel_trace_runs"] = {"dominant_kernel_mean_ns_per_run": [run_means[k] for k in sorted(run_means)], "reported_run": kt_dir,
                                "note": "rocprofv3's per-dispatch overhead differs from run to run (two modes, ~0.4 us apart on a 6 us kernel): "
                                        "all runs' means are listed; durations / stats below are those of the run with the smallest mean (the overhead only adds; "
                                        "bench.py's wall clock of the same launches, no profiler attached, reads 5.96-6.04 us where these read 6.0-6.5)"}
    ks = first(f"{kt_dir}/**/*kernel_stats.csv")
    if ks:
        rows = list(csv.DictReader(open(ks)))
        keep = [r for r in rows if "pcgrl" in r["Name"]]
        for r in keep:
            stats_rows.append(dict(r, Workload=w))
        out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                "pct": float(r["Percentage"])} for r in keep]
    kt = first(f"{kt_dir}/**/*kernel_trace.csv")
    if kt:
        rows = [r for r in csv.DictReader(open(kt)) if is_dom(r["Kernel_Name"])]
        # one launch class: sokoban launches a workgroup per env ("spread") while its solver is busy, a different kernel
        # geometry with different counters -- durations, geometry and PMC below all describe the most frequent class
        classes = defaultdict(int)
        for r in rows:
            classes[launch_class(r)] += 1
        cls = max(classes, key=classes.get) if classes else None
        out["launch_classes"] = {f"grid {g} x workgroup {wg}": n for (g, wg), n in sorted(classes.items())}
        rows = [r for r in rows if launch_class(r) == cls]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))  # (the trace is not always written in time order)
        if len(rows) > 2:
            gaps = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
            r0 = rows[0]
            durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
            periods = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(len(rows) - 1)]
            q = lambda f: durs[min(len(durs) - 1, int(f * len(durs)))]
            out["dominant_kernel_launch"] = {
                "kernel": short(r0["Kernel_Name"]), "launches": len(rows), "grid": int(r0["Grid_Size_X"]),
                "workgroup": int(r0["Workgroup_Size_X"]), "lds_bytes": int(r0.get("LDS_Block_Size", 0) or 0),
                "vgpr": int(r0.get("VGPR_Count", 0) or 0), "sgpr": int(r0.get("SGPR_Count", 0) or 0),
                "scratch_bytes": int(r0.get("Scratch_Size", 0) or 0),
                # every resource column the trace carries, verbatim (the meaning of VGPR_Count / LDS_Block_Size differs between
                # rocprofv3 versions: arch VGPRs only, static LDS only -- profiles/<tag>_kernel_resources.txt has the compiler's)
                "trace_resource_columns": {k: r0[k] for k in r0 if any(t in k for t in ("VGPR", "SGPR", "LDS", "Scratch", "Private", "Group"))},
                "median_gap_between_launches_ns": statistics.median(gaps),
                "duration_ns": {"p10": q(0.1), "median": q(0.5), "p90": q(0.9), "p99": q(0.99), "mean": statistics.mean(durs)},
                "median_start_to_start_ns": statistics.median(periods)}
            # the compiler's own figures for this kernel (template arguments as c++filt prints them)
            full = r0["Kernel_Name"].split("(pcgrl::Params")[0].replace("void ", "").strip()
            comp = resources.get(full)
            if comp:
                out["dominant_kernel_launch"]["compiler"] = comp
                out["dominant_kernel_launch"]["dynamic_lds_bytes"] = max(0, int(r0.get("LDS_Block_Size", 0) or 0) - comp["static_lds_bytes"]) or None
    counters = defaultdict(dict)
    for d in ("fetch", "write", "sq", "sq2"):
        f = first(f"prof_{w}_{d}/**/*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pcgrl" in r["Kernel_Name"]:
                if is_dom(r["Kernel_Name"]) and cls is not None and launch_class(r) != (0, 0) and launch_class(r) != cls:
                    continue  # (another launch class of the dominant kernel: not the one whose durations are reported)
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            counters[k][c] = {"mean_per_launch": statistics.mean(v), "launches": len(v)}
    out["pmc"] = counters
    step = next((k for k in counters if is_dom(k)), None)
    n = n_envs
    if step:
        c = counters[step]
        if "WRITE_SIZE" in c:
            # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x
            # on gfx950; WRITE_SIZE was calibrated in round 1 on observe_kernel (exactly 4096 * 3072 B written with the
            # step kernel's store pattern): factor 1.00.
            wr = c["WRITE_SIZE"]["mean_per_launch"] * 1024
            rd = c.get("FETCH_SIZE", {"mean_per_launch": 0})["mean_per_launch"] * 1024
            algo = int(bench.ALGO_BYTES[w0] * n) * (64 if rollout else 1)  # (a rollout launch = 64 steps, every observation written)
            out["hbm_traffic_per_launch"] = {"write_bytes": wr, "fetch_bytes_raw": rd, "fetch_bytes_x2_correction": 2 * rd,
                                             "traffic_bytes": wr + 2 * rd, "algorithmic_bytes": algo,
                                             "traffic_over_algorithmic": (wr + 2 * rd) / algo}
        if "SQ_WAVES" in c:
            waves = c["SQ_WAVES"]["mean_per_launch"]
            out["occupancy"] = {"waves_per_launch": waves, "waves_per_cu": waves / 256.0}
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] > 0:
            out["lds"] = {"bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"],
                          "lds_active_cycles": c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"],
                          "bank_conflict_rate": c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"],
                          "lds_instructions": c.get("SQ_INSTS_LDS", {}).get("mean_per_launch")}
    return out


def main(tag, outdir=None):
    outdir = outdir or os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    # kernel_sources_sha16: the kernels these passes ran on (bench.py echoes `traffic` / `kernel_mean_us` only from a summary
    # whose hash equals the checkout's); profiled_at_head: filled in by tools/finish_round.py, where the git history is
    out = {"tag": tag, "source": "rocprofv3 on python3 bench.py --workload W (BASELINE.json batch sizes, 1 MI355X); tools/profile_all.sh",
           "kernel_sources_sha16": bench.kernel_sources_sha16(), "profiled_at_head": None,
           "workloads": {}, "hbm_traffic_per_launch_by_workload": {}}
    stats_rows = []
    resources = compiler_resources()
    for d in sorted(glob.glob(os.path.join(G, "prof_*_kt"))):
        ent = os.path.basename(d)[5:-3]
        s = summarize(ent, stats_rows, resources)
        out["workloads"][ent] = s
        w0, n_envs, rollout = parse_entry(ent)
        if "hbm_traffic_per_launch" in s and not rollout:
            out["hbm_traffic_per_launch_by_workload"][f"{w0}@{n_envs}"] = s["hbm_traffic_per_launch"]
    if stats_rows:
        with open(os.path.join(outdir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w_ = csv.DictWriter(f, fieldnames=["Workload"] + [k for k in stats_rows[0].keys() if k != "Workload"])
            w_.writeheader()
            for r in stats_rows:
                w_.writerow(r)
    with open(os.path.join(outdir, f"{tag}_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:6000])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02", sys.argv[2] if len(sys.argv) > 2 else None)
