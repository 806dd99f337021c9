#!/usr/bin/env python3
"""Collect a round's measurements into the tracked files:
  gpurun_out/bench_<tag>_*.log         -> profiles/<tag>_bench_lines.json
  gpurun_out/summary/<tag>_*           -> profiles/
and regenerate the measurement tables of DESIGN.md / README.md between their <!-- NAME --> ... <!-- /NAME --> markers.
  python tools/finish_round.py r02"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
short = tag.replace("r0", "r")  # bench logs are named bench_r2_*.log
G = os.path.join(ROOT, "gpurun_out")

lines = {}
for f in sorted(glob.glob(os.path.join(G, f"bench_{short}_*.log"))):
    name = os.path.basename(f)[len(f"bench_{short}_"):-4]
    txt = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if txt:
        lines[name] = json.loads(txt[-1])
with open(os.path.join(ROOT, "profiles", f"{tag}_bench_lines.json"), "w") as fh:
    json.dump(lines, fh, indent=1)
for f in glob.glob(os.path.join(G, "summary", f"{tag}_*")):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
summ = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_summary.json")))
# the commit the passes belong to: HEAD, if the kernels of this checkout are the ones that were profiled
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import subprocess  # noqa: E402
if summ.get("kernel_sources_sha16") == bench.kernel_sources_sha16() and not summ.get("profiled_at_head"):  # (both families)
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "control_pcgrl_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    summ["profiled_at_head"] = head + (" + uncommitted kernel changes" if dirty else "")
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)


def fmt(x, d=2):
    return f"{x:.{d}f}"


def sci(x):
    e = int(f"{x:e}".split("e")[1])
    return f"{x / 10 ** e:.2f} × 10{str(e).translate(str.maketrans('0123456789-', '⁰¹²³⁴⁵⁶⁷⁸⁹⁻'))}"


# workloads whose kernels were not re-profiled this round (they did not change): the previous round's summary, marked with a dagger
prev = {}
try:
    prev = json.load(open(os.path.join(ROOT, "profiles", f"r{int(tag[1:]) - 1:02d}_summary.json")))["workloads"]
except Exception:
    pass
daggers = []


def wl(w):
    if w in summ["workloads"]:
        return summ["workloads"][w]
    if w in prev:
        if w not in daggers:
            daggers.append(w)
        return prev[w]
    return {}


def kernel_us(w):
    d = wl(w).get("dominant_kernel_launch")
    return d["duration_ns"] if d else None


b = lines["binary-narrow"]
ws = summ["workloads"]["binary-narrow"]
dk = ws["dominant_kernel_launch"]["duration_ns"]
tr = ws["hbm_traffic_per_launch"]
sq = next(v for k, v in ws["pmc"].items() if "step_kernel" in k)
eager = lines.get("binary-narrow-eager")
big = lines.get("binary-narrow-65536")
headline = f"""| quantity | value | source |
|---|---|---|
| env-steps/s | **{sci(b['value'])}** (north-star target 1 × 10⁷) | `bench.py`, wall clock |
| per step | {fmt(b['ms_per_step'] * 1e3)} µs wall = {fmt(b['roofline']['avg_launch_us'])} µs by HIP events on the launch stream""" + (f"; eager launches (`pcgrl_step_seq`): {fmt(eager['ms_per_step'] * 1e3)} µs" if eager else "") + f""" | `bench.py` |
| step kernel, begin–end under `rocprofv3 --kernel-trace` | mean {fmt(dk['mean'] / 1e3)} µs, median {fmt(dk['median'] / 1e3)}, p10 {fmt(dk['p10'] / 1e3)}, p99 {fmt(dk['p99'] / 1e3)} ({ws['dominant_kernel_launch']['launches']} launches; start-to-start {fmt(ws['dominant_kernel_launch']['median_start_to_start_ns'] / 1e3)} µs) | `profiles/{tag}_kernel_stats.csv`, `profiles/{tag}_summary.json` |
| roofline | {b['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic / {fmt(b['ms_per_step'] * 1e3)} µs = {b['roofline']['achieved'] / 1e3:.2f} TB/s = **{fmt(b['roofline']['frac'], 3)} of 8 TB/s** (from the profiled kernel mean: {fmt(b['roofline']['algorithmic_bytes_per_launch'] / (dk['mean'] * 1e-9) / 8e12, 3)}) | `bench.py` |
| HBM traffic | WRITE_SIZE {tr['write_bytes'] / 1e6:.1f} MB + 2×FETCH_SIZE {tr['fetch_bytes_x2_correction'] / 1e6:.1f} MB = {tr['traffic_bytes'] / 1e6:.1f} MB / launch = {fmt(tr['traffic_over_algorithmic'])} × algorithmic | `profiles/{tag}_summary.json` |
| waves / LDS | {ws['occupancy']['waves_per_launch']:.0f} waves per launch ({ws['occupancy']['waves_per_cu']:.0f} per CU); {sq['SQ_INSTS_VALU']['mean_per_launch'] / 1e6:.2f} M VALU, {sq['SQ_INSTS_SALU']['mean_per_launch'] / 1e6:.2f} M SALU, {sq['SQ_INSTS_LDS']['mean_per_launch'] / 1e3:.0f} k LDS instructions; LDS bank-conflict cycles / LDS active cycles = {fmt(ws['lds']['bank_conflict_rate'])} (the byte-granular one-hot scatter into the observation rows) | `profiles/{tag}_summary.json` |
""" + (f"| device fill of the same {b['roofline']['fill_same_bytes']['bytes'] / 1e6:.1f} MB, same run (context: what a write-only kernel of this size reaches) | {fmt(b['roofline']['fill_same_bytes']['us'])} µs = {fmt(b['roofline']['fill_same_bytes']['frac_of_peak'], 2)} of 8 TB/s; the step launch takes {fmt(b['roofline']['fill_same_bytes']['step_over_fill'])} × that | `bench.py` `roofline.fill_same_bytes` |\n" if b['roofline'].get('fill_same_bytes') else "") + (f"| same kernel, 65 536 envs | {fmt(big['ms_per_step'] * 1e3, 1)} µs ⇒ **{sci(big['value'])} steps/s, {fmt(big['roofline']['frac'], 2)} of 8 TB/s**" + (f" (a fill of the same {big['roofline']['fill_same_bytes']['bytes'] / 1e6:.0f} MB: {fmt(big['roofline']['fill_same_bytes']['us'], 1)} µs = {fmt(big['roofline']['fill_same_bytes']['frac_of_peak'], 2)})" if big['roofline'].get('fill_same_bytes') else "") + f" | `bench.py --envs 65536` |\n" if big else "") + f"""| open-loop rollout (64 steps per launch, every observation written) | {fmt(b['open_loop_rollout']['us_per_step'])} µs per step, {sci(b['open_loop_rollout']['value'])} steps/s, {fmt(b['open_loop_rollout']['roofline_frac'], 2)} of the roofline | `bench.py` `open_loop_rollout` |
| CPU oracle (C port, OpenMP, {b['cpu_baseline']['cores']} threads of {b['cpu_baseline']['usable_cores']} usable cores, {b['cpu_baseline']['cpu_model']}) | {sci(b['cpu_baseline']['value'])} steps/s ({sci(b['cpu_baseline']['one_core'])} on one core) | `bench.py` `cpu_baseline` |
| reference Python env (survey probe, 1 core) | ≈ 750 steps/s | BASELINE.md |"""

d20 = lines.get("driver_20_5")
driver20 = (f"{fmt(d20['ms_per_step'] * 1e3)} µs per step, {sci(d20['value'])} env-steps/s, {fmt(d20['ms_per_step'] / b['ms_per_step'])} × the long run"
            if d20 else "n/a")

NOTES = {
    "binary-narrow": "headline",
    "zelda-turtle": "store-bound: 37.7 MB of observations per launch",
    "zelda-turtle-bfs": "SURVEY §8 d \"BFS-active\": maps with exactly one player / key / door re-injected every 128 steps, actions = moves and empty / solid / enemy placements (both single-source searches run); incl. the injection launches",
    "sokoban-wide": "compile-time 16×16 wide kernel; rare solver calls after resets (dense random levels) are in the mean",
    "minecraft_3D_maze-narrow": "the launch waits for the env whose pair of searches has the largest hop depth (§4.2); round 2: 34.6 µs",
    "binary-narrow-static": "static tiles (p ≤ 0.3, 3 walls), general kernel",
    "binary-narrow-patch3x3": "3×3 action patch, general kernel",
    "sokoban-wide-solver": "solver-active, see below",
    "binary_big-narrow": "the reference's `binary_big` task (32², window 64²; `configs/task/binary_big.yaml:5-6`): one-hot rows in LDS (13 KB per workgroup)",
    "binary_bigger-narrow": "`binary_bigger` (64², window 128²; `binary_bigger.yaml:5-6`): 64-bit row masks, observation chunks from tile codes; the launch ends with the env whose path search is longest",
    "zelda_big-turtle": "`zelda_big` (32², window 64²; `zelda_big.yaml:5-6`): observation chunks from tile codes (§4.1)",
    "zelda_bigger-turtle": "`zelda_bigger` (64², window 128²; `zelda_bigger.yaml:5-6`): 621 MB of observations per launch, 64-bit row masks, chunks from tile codes",
    "minecraft_3D_maze-narrow-15": "the reference's stock 3-D map (`configs/config.py:153-157`), two whole episodes: one workgroup per CU (147 KB of LDS), three observe waves",
}
rows = ["| workload | envs/GPU | env-steps/s | µs per step (wall) | kernel mean / median µs (rocprofv3) | roofline frac (wall / kernel mean) | same-size fill µs (step ÷ fill) | rollout µs/step | closed loop µs/step | CPU oracle steps/s (threads) | note |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for w in ("binary-narrow", "zelda-turtle", "sokoban-wide", "minecraft_3D_maze-narrow", "binary_big-narrow", "binary_bigger-narrow", "zelda_big-turtle",
          "zelda_bigger-turtle", "minecraft_3D_maze-narrow-15", "zelda-turtle-bfs", "binary-narrow-static", "binary-narrow-patch3x3", "sokoban-wide-solver"):
    if w not in lines:
        continue
    l = lines[w]
    k = kernel_us(w) if w != "sokoban-wide-solver" else None  # (the profiled solver-active entry is the ASYNCHRONOUS kernel: next row)
    ks = (f"{k['mean'] / 1e3:.2f} / {k['median'] / 1e3:.2f}" + (" †" if w in daggers else "")) if k else "–"
    ro = l.get("open_loop_rollout")
    cb = l.get("cpu_baseline")
    kf = (l['roofline']['algorithmic_bytes_per_launch'] / (k['mean'] * 1e-9) / 8e12) if k else None
    rows.append(f"| {w} | {l['config']['envs_per_gpu']} | {sci(l['value'])} | {l['ms_per_step'] * 1e3:.2f} | {ks} | {l['roofline']['frac']:.3f}" + (f" / {kf:.3f}" if kf else "") + " | "
                + (f"{l['roofline']['fill_same_bytes']['us']:.2f} ({l['roofline']['fill_same_bytes']['step_over_fill']:.2f} ×)" if l['roofline'].get('fill_same_bytes') else "–") + " | "
                + (f"{ro['us_per_step']:.2f}" if ro else "–") + " | "
                + (f"{l['closed_loop_device_actions']['us_per_step']:.2f}" if (l.get('closed_loop_device_actions') or {}).get('us_per_step') else "–") + " | "
                + (f"{sci(cb['value'])} ({cb['cores']})" if cb else "–") + f" | {NOTES.get(w, '')} |")
la = lines.get("sokoban-wide-solver-async16")
if la:
    k = kernel_us("sokoban-wide-solver")
    cb = la.get("cpu_baseline")
    rows.append(f"| sokoban-wide-solver, asynchronous stepping (budget 16) | {la['config']['envs_per_gpu']} | {sci(la['value'])} emitted | {la['ms_per_step'] * 1e3:.2f} per launch | "
                + (f"{k['mean'] / 1e3:.2f} / {k['median'] / 1e3:.2f}" if k else "–") + f" | {la['roofline']['frac']:.3f} | – | – | – | "
                + (f"{sci(cb['value'])} ({cb['cores']})" if cb else "–") + " | `pcgrl_step_ready`, §4.1a: busy envs do not step; the HBM roof is the wrong ruler for a search-bound launch |")
extra = []
for w in ("zelda-turtle", "sokoban-wide", "minecraft_3D_maze-narrow", "binary_big-narrow", "binary_bigger-narrow", "zelda_big-turtle", "zelda_bigger-turtle",
          "minecraft_3D_maze-narrow-15", "binary-narrow-static", "binary-narrow-patch3x3", "zelda-turtle-bfs", "binary-narrow-evo", "binary-stats-for-grids",
          "zelda-stats-for-grids", "binary-narrow+rollout", "binary-narrow@16384", "binary-narrow@65536", "zelda-turtle@16384", "zelda-turtle@65536",
          "sokoban-wide@8192", "sokoban-wide@32768", "minecraft_3D_maze-narrow@4096", "minecraft_3D_maze-narrow@16384"):
    s_ = wl(w)
    if "hbm_traffic_per_launch" in s_ and "lds" in s_:
        extra.append(f"{w}{' †' if w in daggers else ''}: traffic {s_['hbm_traffic_per_launch']['traffic_bytes'] / 1e6:.1f} MB / launch = {s_['hbm_traffic_per_launch']['traffic_over_algorithmic']:.2f} × algorithmic, "
                     f"{s_['occupancy']['waves_per_cu']:.0f} waves per CU, LDS bank-conflict rate {s_['lds']['bank_conflict_rate']:.2f}")
workload_table = ("\n".join(rows) + "\n\nCounters (`profiles/" + tag + "_summary.json`): " + "; ".join(extra) + "."
                  + ("\n\n† kernel unchanged since the previous round and not re-profiled: figures of `profiles/r%02d_summary.json`." % (int(tag[1:]) - 1) if daggers else ""))

sa = lines.get("sokoban-wide-solver")
solver_active = ""
if sa:
    cb = sa.get("cpu_baseline") or {}
    solver_active = (f"Engine: {sci(sa['value'])} env-steps/s ({sa['ms_per_step']:.0f} ms per step launch at 2048 envs, "
                     f"{100 * sa['solver_active']['envs_with_solver_result_at_end']:.0f} % of the envs with a solver result at the end, "
                     f"{100 * sa['solver_active']['envs_solved_at_end']:.0f} % solved); CPU oracle: {sci(cb.get('value', 0))} on {cb.get('cores', '?')} threads.")

# asynchronous stepping on the solver-active workload: one row per solver budget
async_rows = ["| solver budget (units per env and launch) | µs per launch | emitted env-steps/s | env-launches that emitted | env-launches busy | vs synchronous `pcgrl_step` | vs the CPU oracle |", "|---|---|---|---|---|---|---|"]
cpu_sa = ((lines.get("sokoban-wide-solver-async16") or {}).get("cpu_baseline") or (sa or {}).get("cpu_baseline") or {}).get("value")
for B in (4, 8, 16, 32, 64, 256):
    l = lines.get(f"sokoban-wide-solver-async{B}")
    if not l or "asynchronous_stepping" not in l:
        continue
    a_ = l["asynchronous_stepping"]
    async_rows.append(f"| {B} | {a_['us_per_launch']:.1f} | {sci(l['value'])} | {a_['emitted_share_of_env_launches']:.3f} | {a_['busy_share_of_env_launches']:.3f} | "
                      + (f"{l['value'] / sa['value']:.0f} ×" if sa else "–") + " | " + (f"{l['value'] / cpu_sa:.1f} ×" if cpu_sa else "–") + " |")
async_table = "\n".join(async_rows)
forms = (sa or {}).get("solver_active_forms") or {}
if forms.get("pcgrl_rollout"):
    solver_active += (f"  The same workload through the forms that shrink a launch's synchronisation domain without a ready mask: `pcgrl_rollout` "
                      f"({forms['pcgrl_rollout']['steps_per_launch']} steps per launch between re-injections) {sci(forms['pcgrl_rollout']['value'])} env-steps/s "
                      f"({forms['pcgrl_rollout']['us_per_step'] / 1e3:.0f} ms per step), `sub_batches = 4` {sci(forms['sub_batches_4']['value'])} "
                      f"({forms['sub_batches_4']['us_per_step'] / 1e3:.0f} ms per step): both SLOWER than plain stepping — a rollout wave pays the sum of its env's "
                      f"searches over the steps, and four engines split the solver's workspace pool and the helper waves' LDS.")

# saturation sweeps: the BASELINE batch x1 / x4 / x16
sweep_rows = ["| workload | envs/GPU | env-steps/s | µs per step launch | roofline frac | same-size fill µs (step ÷ fill) | kernel mean / median µs (rocprofv3) |", "|---|---|---|---|---|---|---|"]
for w, sizes in (("binary-narrow", ("", "-16384", "-65536", "-262144")), ("zelda-turtle", ("", "-16384", "-65536")), ("sokoban-wide", ("", "-8192", "-32768")),
                 ("minecraft_3D_maze-narrow", ("", "-4096", "-16384"))):
    for sfx in sizes:
        l = lines.get(w + sfx)
        if not l:
            continue
        f_ = l["roofline"].get("fill_same_bytes")
        kk = kernel_us(w + ("@" + sfx[1:] if sfx else ""))
        dg = " †" if (w + ("@" + sfx[1:] if sfx else "")) in daggers else ""
        sweep_rows.append(f"| {w} | {l['config']['envs_per_gpu']} | {sci(l['value'])} | {l['ms_per_step'] * 1e3:.2f} | {l['roofline']['frac']:.3f} | "
                          + (f"{f_['us']:.2f} ({f_['step_over_fill']:.2f} ×)" if f_ else "–") + " | " + (f"{kk['mean'] / 1e3:.2f} / {kk['median'] / 1e3:.2f}{dg}" if kk else "–") + " |")
sweep_table = "\n".join(sweep_rows)

# evolution-driver pattern
evo_rows = ["| workload | units per launch | throughput | µs per launch | roofline frac | CPU oracle (threads) |", "|---|---|---|---|---|---|"]
for w, what in (("binary-narrow-evo", "4096 envs: `pcgrl_update` (+ observation) per step, `pcgrl_refresh_stats` every 256 steps"),
                ("binary-stats-for-grids", "65 536 maps per `pcgrl_stats_for_grids_h` launch"), ("zelda-stats-for-grids", "65 536 maps per launch")):
    l = lines.get(w)
    if not l:
        continue
    cb = l.get("cpu_baseline") or {}
    evo_rows.append(f"| {w}: {what} | {l['config']['envs_per_gpu']} | {sci(l['value'])} {l['unit']} | {l['ms_per_step'] * 1e3:.2f} | {l['roofline']['frac']:.3f} | "
                    + (f"{sci(cb['value'])} {cb.get('unit', '')} ({cb['cores']})" if cb else "–") + " |")
evo_table = "\n".join(evo_rows)

# RLlib-shaped adapter
ad = (lines.get("binary-narrow") or {}).get("rllib_adapter")
base = None
try:
    base = {r["envs"]: r for r in json.load(open(os.path.join(ROOT, "profiles", "r04_rllib_adapter_r3_baseline.json")))["rows"]}
except Exception:
    pass
ad_rows = ["| envs | hand-out | env-steps/s | host µs per `vector_step` | device→host bytes per call | vs the round-3 adapter (float32) |", "|---|---|---|---|---|---|"]
if ad:
    for r in ad["rows"]:
        b0 = base.get(r["envs"]) if base else None
        mode = r["obs_dtype"] + (", kernel writes pinned host memory" if r["kernel_writes_host_memory"] else ", one copy")
        ad_rows.append(f"| {r['envs']} | {mode} | {sci(r['env_steps_per_s'])} | {r['host_us_per_vector_step']:.1f} | {r['d2h_bytes_per_call']} | "
                       + (f"{r['env_steps_per_s'] / b0['env_steps_per_s']:.2f} × ({sci(b0['env_steps_per_s'])})" if b0 else "–") + " |")
adapter_table = "\n".join(ad_rows)

# the driver's --steps 20 --warmup 5 under the short-run protocols, and through the collective path
drv_rows = ["| protocol | µs per step (wall = `value`'s clock) | timed region µs | K launches by HIP events µs | closing exchange µs | env-steps/s |", "|---|---|---|---|---|---|"]
for key, what in (("driver_20_5", "fused (default): one graph = 20 step launches + the reduction launch"),
                  ("driver_20_5_run2", "the same, second run"), ("driver_20_5_run3", "the same, third run"),
                  ("driver_20_5_nccl1", "N > 1 region with one rank (`--force-collective`): the same graph; the previous interval's RCCL all-gather + device→host copy on a side stream (round 6)"),
                  ("driver_20_5_nccl1_run2", "the same, second run"), ("driver_20_5_nccl1_run3", "the same, third run"),
                  ("driver_20_5_nccl1_serial", "round 5's N > 1 region (`--exchange serial`): reduction → all-gather → copy behind the launches"),
                  ("driver_20_5_one", "round 4: one graph of 20 steps, reduction launched separately"),
                  ("driver_20_5_gcd", "graph of gcd(5, 20) = 5 steps: 1 untimed + 4 timed replays"),
                  ("driver_20_5_eager", "20 eager launches (`pcgrl_step_seq`)")):
    l = lines.get(key)
    if not l:
        continue
    tr = l.get("timed_region", {})
    drv_rows.append(f"| {what} | {l['ms_per_step'] * 1e3:.2f} | {tr.get('wall_ms', 0) * 1e3:.1f} | {tr.get('launches_ms', 0) * 1e3:.1f} | {tr.get('exchange_ms', 0) * 1e3:.1f} | {sci(l['value'])} |")
driver_table = "\n".join(drv_rows)

# sub-batch chains (bench.py async_sub_batches)
sb_rows = ["| workload | envs/GPU | one batch µs/step (same protocol) | k = 2 | k = 4 | best |", "|---|---|---|---|---|---|"]
for w in ("binary-narrow", "zelda-turtle", "sokoban-wide", "minecraft_3D_maze-narrow", "binary_big-narrow", "binary_bigger-narrow", "zelda_big-turtle",
          "zelda_bigger-turtle", "minecraft_3D_maze-narrow-15"):
    sb = (lines.get(w) or {}).get("async_sub_batches")
    if not sb or "rows" not in sb:
        continue
    r_ = {r["sub_batches"]: r for r in sb["rows"]}
    best = max(r_.values(), key=lambda r: r["speedup_vs_one_batch_same_protocol"])
    sb_rows.append(f"| {w} | {lines[w]['config']['envs_per_gpu']} | {r_[1]['us_per_step_of_whole_batch']:.2f} | "
                   + " | ".join((f"{r_[k]['us_per_step_of_whole_batch']:.2f} ({r_[k]['speedup_vs_one_batch_same_protocol']:.2f} ×)" if k in r_ else "–") for k in (2, 4))
                   + f" | k = {best['sub_batches']}: {best['speedup_vs_one_batch_same_protocol']:.2f} × |")
subbatch_table = "\n".join(sb_rows)

cl = (lines.get("binary-narrow") or {}).get("closed_loop_device_actions") or {}
closed_loop = (f"{cl['us_per_step']:.2f} µs per step = {sci(cl['value'])} env-steps/s, {cl['roofline_frac']:.3f} of the roofline "
               f"({cl['us_per_step'] - b['ms_per_step'] * 1e3:.2f} µs more than the pool figure: the sampler kernel and one more kernel boundary per step)") if cl.get("us_per_step") else "n/a"

blocks = {"ASYNC_TABLE": async_table, "DRIVER_TABLE": driver_table, "SUBBATCH_TABLE": subbatch_table, "CLOSED_LOOP": closed_loop, "HEADLINE_TABLE": headline, "DRIVER20": driver20, "WORKLOAD_TABLE": workload_table, "SOLVER_ACTIVE": solver_active,
          "SWEEP_TABLE": sweep_table, "EVO_TABLE": evo_table, "ADAPTER_TABLE": adapter_table}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for name, text in blocks.items():
    inline = name in ("DRIVER20", "SOLVER_ACTIVE", "CLOSED_LOOP")
    new = f"<!-- {name} -->{'' if inline else chr(10)}{text}{'' if inline else chr(10)}<!-- /{name} -->"
    if f"<!-- {name} -->" not in s and f"@@{name}@@" not in s:
        continue
    if f"@@{name}@@" in s:
        s = s.replace(f"@@{name}@@", new)
    else:
        s = re.sub(rf"<!-- {name} -->.*?<!-- /{name} -->", lambda m: new, s, flags=re.S)
open(p, "w").write(s)
print(headline)
print(workload_table)
print(sweep_table)
print(evo_table)
print(adapter_table)
print(driver20)
print(driver_table)
print(subbatch_table)
print(closed_loop)
print(solver_active)
