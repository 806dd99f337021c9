#!/usr/bin/env python3
"""Per-level cost of the solver-active workload of bench.py: its playable 16 x 16 levels through pcgrl_stats_for_grids, all
at once (one level per workgroup: the launch = the slowest level + dispatch) and one by one (development probe: is the
step launch of that workload the cost of its slowest level?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from control_pcgrl_amd import VecPcgrlEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
maps, cells = bench.solver_active_maps(n, 77)
env = VecPcgrlEnv("sokoban", "wide", (16, 16), 1, auto_reset=False)
g = torch.as_tensor(maps).to(env.device)
env.stats_for_grids(g[:2]); torch.cuda.synchronize()
for rep in range(3):  # (the first call grows the engine's solver workspace pool for this batch size)
    t0 = time.perf_counter(); st = env.stats_for_grids(g); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} levels in one launch: {dt * 1e3:.1f} ms")
per = []
for i in range(n):
    t0 = time.perf_counter(); env.stats_for_grids(g[i:i + 1]); torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
per = np.array(per) * 1e3
top = np.argsort(-per)[:8]
print("slowest levels (ms, crates):", [(int(i), round(float(per[i]), 1), int((maps[i] == 3).sum())) for i in top])
print("levels over 5 ms: %d, over 15 ms: %d; sum of all %.0f ms; median %.3f ms" % ((per > 5).sum(), (per > 15).sum(), per.sum(), np.median(per)))
