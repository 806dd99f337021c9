#!/usr/bin/env python3
"""How much solver work does the solver-active sokoban workload of bench.py carry?  Replays it on the CPU oracle and prints
the oracle's development counters (oracle/sokoban_solver.c orc_solver_hist): calls, pops per stage, outcome, histogram of
pops per call.  CPU only."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pcgrl_oracle as po
import bench

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 64
maps, cells = bench.solver_active_maps(n, 77)
acts = bench.solver_active_actions(cells, 64, 1234)
env = po.OracleVecEnv("sokoban", "wide", (16, 16), n, seeds=np.arange(n), threads=8)
hist = (C.c_longlong * 40).in_dll(po.lib(), "orc_solver_hist")
t0 = time.time()
per_step = []
for t in range(steps):
    if t % 8 == 0:
        env.reset(init_grids=maps)
    before = list(hist)
    env.step(acts[t % 64], want_obs=False)
    after = list(hist)
    per_step.append([a - b for a, b in zip(after, before)])
dt = time.time() - t0
h = np.array(list(hist), np.int64)
print(f"{steps} steps x {n} envs in {dt:.1f} s on 8 threads")
print("solver calls %d (%.1f per step); BFS pops %d (%.0f per call), wins %d, exhausted %d" % (h[0], h[0] / steps, h[1], h[1] / max(h[0], 1), h[2], h[3]))
print("A* stages %d, pops %d (%.0f per stage), wins %d, exhausted %d" % (h[4], h[5], h[5] / max(h[4], 1), h[6], h[7]))
print("pops per call histogram [2^b, 2^(b+1)):", {b: int(h[8 + b]) for b in range(20) if h[8 + b]})
ps = np.array(per_step)
print("per step: calls mean %.1f max %d; total pops mean %.0f max %d; A* stages per step mean %.2f max %d" % (
    ps[:, 0].mean(), ps[:, 0].max(), (ps[:, 1] + ps[:, 5]).mean(), (ps[:, 1] + ps[:, 5]).max(), ps[:, 4].mean(), ps[:, 4].max()))
