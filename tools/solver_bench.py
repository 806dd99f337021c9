#!/usr/bin/env python3
"""Time the device Sokoban solver on the known-answer levels of tests/golden/stats_sokoban_solver.npz (one launch,
one lane per level) and check the answers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
z = np.load(os.path.join(ROOT, "tests", "golden", "stats_sokoban_solver.npz"))
env = VecPcgrlEnv("sokoban", "narrow", z["grids"].shape[1:], 1, auto_reset=False)
g = torch.as_tensor(z["grids"]).to(env.device)
env.stats_for_grids(g[:2]); torch.cuda.synchronize()
t0 = time.perf_counter(); got = env.stats_for_grids(g); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ok = np.array_equal(got.cpu().numpy(), z["stats"])
print(f"{len(g)} levels: {dt * 1e3:.1f} ms, answers {'match' if ok else 'DIFFER'}")
per = []
for i in range(len(g)):
    t0 = time.perf_counter(); env.stats_for_grids(g[i:i + 1]); torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
per = np.array(per) * 1e3
top = np.argsort(-per)[:5]
print("slowest levels (ms):", [(int(i), round(float(per[i]), 1), int((z['grids'][i] == 3).sum())) for i in top], "sum %.0f ms" % per.sum())
