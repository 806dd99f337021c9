#!/usr/bin/env python3
"""Phases of the slowest simulate wave of each 3-D step launch (timing build:  python tools/phase_timing.py --build --m3-phases
here, then this script on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib
_lib.LIB_PATH = os.path.join(_lib.CSRC, "libpcgrl_amd_timing.so")
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
n = int(next((a[7:] for a in sys.argv if a.startswith("--envs=")), "1024"))
env = VecPcgrlEnv("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=np.arange(n), auto_reset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, 2, (1021, n), generator=g, device="cuda", dtype=torch.int32)
sp = torch.cuda.current_stream().cuda_stream
out = np.zeros(8 * n, np.uint64)
names = ["loads", "edit+moves", "regions", "walk", "overlay", "outputs+writeback", "fresh", "wall(10ns)"]
if "--headsplit" in sys.argv:  # timing build with --m3-headsplit
    names = ["loads+fill", "edit", "moves+slot-drop", "walk", "position", "region-job+planes", "after-walk", "wall(10ns)"]
if "--tailsplit" in sys.argv:  # timing build with --m3-tailsplit
    names = ["write-back", "before-walk", "wait-regions", "walk", "overlay", "loss+outputs", "barrier", "wall(10ns)"]
rows, means = [], []
for k in range(2500):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
    if k >= 500:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
        a = out.reshape(n, 8).astype(np.float64)
        tot = a[:, :7].sum(1)
        i = int(tot.argmax())
        rows.append(np.concatenate([a[i], [tot[i]]]))
        means.append(np.concatenate([a.mean(0), [tot.mean()]]))
    elif k == 499:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
r, m = np.array(rows), np.array(means)
print("mean wave     :", "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names + ["total"], m.mean(0))))
order = np.argsort(r[:, 8])
for name, sel in (("all launches", order), ("median +-5 %", order[int(len(order) * .45): int(len(order) * .55)]), ("slowest 10 %", order[-len(order) // 10:])):
    print(f"{name:14s}:", "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names + ["total"], r[sel].mean(0))))
