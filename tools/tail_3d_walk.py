#!/usr/bin/env python3
"""What the slowest wave of a 3-D step launch does (timing build with -DPCGRL_M3_TAIL:  python tools/phase_timing.py --build
--m3-tail  here, then this script on the GPU box): per launch the simulate wave with the longest lifetime, how many cached
start planes it found missing, how many pairs of searches it ran, how often the speculative second search was used, and how
its cycles split into the candidate walk and everything else.  (The "runner" columns belong to the round-5 experiment with a
fourth wavefront per env, commit bb5a491: zero with the shipped kernels.  profiles/r05_dev_traces.md has both outputs.)"""
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
names = ["missing", "own pairs", "runner pairs", "walk cyc", "wait cyc", "spec used", "rest cyc", "wall"]
rows, allw = [], []
for k in range(2500):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
    if k >= 500:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
        a = out.reshape(n, 8).astype(np.float64)
        tot = a[:, 3] + a[:, 6]
        i = int(tot.argmax())
        rows.append(np.concatenate([a[i], [tot[i], tot.mean()]]))
        allw.append(a.copy())
    elif k == 499:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
r = np.array(rows)
A = np.concatenate(allw)
print(f"{len(r)} launches x {n} envs; mean wave {r[:, 9].mean():.0f} cycles, mean per-launch max {r[:, 8].mean():.0f} (x{r[:, 8].mean() / r[:, 9].mean():.1f})")
print("all waves: fraction with >= 1 missing slot %.4f, >= 2 %.4f, >= 3 %.4f; pairs per wave-step: own %.4f runner %.4f" % (
    (A[:, 0] >= 1).mean(), (A[:, 0] >= 2).mean(), (A[:, 0] >= 3).mean(), A[:, 1].mean(), A[:, 2].mean()))
print("slowest wave of a launch: missing slots  0: %.3f  1: %.3f  2: %.3f  >=3: %.3f" % tuple(
    [(r[:, 0] == k).mean() for k in (0, 1, 2)] + [(r[:, 0] >= 3).mean()]))
print("slowest wave: pairs it ran itself 0: %.3f 1: %.3f 2: %.3f >=3: %.3f; with >= 1 runner pair: %.3f; spec used (per own pair): %.3f" % tuple(
    [(r[:, 1] == k).mean() for k in (0, 1, 2)] + [(r[:, 1] >= 3).mean(), (r[:, 2] >= 1).mean(), r[:, 5].sum() / max(r[:, 1].sum(), 1)]))
order = np.argsort(r[:, 8])
for name, sel in (("all launches", order), ("median +-5 %", order[int(len(order) * .45): int(len(order) * .55)]), ("slowest 10 %", order[-len(order) // 10:])):
    m = r[sel].mean(0)
    print(f"{name:14s} slowest wave: total {m[8]:8.0f} = walk {m[3]:8.0f} (waiting for the runner {m[4]:6.0f}) + rest {m[6]:7.0f}; missing {m[0]:.2f} own pairs {m[1]:.2f} runner pairs {m[2]:.2f} spec used {m[5]:.2f}")
