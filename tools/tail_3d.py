#!/usr/bin/env python3
"""Per-launch distribution of the 3-D kernel's wave lifetimes (phase-timing build): which env is the slowest and why."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib
_lib.LIB_PATH = os.path.join(_lib.CSRC, "libpcgrl_amd_timing.so")
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
n = 1024
env = VecPcgrlEnv("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=np.arange(n), auto_reset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, 2, (1021, n), generator=g, device="cuda", dtype=torch.int32)
sp = torch.cuda.current_stream().cuda_stream
out = np.zeros(8 * n, np.uint64)
names = ["trips", "entries", "n_searches", "search loops", "regions", "everything else", "candidate walk", "wall"]
rows = []
for k in range(1500):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
    if k >= 500:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
        a = out.reshape(n, 8).astype(np.float64)
        tot = a[:, 3:7].sum(1) - a[:, 3]  # regions + rest + candidate walk (which contains the search loops)
        i = int(tot.argmax())
        rows.append((tot.mean(), tot.max(), a[i, :8]))
    elif k == 499:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * n)
m = np.array([r[0] for r in rows]); mx = np.array([r[1] for r in rows]); ph = np.array([r[2] for r in rows])
print("mean wave cycles %.0f, mean of per-launch max %.0f (x%.1f)" % (m.mean(), mx.mean(), mx.mean() / m.mean()))
print("phases of the slowest wave (mean over launches):", dict(zip(names[:8], ph.mean(0).round().tolist())))

# distribution of the search work per launch (all waves)
print("per-wave means over the last launch: trips %.1f entries %.1f searches %.2f; max trips %d entries %d searches %d" % (
    a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 0].max(), a[:, 1].max(), a[:, 2].max()))

# the slowest wave of each launch, by how slow the launch was
order = np.argsort(mx)
for name, sel in (("median launches", order[len(order) // 2 - 10: len(order) // 2 + 10]), ("slowest 10 %", order[-len(order) // 10:]), ("slowest 2 %", order[-max(4, len(order) // 50):])):
    print(f"{name:16s} slowest wave: cycles {mx[sel].mean():8.0f}  " + "  ".join(f"{n} {ph[sel, i].mean():.0f}" for i, n in enumerate(names[:7])))
