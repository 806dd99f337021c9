#!/usr/bin/env python3
"""Per-launch distribution of the headline kernel's simulate-wave lifetimes (phase-timing build, tools/phase_timing.py
--build): phases of the slowest wave of each launch vs. the mean wave."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib
_lib.LIB_PATH = os.path.join(_lib.CSRC, "libpcgrl_amd_timing.so")
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
n = 4096
env = VecPcgrlEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, 2, (1021, n), generator=g, device="cuda", dtype=torch.int32)
sp = torch.cuda.current_stream().cuda_stream
blocks = n // 4
out = np.zeros(8 * blocks, np.uint64)
names = ["loads+barrier", "action+state", "stats rest", "flood", "first sweeps", "second sweep", "write-back"]
rows = []
for k in range(800):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
    if k >= 500:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * blocks)
        a = out.reshape(blocks, 8).astype(np.float64)
        tot = a[:, :7].sum(1)
        i = int(tot.argmax())
        rows.append((tot.mean(), tot.max(), a[i, :7], a[:, :7].mean(0), np.quantile(tot, 0.95)))
    elif k == 499:
        env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * blocks)
m = np.array([r[0] for r in rows]); mx = np.array([r[1] for r in rows])
print("mean wave %.0f cycles, p95 %.0f, mean of per-launch max %.0f (x%.2f)" % (m.mean(), np.mean([r[4] for r in rows]), mx.mean(), mx.mean() / m.mean()))
print("slowest wave :", dict(zip(names, np.array([r[2] for r in rows]).mean(0).round().tolist())))
print("mean wave    :", dict(zip(names, np.array([r[3] for r in rows]).mean(0).round().tolist())))
