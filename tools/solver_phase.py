#!/usr/bin/env python3
"""Phase breakdown of the device Sokoban solver's search iterations (shader-clock cycles per iteration, BFS and A*
stages apart) from a library built with -DPCGRL_SK_TIMING.
Build here (hipcc cross-compiles):  python tools/solver_phase.py --build
then on the GPU box:                python tools/solver_phase.py [n_levels]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib

TIMING_LIB = os.path.join(_lib.CSRC, "libpcgrl_amd_sktiming.so")
if "--build" in sys.argv:
    _lib.build(force=True, out=TIMING_LIB, defines=("PCGRL_SK_TIMING",))
    print("built", TIMING_LIB)
    sys.exit(0)

import numpy as np
import torch

PLAIN = "--plain" in sys.argv  # the shipped library (no counters): for rocprofv3 --pmc runs of the same launch
if PLAIN:
    sys.argv.remove("--plain")
else:
    _lib.LIB_PATH = TIMING_LIB
from control_pcgrl_amd import VecPcgrlEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
g = np.ones((n, 16, 16), np.uint8)
for i in range(n):  # open rooms with 3 crates: state spaces far beyond solver_power, most not solved by the BFS stage
    h, w = int(rng.integers(7, 12)), int(rng.integers(7, 12))
    y0, x0 = int(rng.integers(1, 15 - h)), int(rng.integers(1, 15 - w))
    g[i, y0:y0 + h, x0:x0 + w] = 0
    inner = [(y, x) for y in range(y0 + 1, y0 + h - 1) for x in range(x0 + 1, x0 + w - 1)]
    pick = rng.permutation(len(inner))[:7]
    for c, t in zip(pick, [2, 3, 3, 3, 4, 4, 4]):
        g[i, inner[c][0], inner[c][1]] = t
env = VecPcgrlEnv("sokoban", "wide", (16, 16), max(n, 64), auto_reset=False)
gd = torch.as_tensor(g).to(env.device)
env.stats_for_grids(gd[:1])
torch.cuda.synchronize()
out = np.zeros(16, np.uint64)
env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 16)
t0 = time.perf_counter()
st = env.stats_for_grids(gd)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if PLAIN:
    print(f"{n} levels in one launch: {dt * 1e3:.1f} ms")
    sys.exit(0)
env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 16)
print(f"{n} levels in one launch: {dt * 1e3:.1f} ms; solved: {(st[:, -1] > 0).sum().item() if st.shape[1] else '?'}")
names = ["pop", "record loads + win test", "visited set", "children", "loop overhead"]
for kind, nm in ((0, "BFS"), (1, "A*")):
    o = out[kind * 8:kind * 8 + 8].astype(np.float64)
    it = max(o[5], 1.0)
    print(f"{nm}: {int(o[6])} stages, {int(o[5])} iterations, {o[:5].sum() / it:.0f} cycles / iteration")
    for i, x in enumerate(names):
        print(f"   {x:26s} {o[i] / it:8.0f}")
