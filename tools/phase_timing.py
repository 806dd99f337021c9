#!/usr/bin/env python3
"""Phase breakdown of the simulate wave (shader-clock cycles, summed over workgroups' lane 0) using a library built
with -DPCGRL_PHASE_TIMING.  Build here (hipcc cross-compiles):  python tools/phase_timing.py --build
then on the GPU box:                                            python tools/phase_timing.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib

TIMING_LIB = os.path.join(_lib.CSRC, "libpcgrl_amd_timing.so")
NAMES = ["loads+barrier", "action+state", "stats refresh (total)", "  flood", "  first sweeps", "  second sweep",
         "loss/outputs/write-back"]

if "--build" in sys.argv:
    _lib.build(force=True, out=TIMING_LIB, defines=("PCGRL_PHASE_TIMING",) + (("PCGRL_M3_PHASES",) if "--m3-phases" in sys.argv else ()) + (("PCGRL_M3_PHASES", "PCGRL_M3_TRIPS") if "--m3-trips" in sys.argv else ()) + (("PCGRL_M3_SPEC",) if "--m3-spec" in sys.argv else ()) + (("PCGRL_M3_TAIL",) if "--m3-tail" in sys.argv else ()) + (("PCGRL_M3_TAILSPLIT",) if "--m3-tailsplit" in sys.argv else ()) + (("PCGRL_M3_HEADSPLIT",) if "--m3-headsplit" in sys.argv else ()))
    print("built", TIMING_LIB)
    sys.exit(0)

import numpy as np
import torch

_lib.LIB_PATH = TIMING_LIB
from control_pcgrl_amd import VecPcgrlEnv

three_d = "--3d" in sys.argv
soko = "--sokoban" in sys.argv  # sokoban-wide 16x16, 2048 envs (BASELINE C4)
n, iters = (1024, 1000) if three_d else ((2048, 2000) if soko else (4096, 2000))
if three_d:
    if "--m3-spec" in sys.argv:
        NAMES = ["(count) search pairs", "(count) with a remembered farthest cell", "(count) helper result used", "(count) farthest cell = the last far END", "(count) start cell changed", "-", "-"]
    elif "--m3-trips" in sys.argv:
        NAMES = ["(count) chain trips", "(count) general trips", "chain-trip cycles", "general-trip cycles", "overlay", "outputs + write-back", "fresh tables"]
    elif "--m3-phases" in sys.argv:
        NAMES = ["loads until the barrier", "columns + move-table update", "regions", "candidate walk", "overlay", "outputs + write-back", "fresh tables (reset)"]
    else:
      NAMES = ["(count) trips", "(count) queue entries", "(count) searches", "  search loops", "regions", "everything else", "candidate walk incl. search loops"]
    size = int(next((a[7:] for a in sys.argv if a.startswith("--size=")), "7"))  # --size=15: the reference's stock map
    knob = int(next((a[7:] for a in sys.argv if a.startswith("--knob=")), "10000"))  # development: see PCGRL_PHASE_TIMING in pcgrl_kernels3d.h
    env = VecPcgrlEnv("minecraft_3D_maze", "narrow", (size, size, size), n, seeds=np.arange(n), auto_reset=True, solver_power=knob)
elif soko:
    env = VecPcgrlEnv("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=True)
else:
    # --static / --patch: the representation wrappers (general kernels); --general: the general kernel on the plain config
    # (statistics left stale by one pcgrl_update keep Params::no_fast set)
    kw = dict(static_prob=0.3, n_static_walls=3) if "--static" in sys.argv else dict(act_window=[3, 3]) if "--patch" in sys.argv else {}
    env = VecPcgrlEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, **kw)
if soko:
    NAMES = ["loads+barrier", "action+state", "stats refresh", "loss + outputs", "auto-reset block", "-", "state write-back"]
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, 256 * 5 if soko else 2, (1021, n * env.action_entries), generator=g, device="cuda", dtype=torch.int32)
if "--general" in sys.argv:
    env.update(pool[0])
sp = torch.cuda.current_stream().cuda_stream
WARM = int(next((a[7:] for a in sys.argv if a.startswith("--warm=")), "300"))
for k in range(WARM):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
blocks = n if three_d else n // 4
out = np.zeros(8 * blocks, np.uint64)
env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * blocks)
NO_OBS = "--no-obs" in sys.argv  # launches without an observation buffer: no observe waves, no observation stores
for k in range(iters):
    if NO_OBS:
        env._L.pcgrl_step(env._h, pool[k % 1021].data_ptr(), 1, None, env._ptrs[1], env._ptrs[2], env._ptrs[3], sp)
    else:
        env.step_raw(pool[k % 1021].data_ptr(), sp)
env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * blocks)
out = out.reshape(blocks, 8).astype(np.float64).sum(0)
tot = 0
for i, nm in enumerate(NAMES):
    cyc = float(out[i]) / (iters * blocks)
    if not nm.startswith("  "):
        tot += cyc
    print(f"{nm:28s} {cyc:11.3f} cycles/launch/wave")
print(f"{'sum of top-level phases':28s} {tot:9.0f}")
wall = float(out[7]) / (iters * blocks)  # 100 MHz constant clock ticks
print(f"simulate wave lifetime: {wall * 10:.0f} ns  => effective shader clock {sum(out[:7]) / (iters * blocks) / (wall * 10):.2f} GHz")
