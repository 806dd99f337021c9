#!/usr/bin/env python3
"""Where does a step launch spend its time?  Per-workgroup wall-clock stamps (100 MHz) of the simulate and observe
waves of single launches, from a library built with -DPCGRL_WAVE_TRACE.
Build here:  python tools/wave_trace.py --build      Run on the GPU box:  python tools/wave_trace.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_pcgrl_amd import _lib

TRACE_LIB = os.path.join(_lib.CSRC, "libpcgrl_amd_trace.so")
if "--build" in sys.argv:
    _lib.build(force=True, out=TRACE_LIB, defines=("PCGRL_WAVE_TRACE",))
    print("built", TRACE_LIB)
    sys.exit(0)

import numpy as np
import torch

_lib.LIB_PATH = TRACE_LIB
from control_pcgrl_amd import VecPcgrlEnv

three_d = "--3d" in sys.argv
soko, zelda = "--sokoban" in sys.argv, "--zelda" in sys.argv
n = 1024 if three_d else (2048 if soko else 4096)
custom = next((a for a in sys.argv if a.startswith("--cfg=")), None)  # --cfg=problem,rep,envs,dim0,dim1[,dim2]
epb = 4  # envs per workgroup (16 lanes per env)
if custom:
    f = custom[6:].split(",")
    shape = tuple(int(x) for x in f[3:])
    n = int(f[2])
    three_d = len(shape) == 3
    kw = dict(static_prob=0.3, n_static_walls=3) if "--static" in sys.argv else dict(act_window=[3, 3]) if "--patch" in sys.argv else {}
    env = VecPcgrlEnv(f[0], f[1], shape, n, seeds=np.arange(n), auto_reset=True, **kw)
    lpe = 8 if shape[0] <= 8 else 16 if shape[0] <= 16 else 32 if shape[0] <= 32 else 64
    epb = 64 // lpe
elif soko:
    env = VecPcgrlEnv("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=True)
elif zelda:
    env = VecPcgrlEnv("zelda", "turtle", (16, 16), n, seeds=np.arange(n), auto_reset=True)
elif three_d:
    env = VecPcgrlEnv("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=np.arange(n), auto_reset=True)
else:
    env = VecPcgrlEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset="--no-reset" not in sys.argv)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, env.num_actions, (1021, n * env.action_entries), generator=g, device="cuda", dtype=torch.int32)
if "--general" in sys.argv:  # statistics left stale by one pcgrl_update keep the general kernels in use (Params::no_fast)
    env.update(pool[0])
sp = torch.cuda.current_stream().cuda_stream
WARM = int(next((a[7:] for a in sys.argv if a.startswith("--warm=")), "500"))
for k in range(WARM):
    env.step_raw(pool[k % 1021].data_ptr(), sp)
torch.cuda.synchronize()
blocks = n if three_d else n // epb
rows = []
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(200):
    # a few back-to-back launches so that the traced (last) one starts right behind its predecessor, as in bench.py
    ev0.record()
    for j in range(8):
        env.step_raw(pool[(500 + it * 8 + j) % 1021].data_ptr(), sp)
    ev1.record()
    torch.cuda.synchronize()
    out = np.zeros(8 * blocks, np.uint64)
    env._L.pcgrl_debug_counters(env._h, out.ctypes.data, 8 * blocks)
    t = out.reshape(blocks, 8).astype(np.int64)
    if three_d:  # simulate wave: slots 0 start, 1 end, 2 end after the stores were acknowledged; observe wave: 4 start, 3 stores issued
        t0 = t[:, 0].min()
        rows.append(dict(start=t[:, 0] - t0, end=t[:, 1] - t0, acked=t[:, 2] - t0, obs_start=t[:, 4] - t0, obs_issued=t[:, 3] - t0,
                         per_launch_us=ev0.elapsed_time(ev1) * 1e3 / 8))
        continue
    t0 = min(t[:, 0].min(), t[:, 2].min())
    rows.append(dict(sim_start=(t[:, 0] - t0), sim_end=(t[:, 1] - t0), obs_start=(t[:, 2] - t0),
                     obs_issued=(t[:, 3] - t0), obs_acked=(t[:, 4] - t0), per_launch_us=ev0.elapsed_time(ev1) * 1e3 / 8))


def q(a, f):
    return float(np.quantile(a, f)) * 10.0  # ticks of 10 ns -> ns


print("all numbers in ns relative to the first wave start of the launch; median over 200 traced launches")
if three_d:
    for key in ("start", "end", "acked", "obs_start", "obs_issued"):
        print(f"{key:8s} median wave {np.median([q(r[key], 0.5) for r in rows]):8.0f}   p95 {np.median([q(r[key], 0.95) for r in rows]):8.0f}"
              f"   last wave {np.median([q(r[key], 1.0) for r in rows]):8.0f}")
    print(f"wave lifetime mean {np.median([np.mean(r['end'] - r['start']) * 10.0 for r in rows]):.0f} max "
          f"{np.median([np.max(r['end'] - r['start']) * 10.0 for r in rows]):.0f} ns; per-launch time by HIP events "
          f"{np.median([r['per_launch_us'] for r in rows]):.2f} us")
    sys.exit(0)
for key in ("sim_start", "obs_start", "sim_end", "obs_issued", "obs_acked"):
    med = np.median([q(r[key], 0.5) for r in rows])
    p95 = np.median([q(r[key], 0.95) for r in rows])
    mx = np.median([q(r[key], 1.0) for r in rows])
    print(f"{key:11s} median wave {med:7.0f}   p95 {p95:7.0f}   last wave {mx:7.0f}")
last = np.median([max(r["sim_end"].max(), r["obs_acked"].max()) * 10.0 for r in rows])
print(f"last stamp of the launch {last:.0f} ns; per-launch time by HIP events (8 back-to-back) {np.median([r['per_launch_us'] for r in rows]):.2f} us")
simlife = np.median([np.mean(r["sim_end"] - r["sim_start"]) * 10.0 for r in rows])
simmax = np.median([np.max(r["sim_end"] - r["sim_start"]) * 10.0 for r in rows])
obslife = np.median([np.mean(r["obs_acked"] - r["obs_start"]) * 10.0 for r in rows])
print(f"simulate wave lifetime mean {simlife:.0f} max {simmax:.0f} ns; observe wave lifetime (to store ack) mean {obslife:.0f} ns")
