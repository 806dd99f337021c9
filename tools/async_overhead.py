#!/usr/bin/env python3
"""What asynchronous stepping costs when the solver is NOT firing (development measurement): BASELINE config 4 -- sokoban-wide
16x16, 2048 envs, uniform random actions, auto-reset -- stepped through pcgrl_step (synchronous: 64 / LPE envs per wavefront) and
through pcgrl_step_ready with a solver budget (one env per workgroup, the resumable-solver kernel).  HIP graphs of 100 launches,
HIP events.  On the GPU box:  python tools/async_overhead.py [budget]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from control_pcgrl_amd import VecPcgrlEnv

budget = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, G, R = 2048, 100, 40
dev = torch.device("cuda:0")
acts = torch.randint(0, 1280, (G, n), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
for mode in ("pcgrl_step", f"pcgrl_step_ready (budget {budget})"):
    env = VecPcgrlEnv("sokoban", "wide", (16, 16), n, device=dev, seeds=np.arange(n), auto_reset=True)
    ready = mode != "pcgrl_step"
    if ready:
        env.set_solver_budget(budget)
    env.reset()
    step = env.step_ready if ready else env.step
    for t in range(20):
        step(acts[t])
    torch.cuda.synchronize()
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for t in range(G):
                step(acts[t])
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    env.check_errors()
    us = e0.elapsed_time(e1) * 1e3 / (R * G)
    extra = ""
    if ready:
        st = env._status
        extra = f"  (last launch: {int((st & 1).sum())} of {n} envs emitted, {int((st & 2).ne(0).sum())} busy)"
    print(f"{mode:36s} {us:7.2f} us per launch  {n / us * 1e6:.3e} env-launches/s{extra}", flush=True)
    env.close()
