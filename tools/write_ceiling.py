#!/usr/bin/env python3
"""What a write-only kernel reaches on this GPU: the practical ceiling next to the 8 TB/s spec peak the roofline uses.
(a) device fill of buffers the size of one launch's algorithmic bytes (4096 / 65536 binary envs) and of 2 GiB;
(b) the engine's observe kernel alone (the step's dominant traffic: 3072 B of one-hot observation per env).
Times are HIP events on the launch stream, mean over many back-to-back launches.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from control_pcgrl_amd import VecPcgrlEnv

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)


def timed(fn, iters):
    for _ in range(max(5, iters // 10)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3  # s


out = {"fill": [], "observe_kernel": []}
for nbytes, iters in ((4096 * 3348, 4000), (65536 * 3348, 1000), (2 << 30, 60)):
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            buf.fill_(1)
    t = timed(g.replay, max(iters // 20, 3)) / 20
    out["fill"].append({"bytes": nbytes, "us": t * 1e6, "GBps": nbytes / t / 1e9})
    del buf, g
for n, iters in ((4096, 4000), (65536, 600)):
    env = VecPcgrlEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True)
    env.reset()
    obs = torch.empty((n,) + env.obs_shape, dtype=torch.uint8, device=dev)
    sp = stream.cuda_stream
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            cap = torch.cuda.current_stream(dev).cuda_stream
            for _ in range(20):
                env._L.pcgrl_observe(env._h, obs.data_ptr(), cap)
    stream.wait_stream(side)
    t = timed(g.replay, max(iters // 20, 3)) / 20
    out["observe_kernel"].append({"envs": n, "bytes": obs.numel(), "us": t * 1e6, "GBps": obs.numel() / t / 1e9})
    del env, obs, g
print(json.dumps(out))
