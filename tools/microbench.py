#!/usr/bin/env python3
"""Kernel-level timing of the step path (HIP events on the launch stream): full step, step without the
observation, observe only; eager launches vs a captured HIP graph.  Development aid for DESIGN.md's tables."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from control_pcgrl_amd import VecPcgrlEnv


def timeit(fn, iters, stream):
    for _ in range(max(10, iters // 10)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="binary-narrow")
    ap.add_argument("--envs", default="4096,16384,65536")
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--static", action="store_true", help="static tiles (p <= 0.3, 3 walls)")
    ap.add_argument("--patch", default="", help="action patch, e.g. 3x3")
    args = ap.parse_args()
    problem, rep = args.workload.split("-")
    dev = torch.device("cuda:0")
    for n in [int(x) for x in args.envs.split(",")]:
        kw = {}
        if args.static:
            kw.update(static_prob=0.3, n_static_walls=3)
        if args.patch:
            kw.update(act_window=[int(x) for x in args.patch.split("x")])
        env = VecPcgrlEnv(problem, rep, (16, 16), n, device=dev, seeds=np.arange(n), auto_reset=True, **kw)
        env.reset()
        g = torch.Generator(device=dev).manual_seed(1)
        pool = torch.randint(0, env.num_actions, (1021, n * env.action_entries), generator=g, device=dev, dtype=torch.int32)
        stream = torch.cuda.current_stream(dev)
        sp = stream.cuda_stream
        L, h = env._L, env._h
        k = [0]

        def full():
            k[0] += 1
            env.step_raw(pool[k[0] % 1021].data_ptr(), sp)

        def noobs():
            k[0] += 1
            L.pcgrl_step(h, pool[k[0] % 1021].data_ptr(), 1, None, env._ptrs[1], env._ptrs[2], env._ptrs[3], sp)

        def obs_only():
            L.pcgrl_observe(h, env._ptrs[0], sp)

        res = {"envs": n, "full_us": timeit(full, args.iters, stream), "noobs_us": timeit(noobs, args.iters, stream),
               "observe_us": timeit(obs_only, args.iters, stream)}
        if args.graph:
            gr = torch.cuda.CUDAGraph()
            K = 256
            s2 = torch.cuda.Stream(dev)
            with torch.cuda.stream(s2):
                ptrs = [pool[i].data_ptr() for i in range(K)]
                with torch.cuda.graph(gr, stream=s2):
                    sp2 = torch.cuda.current_stream(dev).cuda_stream
                    for i in range(K):
                        env.step_raw(ptrs[i], sp2)
            torch.cuda.synchronize()
            res["graph_us_per_step"] = timeit(gr.replay, max(10, args.iters // K), torch.cuda.current_stream(dev)) / K
        res["steps_per_s_full"] = n / res["full_us"] * 1e6
        print(res, flush=True)
        env.close()


if __name__ == "__main__":
    main()
