#!/usr/bin/env python3
"""Shrinking the synchronisation domain (Weak 8): every step launch waits for its slowest env.  With the batch
cut into k independent sub-batches -- k engines of N / k envs, each its own chain of step launches on its own stream,
captured as k parallel branches of ONE HIP graph -- a chain only waits for the slowest env of ITS sub-batch, and the
chains' launches overlap on the device.  Envs are independent objects in the reference (rl/utils.py:412-415: workers x
envs; reps/wrappers.py:80-87), so the results are those of the single batch (tests/test_gpu_round5.py checks two handles
on two streams against the oracle).

  python tools/sub_batch_chains.py [--workloads a,b] [--ks 1,2,4,8,16] [--graph-steps 50] [--replays 20] [--envs N]

Prints one JSON line per (workload, k): us per step of the WHOLE batch (= graph time / steps), env-steps/s, and the
roofline fraction with bench.py's algorithmic bytes."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from control_pcgrl_amd import VecPcgrlEnv

    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="minecraft_3D_maze-narrow,minecraft_3D_maze-narrow-15,binary_bigger-narrow,binary-narrow")
    ap.add_argument("--ks", default="1,2,4,8,16")
    ap.add_argument("--graph-steps", type=int, default=50)
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--warm-replays", type=int, default=4)
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--mode", default="graph", choices=["graph", "threads"],
                    help="graph: k parallel branches of one HIP graph; threads: k host threads, each issuing its own engine's launches "
                         "eagerly (pcgrl_step_seq: one foreign call per chain, the GIL is released) on its own stream")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rows = []
    for wl in args.workloads.split(","):
        problem, rep, shape, n_default = bench.WORKLOADS[wl][:4]
        N = args.envs or n_default
        for k in [int(x) for x in args.ks.split(",")]:
            if N % k:
                continue
            n = N // k
            envs, acts = [], []
            gen = torch.Generator(device=dev).manual_seed(1234)
            POOL = 256
            for i in range(k):
                e = VecPcgrlEnv(problem, rep, shape, n, device=dev, seeds=0x5EED + i * n + np.arange(n), auto_reset=True)
                e.reset()
                envs.append(e)
                acts.append(torch.randint(0, e.num_actions, (POOL, n), generator=gen, device=dev, dtype=torch.int32))
            G = args.graph_steps
            main_s = torch.cuda.current_stream(dev)
            cap = torch.cuda.Stream(dev)
            branches = [torch.cuda.Stream(dev) for _ in range(k)]
            # warm every engine eagerly first (lazy allocations, first-launch costs)
            for i, e in enumerate(envs):
                for t in range(3):
                    e.step_raw(acts[i][t].data_ptr(), main_s.cuda_stream)
            torch.cuda.synchronize(dev)
            if args.mode == "threads":
                import threading, time
                steps = G * args.replays
                bar = threading.Barrier(k + 1)
                strs = [torch.cuda.Stream(dev) for _ in range(k)]

                def chain(i):
                    torch.cuda.set_device(dev)
                    e, a, st = envs[i], acts[i], strs[i]
                    e.step_seq_raw(a.data_ptr(), n, POOL, 0, G, st.cuda_stream)  # warm
                    st.synchronize()
                    bar.wait()
                    rc = e.step_seq_raw(a.data_ptr(), n, POOL, 0, steps, st.cuda_stream)
                    assert rc == 0, rc
                    st.synchronize()
                    bar.wait()

                ths = [threading.Thread(target=chain, args=(i,)) for i in range(k)]
                for t in ths:
                    t.start()
                bar.wait()
                t0 = time.perf_counter()
                bar.wait()
                us = (time.perf_counter() - t0) * 1e6 / steps
                for t in ths:
                    t.join()
                for e in envs:
                    e.check_errors()
                row = {"workload": wl, "envs": N, "sub_batches": k, "envs_per_sub_batch": n, "mode": "threads", "steps": steps,
                       "us_per_step_of_whole_batch": us, "env_steps_per_s": N / (us * 1e-6),
                       "roofline_frac": bench.ALGO_BYTES[wl] * N / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBS}
                rows.append(row)
                print(json.dumps(row), flush=True)
                for e in envs:
                    e.close()
                torch.cuda.empty_cache()
                continue
            g = torch.cuda.CUDAGraph()
            cap.wait_stream(main_s)
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    cs = torch.cuda.current_stream(dev)
                    for i, e in enumerate(envs):
                        b = branches[i] if k > 1 else cs
                        if k > 1:
                            b.wait_stream(cs)  # fork
                        for t in range(G):
                            rc = e.step_raw(acts[i][t % POOL].data_ptr(), b.cuda_stream)
                            assert rc == 0, rc
                    if k > 1:
                        for b in branches:
                            cs.wait_stream(b)  # join
            main_s.wait_stream(cap)
            for _ in range(args.warm_replays):
                g.replay()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            for _ in range(args.replays):
                g.replay()
            e1.record(main_s)
            torch.cuda.synchronize(dev)
            for e in envs:
                e.check_errors()
            us = e0.elapsed_time(e1) * 1e3 / (args.replays * G)
            row = {"workload": wl, "envs": N, "sub_batches": k, "envs_per_sub_batch": n, "graph_steps": G,
                   "us_per_step_of_whole_batch": us, "env_steps_per_s": N / (us * 1e-6),
                   "roofline_frac": bench.ALGO_BYTES[wl] * N / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBS}
            rows.append(row)
            print(json.dumps(row), flush=True)
            del g
            for e in envs:
                e.close()
            torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
