#!/usr/bin/env python3
"""Open-loop rollout kernel (pcgrl_rollout) vs per-step launches: env-steps/s for several steps-per-launch values.
python tools/rollout_bench.py [workload] [envs] [form: -1 by shape (default) | 0 step launches | 1 one two-role kernel | 2 two kernels] [steps per launch, comma list]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv

workload = sys.argv[1] if len(sys.argv) > 1 else "binary-narrow"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
problem, rep = workload.split("-")
env = VecPcgrlEnv(problem, rep, (16, 16), n, seeds=np.arange(n), auto_reset=True)
form = int(sys.argv[3]) if len(sys.argv) > 3 else -1
assert env._L.pcgrl_set_rollout_form(env._h, form) == 0
GS = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 4, 16, 64, 256)
env.reset()
dev = env.device
TOTAL = 4096
pool = torch.randint(0, env.num_actions, (TOTAL, n), generator=torch.Generator(device=dev).manual_seed(1), device=dev, dtype=torch.int32)
sp = torch.cuda.current_stream().cuda_stream
for mode in ("all", "last", "none"):
    for G in GS:
        obs = torch.empty(((G if mode == "all" else 1), n) + env.obs_shape, dtype=torch.uint8, device=dev)
        rew = torch.empty((G, n), dtype=torch.float32, device=dev); done = torch.empty((G, n), dtype=torch.uint8, device=dev)
        stats = torch.empty((G, n, env.n_stats), dtype=torch.int32, device=dev)
        def run(iters):
            for i in range(iters):
                rc = env._L.pcgrl_rollout(env._h, pool[(i * G) % (TOTAL - G + 1)].data_ptr(), G, 1, obs.data_ptr() if mode != "none" else 0, 0 if mode == "all" else 1,
                                          rew.data_ptr(), done.data_ptr(), stats.data_ptr(), sp)
                assert rc == 0
        iters = max(8, 8192 // G)
        run(max(2, iters // 8)); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(iters); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (iters * G)
        print(f"{workload} {n} envs, form {form}, obs={mode:4s} steps/launch {G:4d}: {us:7.3f} us/step  {n / us * 1e6:.3e} env-steps/s", flush=True)
env.check_errors()
