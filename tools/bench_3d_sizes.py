import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
for shape, n in (((15, 15, 15), 1024), ((10, 10, 10), 1024), ((7, 7, 7), 1024)):
    env = VecPcgrlEnv("minecraft_3D_maze", "narrow", shape, n, seeds=np.arange(n), auto_reset=True)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    pool = torch.randint(0, 2, (256, n), generator=g, device="cuda", dtype=torch.int32)
    sp = torch.cuda.current_stream().cuda_stream
    for k in range(300):
        env.step_raw(pool[k % 256].data_ptr(), sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 1000
    for k in range(K):
        env.step_raw(pool[k % 256].data_ptr(), sp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.check_errors()
    print(shape, n, "envs: %.1f us per step launch, %.3g env-steps/s, obs %.1f MB per launch" % (dt / K * 1e6, n * K / dt, n * np.prod(env.obs_shape) / 1e6), flush=True)
    env.close()
