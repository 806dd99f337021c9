import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
n = 4096
def run(tag, stale, **kw):
    env = VecPcgrlEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, **kw)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    na = env.action_entries
    pool = torch.randint(0, 2, (256, n * na), generator=g, device="cuda", dtype=torch.int32)
    if stale:  # pcgrl_update marks the engine "maybe stale": the general kernels run from here on
        env.update(pool[0].view(n, na) if na > 1 else pool[0], want_obs=False)
    sp = torch.cuda.current_stream().cuda_stream
    for k in range(500):
        env.step_raw(pool[k % 256].data_ptr(), sp)
    torch.cuda.synchronize()
    K = 4000
    t0 = time.perf_counter()
    for k in range(K):
        env.step_raw(pool[k % 256].data_ptr(), sp)
    torch.cuda.synchronize()
    print(tag, "%.2f us per eager step launch" % ((time.perf_counter() - t0) / K * 1e6), flush=True)
    env.close()
run("FAST 16x16 kernel      ", False)
run("general kernel, no ext ", True)
run("general, static tiles  ", False, static_prob=0.3, n_static_walls=3)
run("general, 3x3 patch     ", False, act_window=[3, 3])
