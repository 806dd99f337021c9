import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import pcgrl_oracle as po
from control_pcgrl_amd import VecPcgrlEnv
def soak(shape, n, T, seed0, **kw):
    env = VecPcgrlEnv("minecraft_3D_maze", "narrow", shape, n, seeds=seed0 + np.arange(n), auto_reset=True, **kw)
    orc = po.OracleVecEnv("minecraft_3D_maze", "narrow", shape, n, seeds=seed0 + np.arange(n), threads=8, **kw)
    obs, _ = env.reset(); oobs = orc.reset()
    assert np.array_equal(obs.cpu().numpy(), oobs)
    g = torch.Generator().manual_seed(seed0)
    for t in range(T):
        # biased actions: long runs of AIR / DIRT make big open components and long corridors
        p = 0.5 + 0.45 * np.sin(t / 97.0)
        a = (torch.rand(n, generator=g) < p).to(torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        want = t % 211 == 0 or t == T - 1
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=want)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), (shape, t)
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= 1e-6, (shape, t)
        assert np.array_equal(done.cpu().numpy(), odone)
        if want:
            assert np.array_equal(obs.cpu().numpy(), oobs), (shape, t)
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    env.check_errors()
    print("ok", shape, n, T, "max path", int(ostats[:, 1].max()), flush=True)
soak((7, 7, 7), 2048, 2300, 5000)
soak((15, 15, 15), 96, 900, 6000, change_percentage=0.05)
soak((8, 8, 8), 512, 1700, 7000)
soak((4, 4, 4), 700, 500, 8000)
