"""Round-6 GPU tests (through the C ABI, against the oracle): the N > 1 bench protocol on RCCL (two real GPUs when the box
has them), the resumable Sokoban solver behind the ready mask, device-side target resampling at auto-reset."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pcgrl_oracle as po  # noqa: E402  (checker only)
from conftest import GOLDEN, ROOT  # noqa: E402

REW_TOL = 1e-6


def _vec(*a, **k):
    from control_pcgrl_amd import VecPcgrlEnv
    return VecPcgrlEnv(*a, **k)


def _bench(*argv, env=None, timeout=900):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines and r.stdout.strip().splitlines()[-1] == lines[-1], "the JSON line is the LAST line of stdout:\n" + r.stdout[-2000:]
    return json.loads(lines[-1])


# ------------------------------------------------------------------------------------- N > 1 on RCCL (SURVEY 8 e)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the first real multi-rank RCCL run of the path")
def test_bench_two_gpus_on_rccl():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` (the driver's command line at N = 2), one rank per GPU over RCCL:
    shard seeds, per-rank episode counts, the overlapped exchange, and rank 0's line as the LAST line of stdout."""
    out = _bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rollout-steps", "0", "--rllib-adapter", "0",
                 "--closed-loop-steps", "0", "--sub-batches", "")
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["config"]["envs_per_gpu"] == 4096 and out["config"]["global_envs"] == 8192
    assert out["config"]["seed_ranges"] == [[0x5EED, 0x5EED + 4095], [0x5EED + 4096, 0x5EED + 8191]]
    pr = out["per_rank"]
    assert pr["collective"].startswith("nccl all-gather") and "side stream" in pr["collective"]
    assert len(pr["env_steps_per_s"]) == 2 and pr["episodes"] == [0.0, 0.0]
    tr = out["timed_region"]
    assert tr["exchange"] == "overlap" and tr["protocol_efficiency_bound"] >= 0.95, tr
    # a run long enough for every env of both shards to finish one episode per interval
    out = _bench("--gpus", "2", "--envs", "256", "--steps", "800", "--warmup", "800", "--no-cpu-baseline", "--rollout-steps", "0",
                 "--rllib-adapter", "0", "--closed-loop-steps", "0", "--sub-batches", "")
    assert out["per_rank"]["episodes"] == [256.0, 256.0] and out["episodes"]["episodes"] == 512.0
    assert out["episodes"]["mean_length"] == 770.0
    assert out["timed_region"]["previous_interval_sums"][2] == 512.0


# ------------------------------------------------------------------------------------- resumable solver + ready mask
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _drive_ready(env, orc, n, steps, actions_of, auto_reset, reinject=None, reinject_every=0, check_obs_every=1):
    """Drive the engine through pcgrl_step_ready and the oracle through step_masked: the oracle steps an env exactly when the
    engine says that env emitted a transition, with the action the env consumed.  Returns (emitted transitions, launches in
    which some env was busy, max consecutive busy launches of one env)."""
    busy = env.env_busy().cpu().numpy().astype(bool)
    pend = np.zeros(n, np.int32)
    has_pend = np.zeros(n, bool)
    total = busy_launches = 0
    streak, max_streak = np.zeros(n, int), 0
    for t in range(steps):
        if reinject is not None and reinject_every and t % reinject_every == 0 and t > 0:
            env.reset(init_grids=reinject)  # abandons the steps in flight (the oracle never played them)
            orc.reset(init_grids=reinject)
            has_pend[:] = False
            busy = env.env_busy().cpu().numpy().astype(bool)
        a = actions_of(t)
        consume = ~busy
        pend[consume] = a[consume]
        has_pend[consume] = True
        obs, rew, done, _, info = env.step_ready(torch.as_tensor(a, dtype=torch.int32).cuda())
        status = info["status"].cpu().numpy()
        emitted = (status & 1) != 0
        assert not (emitted & ~has_pend).any(), "a transition without a consumed action"
        oobs, orew, odone, ostats = orc.step_masked(emitted, pend, auto_reset=auto_reset)
        if emitted.any():
            assert np.array_equal(info["stats"].cpu().numpy()[emitted], ostats[emitted]), t
            assert np.abs(rew.cpu().numpy()[emitted] - orew[emitted]).max() <= REW_TOL, t
            assert np.array_equal(done.cpu().numpy()[emitted], odone[emitted]), t
            if t % check_obs_every == 0:
                assert np.array_equal(obs.cpu().numpy()[emitted], oobs[emitted]), t
        has_pend[emitted] = False
        busy = (status & 2) != 0
        assert not (~busy & has_pend).any(), "an idle env still owes a transition"
        total += int(emitted.sum())
        busy_launches += int(busy.any())
        streak = np.where(busy, streak + 1, 0)
        max_streak = max(max_streak, int(streak.max()))
    return total, busy_launches, max_streak


@pytest.mark.parametrize("budget,power", [(24, 300), (200, 300), (64, 10000)])
def test_step_ready_resumable_solver_vs_oracle(budget, power):
    """BASELINE config 4 (sokoban-wide 16x16) with playable levels: the device solver works to a budget per launch, parks
    unfinished searches, busy envs ignore their actions; every emitted transition (stats incl. dist-win / sol-length, reward,
    done, observation) equals the oracle's, which steps an env only when the engine says it stepped."""
    bench = _bench_module()
    n, steps = 96, 160
    maps, cells = bench.solver_active_maps(n, 5)
    acts = bench.solver_active_actions(cells, steps, 9)
    env = _vec("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=False, solver_power=power)
    orc = po.OracleVecEnv("sokoban", "wide", (16, 16), n, seeds=np.arange(n), solver_power=power)
    env.set_solver_budget(budget)
    with pytest.raises(ValueError):
        env.step(torch.zeros(n, dtype=torch.int32).cuda())  # refused: it could not say "busy"
    env.reset(init_grids=torch.as_tensor(maps))
    orc.reset(init_grids=maps)
    total, busy_launches, max_streak = _drive_ready(env, orc, n, steps, lambda t: acts[t], False, reinject=maps, reinject_every=40)
    assert total > 0 and busy_launches > 0, (total, busy_launches)
    if budget <= 64 and power == 300:
        assert max_streak >= 3, "a search parked over several launches"
    # the state of every env that is not waiting for a reset's statistics equals the oracle's
    idle = ~env.env_busy().cpu().numpy().astype(bool)
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"])
    assert np.array_equal(st.stats.cpu().numpy()[idle], ost["stats"][idle])
    env.check_errors()


def test_step_ready_large_budget_equals_synchronous_stepping():
    """with a budget no search exceeds, every env emits in every launch and the outputs are those of pcgrl_step"""
    bench = _bench_module()
    n, steps = 64, 60
    maps, cells = bench.solver_active_maps(n, 6)
    acts = bench.solver_active_actions(cells, steps, 10)
    a = _vec("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=True, solver_power=200)
    b = _vec("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=True, solver_power=200)
    a.set_solver_budget(1 << 20)
    a.reset(init_grids=torch.as_tensor(maps))
    b.reset(init_grids=torch.as_tensor(maps))
    assert int(a.env_busy().sum()) == 0
    for t in range(steps):
        act = torch.as_tensor(acts[t]).cuda()
        oa, ra, da, _, ia = a.step_ready(act)
        ob, rb, db, _, ib = b.step(act)
        assert int((ia["status"] != 1).sum()) == 0
        assert torch.equal(ia["stats"], ib["stats"]) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(oa, ob)
    a.set_solver_budget(0)  # nobody is busy: back to synchronous stepping
    a.step(torch.as_tensor(acts[0]).cuda())
    a.check_errors(); b.check_errors()


def test_step_ready_auto_reset_on_small_maps_vs_oracle():
    """general kernels (6x6 map), auto-reset with short episodes: random 6x6 maps are playable often enough that auto-resets
    leave envs waiting for their new episode's statistics (EMITTED | BUSY, then 0)"""
    n, steps = 256, 400
    kw = dict(change_percentage=0.2, solver_power=400)
    env = _vec("sokoban", "wide", (6, 6), n, seeds=1000 + np.arange(n), auto_reset=True, **kw)
    orc = po.OracleVecEnv("sokoban", "wide", (6, 6), n, seeds=1000 + np.arange(n), **kw)
    env.set_solver_budget(12)
    env.reset()
    orc.reset()
    rng = np.random.default_rng(3)
    # edits biased towards floor / player / crate / target so that levels stay playable now and then
    tiles = rng.choice(5, size=(steps, n), p=[0.5, 0.1, 0.1, 0.15, 0.15])
    acts = (rng.integers(0, 36, size=(steps, n)) * 5 + tiles).astype(np.int32)
    seen = {"eb": 0, "zero": 0}

    class Spy:
        def __init__(self, e):
            self.e = e

        def __getattr__(self, k):
            return getattr(self.e, k)

        def step_ready(self, a):
            out = self.e.step_ready(a)
            s = out[4]["status"]
            seen["eb"] += int((s == 3).sum())
            return out

    total, busy_launches, max_streak = _drive_ready(Spy(env), orc, n, steps, lambda t: acts[t], True)
    assert total > n and busy_launches > 0
    idle = ~env.env_busy().cpu().numpy().astype(bool)
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"])
    assert np.array_equal(st.stats.cpu().numpy()[idle], ost["stats"][idle])
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"])
    assert np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"])
    env.check_errors()
