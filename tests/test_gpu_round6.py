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
    e.setdefault("PCGRL_BENCH_RANK_TIMEOUT", "300")  # (a rank that hangs ends itself well inside this test's own timeout)
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


# ------------------------------------------------------------------------------------- device-side target resampling
M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
    return z ^ (z >> 31)


def _resampled(seed, env, c, j, lo, hi):
    """host restatement of trg_resampled (csrc/pcgrl_kernels2d.h)"""
    r = _mix64(_mix64((seed + c * 0x9e3779b97f4a7c15) & M64) ^ ((env * 0xd1b54a32d192ed03 + (j + 1) * 0x8cb92ba72f3d8dd7) & M64))
    return float(r >> 11) * (1.0 / 9007199254740992.0) * (hi - lo) + lo


@pytest.mark.parametrize("problem,rep,shape,controls", [("binary", "narrow", (16, 16), ["regions", "path-length"]),
                                                        ("zelda", "turtle", (16, 16), ["nearest-enemy", "path-length"]),
                                                        ("minecraft_3D_maze", "narrow", (7, 7, 7), ["n_jump", "path-length"])])
def test_target_resampling_in_a_captured_graph_vs_oracle(problem, rep, shape, controls):
    """pcgrl_set_target_resampling: a HIP graph of T step launches spanning several episodes re-targets every env at every
    auto-reset with no host call; rewards (float64), control observations and stats equal the oracle's, which is handed the
    same draws through its target queue (the reference's UniformNoiseyTargets.reset, control_wrappers.py:453-471)."""
    n, T, seed = 192, 150, 12345
    kw = dict(controls=controls, change_percentage=0.05)
    env = _vec(problem, rep, shape, n, seeds=np.arange(n), auto_reset=True, reward_dtype=torch.float64, **kw)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=np.arange(n), **kw)
    env.set_target_resampling(True, seed)
    bounds = [env.spec.cond_bounds[k] for k in controls]
    cnt = np.zeros(n, int)

    def draws():
        return {k: np.array([_resampled(seed, i, int(cnt[i]), j, bounds[j][0], bounds[j][1]) for i in range(n)]) for j, k in enumerate(controls)}

    orc.queue_targets(draws())
    cnt += 1
    env.reset()
    orc.reset()
    assert np.allclose(env.ctrl_obs.cpu().numpy(), orc.ctrl_obs(), rtol=0, atol=1e-6)
    g = torch.Generator().manual_seed(3)
    acts = torch.randint(0, env.num_actions, (T, n), generator=g, dtype=torch.int32).cuda()
    rew = torch.zeros((T, n), dtype=torch.float64, device="cuda")
    done = torch.zeros((T, n), dtype=torch.uint8, device="cuda")
    stats = torch.zeros((T, n, env.n_stats), dtype=torch.int32, device="cuda")
    cobs = torch.zeros((T, n, 2 * len(controls)), dtype=torch.float32, device="cuda")
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            cap = torch.cuda.current_stream().cuda_stream
            for t in range(T):
                rc = env._L.pcgrl_step_ex(env._h, acts[t].data_ptr(), 1, None, None, rew[t].data_ptr(), done[t].data_ptr(),
                                          stats[t].data_ptr(), cobs[t].data_ptr(), cap)
                assert rc == 0
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    env.check_errors()
    rew, done, stats, cobs, a_np = rew.cpu().numpy(), done.cpu().numpy().astype(bool), stats.cpu().numpy(), cobs.cpu().numpy(), acts.cpu().numpy()
    for t in range(T):
        orc.queue_targets(draws())  # what the next reset of each env will draw
        _, orew, odone, ostats = orc.step(a_np[t], auto_reset=True, want_obs=False)
        assert np.array_equal(done[t], odone), t
        assert np.array_equal(stats[t], ostats), t
        assert np.abs(rew[t] - orew).max() <= 1e-9, (t, np.abs(rew[t] - orew).max())
        assert np.allclose(cobs[t], orc.ctrl_obs(), rtol=0, atol=1e-6), t
        cnt += odone
    assert done.sum(0).min() >= 2, "every env re-targeted at least twice inside the graph"
    # the same through pcgrl_rollout_ex (several resets of an env inside ONE launch) on a twin engine
    twin = _vec(problem, rep, shape, n, seeds=np.arange(n), auto_reset=True, reward_dtype=torch.float64, **kw)
    twin.set_target_resampling(True, seed)
    twin.reset()
    _, r2, d2, s2 = twin.rollout(acts, want_obs="none")
    assert np.array_equal(d2.cpu().numpy(), done) and np.array_equal(s2.cpu().numpy(), stats)
    assert np.abs(r2.cpu().numpy() - rew).max() <= 1e-9
    assert np.allclose(twin.ctrl_obs.cpu().numpy(), cobs[-1], rtol=0, atol=1e-6)
    twin.check_errors()


# ------------------------------------------------------------------------------------- round-5 advisor findings
def test_set_static_after_a_checkpoint_keeps_the_imported_eval_mode():
    """a checkpoint taken in evaluation mode, loaded into an engine created in training mode; set_static(static_prob=x)
    afterwards must not switch the engine back (the wrapper used to pass its construction-time flag)"""
    n = 48
    kw = dict(static_prob=0.4, n_static_walls=2)
    src = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), static_eval=True, **kw)
    dst = _vec("binary", "narrow", (16, 16), n, seeds=100 + np.arange(n), static_eval=False, **kw)
    src.reset()
    dst.reset()
    dst.load_state_dict(src.state_dict())
    for e in (src, dst):
        e.set_static(static_prob=0.25)  # eval_mode=None: keep what the engine holds
        e.reset()
    assert torch.equal(src.get_static(), dst.get_static())
    orc = po.OracleVecEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), static_eval=True, **kw)
    orc.reset()
    orc.set_static(static_prob=0.25, eval_mode=True)
    orc.reset()
    assert np.array_equal(dst.get_static().cpu().numpy(), orc.static_tiles())
    src.check_errors(); dst.check_errors()


def test_3d_non_temporal_observation_stores_vs_oracle(monkeypatch):
    """the 3-D observe waves' non-temporal store path (launches beyond the threshold, 384 MB by default): the threshold is read
    at pcgrl_create, so an engine created under PCGRL_OBS_NT_MB=1 takes that path at a testable batch size"""
    monkeypatch.setenv("PCGRL_OBS_NT_MB", "1")
    n = 160  # 160 x 10 976 B = 1.76 MB per launch
    env = _vec("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=np.arange(n), auto_reset=True)
    monkeypatch.delenv("PCGRL_OBS_NT_MB")
    orc = po.OracleVecEnv("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=np.arange(n), threads=8)
    obs, _ = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(1)
    for t in range(60):
        a = torch.randint(0, 2, (n,), generator=g, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a.cuda())
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True)
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), t
    K = 40
    a = torch.randint(0, 2, (K, n), generator=g, dtype=torch.int32)
    robs, _, _, _ = env.rollout(a.cuda(), want_obs="all")
    for t in range(K):
        oobs, _, _, _ = orc.step(a[t].numpy(), auto_reset=True)
        assert np.array_equal(robs[t].cpu().numpy(), oobs), t
    env.check_errors()


def test_sub_batched_env_forwards_the_rest_of_the_surface():
    from control_pcgrl_amd import SubBatchedVecEnv, make_vec_env
    n = 64
    kw = dict(static_prob=0.3, n_static_walls=2)
    sb = SubBatchedVecEnv("binary", "narrow", (16, 16), n, 2, seeds=np.arange(n), **kw)
    one = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), **kw)
    for e in (sb, one):
        e.set_static(static_prob=0.2, eval_mode=True)
        e.reset()
    assert torch.equal(sb.get_static(), one.get_static())
    assert torch.equal(sb.get_rng_state(), one.get_rng_state())
    sb.seed(500 + np.arange(n)); one.seed(500 + np.arange(n))
    sb.reset(); one.reset()
    assert torch.equal(sb.get_static(), one.get_static())
    # a temporary handed to step_async (freed as soon as the call returns) is kept alive for the sub-stream's read
    for t in range(20):
        for i in range(2):
            sb.step_async(i, torch.full((n // 2,), t % 2, dtype=torch.int64, device="cuda").int())
        sb.wait()
        one.step(torch.full((n,), t % 2, dtype=torch.int32, device="cuda"))
    assert torch.equal(sb.get_state().grids, one.get_state().grids)
    mask = torch.zeros(n, dtype=torch.uint8); mask[::3] = 1
    sb.load_state_dict(sb.state_dict(), mask=mask)
    assert sb.solver_pool_slots() == (0, 0, False)
    with pytest.raises(NotImplementedError):
        make_vec_env({"task": {"problem": "binary", "map_shape": [16, 16]}, "representation": "narrow", "controls": ["regions"]}, 8, sub_batches=2)
    sb.check_errors(); one.check_errors()


def test_step_ready_captured_in_a_hip_graph_equals_eager_launches():
    """pcgrl_step_ready takes no host-side decision per launch: a captured chain of T launches (each with its own status row)
    replays to exactly what T eager launches of a twin engine produce -- statuses, stats, rewards"""
    bench = _bench_module()
    n, T = 128, 48
    maps, cells = bench.solver_active_maps(n, 11)
    acts = torch.as_tensor(bench.solver_active_actions(cells, T, 12)).cuda()
    out = []
    for captured in (True, False):
        env = _vec("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=False, solver_power=300)
        env.set_solver_budget(20)
        env.reset(init_grids=torch.as_tensor(maps))
        status = torch.zeros((T, n), dtype=torch.uint8, device="cuda")
        stats = torch.zeros((T, n, env.n_stats), dtype=torch.int32, device="cuda")
        rew = torch.zeros((T, n), dtype=torch.float32, device="cuda")

        def launches(stream):
            for t in range(T):
                rc = env._L.pcgrl_step_ready(env._h, acts[t].data_ptr(), 0, None, rew[t].data_ptr(), None, stats[t].data_ptr(),
                                             status[t].data_ptr(), stream)
                assert rc == 0
        if captured:
            graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    launches(torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().wait_stream(side)
            graph.replay()
        else:
            launches(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        env.check_errors()
        emitted = (status & 1).bool()
        out.append((status.cpu(), torch.where(emitted[..., None], stats, torch.zeros_like(stats)).cpu(),
                    torch.where(emitted, rew, torch.zeros_like(rew)).cpu(), env.get_state().grids.cpu()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert int((out[0][0] & 2).sum()) > 0 and int((out[0][0] & 1).sum()) > 0


# ------------------------------------------------------------------------------------- rollout roles as kernels of their own
@pytest.mark.parametrize("form", [2, -1, "crowded"])
@pytest.mark.parametrize("problem,rep,n", [("binary", "narrow", 300), ("zelda", "turtle", 150), ("sokoban", "wide", 96)])
def test_rollout_two_kernel_form_and_simulate_only_vs_oracle(problem, rep, n, form, monkeypatch):
    """pcgrl_set_rollout_form(2): the simulate role on the caller's stream, the observe role on the engine's side stream from a
    snapshot of the pre-call state.  Form -1 (the default): the simulate kernel alone for a call without observations, the simulate
    kernel followed by the observe kernel for a call that wants the last one and the two kernels for a call that wants every one
    when the two-role kernel's workgroups would not all be resident at once ("crowded": PCGRL_ROLLOUT_RESIDENT makes this batch
    count as one that large; test_rollout_by_shape_at_batches_beyond_one_round does it at real sizes), the two-role kernel
    otherwise.  Auto-resets
    inside the launches (short episodes), several calls in a row, then the same under HIP-graph capture (fork / join captured)."""
    kw = dict(change_percentage=0.1)
    seeds = 40 + np.arange(n)
    orc = po.OracleVecEnv(problem, rep, (16, 16), n, seeds=seeds, threads=8, **kw)
    if form == "crowded":
        monkeypatch.setenv("PCGRL_ROLLOUT_RESIDENT", "8")
        form = -1
    env = _vec(problem, rep, (16, 16), n, seeds=seeds, auto_reset=True, **kw)
    monkeypatch.delenv("PCGRL_ROLLOUT_RESIDENT", raising=False)
    assert env._L.pcgrl_set_rollout_form(env._h, form) == 0
    env.reset()
    orc.reset()
    g = torch.Generator().manual_seed(8)
    K = 70
    for call in range(3):
        a = torch.randint(0, env.num_actions, (K, n), generator=g, dtype=torch.int32)
        want = ("all", "none", "last")[call]
        obs, rew, done, stats = env.rollout(a.cuda(), want_obs=want)
        for t in range(K):
            oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=True)
            assert np.array_equal(stats[t].cpu().numpy(), ostats), (call, t)
            assert np.abs(rew[t].cpu().numpy().astype(np.float64) - orew).max() <= REW_TOL and np.array_equal(done[t].cpu().numpy(), odone)
            if want == "all":
                assert np.array_equal(obs[t].cpu().numpy(), oobs), (call, t)
        if want == "last":
            assert np.array_equal(obs.cpu().numpy(), oobs)
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    # captured: the fork into the side stream and the join back are part of the graph
    a = torch.randint(0, env.num_actions, (K, n), generator=g, dtype=torch.int32).cuda()
    obs = torch.zeros((K, n) + env.obs_shape, dtype=torch.uint8, device="cuda")
    rew = torch.zeros((K, n), dtype=torch.float32, device="cuda")
    done = torch.zeros((K, n), dtype=torch.uint8, device="cuda")
    stats = torch.zeros((K, n, env.n_stats), dtype=torch.int32, device="cuda")
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            rc = env._L.pcgrl_rollout(env._h, a.data_ptr(), K, 1, obs.data_ptr(), 0, rew.data_ptr(), done.data_ptr(), stats.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    torch.cuda.current_stream().wait_stream(side)
    for rep_ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for t in range(K):
            oobs, orew, odone, ostats = orc.step(a[t].cpu().numpy(), auto_reset=True)
            assert np.array_equal(stats[t].cpu().numpy(), ostats), (rep_, t)
            assert np.array_equal(obs[t].cpu().numpy(), oobs), (rep_, t)
    env.check_errors()
    big = _vec(problem, rep, (32, 32) if rep != "wide" else (20, 20), 4, seeds=np.arange(4))
    assert big._L.pcgrl_set_rollout_form(big._h, 2) == 2  # EUNSUPPORTED off the 16 x 16 point


@pytest.mark.parametrize("problem,rep,n", [("binary", "narrow", 4608), ("zelda", "turtle", 2304)])
def test_rollout_by_shape_at_batches_beyond_one_round(problem, rep, n):
    """More envs than the two-role rollout kernel holds resident at once (binary: 4096, zelda: 2048 on an MI355X): by shape the
    engine then runs the roles as kernels of their own -- every result the oracle's, whatever the form."""
    seeds = 7 + np.arange(n)
    orc = po.OracleVecEnv(problem, rep, (16, 16), n, seeds=seeds, threads=8, change_percentage=0.1)
    env = _vec(problem, rep, (16, 16), n, seeds=seeds, auto_reset=True, change_percentage=0.1)
    env.reset()
    orc.reset()
    g = torch.Generator().manual_seed(9)
    K = 40
    for want in ("all", "last", "none"):
        a = torch.randint(0, env.num_actions, (K, n), generator=g, dtype=torch.int32)
        obs, rew, done, stats = env.rollout(a.cuda(), want_obs=want)
        for t in range(K):
            oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=True)
            assert np.array_equal(stats[t].cpu().numpy(), ostats), (want, t)
            assert np.array_equal(done[t].cpu().numpy(), odone)
            assert np.abs(rew[t].cpu().numpy().astype(np.float64) - orew).max() <= REW_TOL
            if want == "all":
                assert np.array_equal(obs[t].cpu().numpy(), oobs), (want, t)
        if want == "last":
            assert np.array_equal(obs.cpu().numpy(), oobs)
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    env.check_errors()
