"""Round-5 GPU tests (through the C ABI, against the oracle): the multi-GPU exchange on RCCL with one rank, device-side
action sampling (closed loop), checkpoints across pcgrl_set_static, a captured pcgrl_export_state, the solver-pool
entry points, and handles used concurrently from several host threads / streams (include/pcgrl_amd.h "Conventions")."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pcgrl_oracle as po  # noqa: E402  (checker only)
from conftest import ROOT  # noqa: E402

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


# ------------------------------------------------------------------------------------- RCCL with one rank (SURVEY 8 e)
def test_bench_force_collective_runs_rccl_with_one_rank():
    """bench.py --gpus 1 --force-collective: a world-size-1 "nccl" (= RCCL) process group; the timed region is K launches +
    the pcgrl_reduce_episodes launch + one synchronise while the all-gather + device->host copy of the PREVIOUS interval's
    sums run on a side stream (round 6); step launches replayed from a HIP graph captured while the RCCL watchdog thread is
    alive.  800 warm-up + 800 timed steps of a 770-step episode: each interval's gathered episode count is exactly the batch."""
    out = _bench("--gpus", "1", "--force-collective", "--envs", "256", "--steps", "800", "--warmup", "800", "--no-cpu-baseline",
                 "--rollout-steps", "0", "--closed-loop-steps", "0")
    pr = out["per_rank"]
    assert pr["collective"].startswith("nccl all-gather") and "side stream" in pr["collective"], pr
    assert out["config"]["launch"].startswith("HIP graph"), out["config"]["launch"]
    assert out["episodes"]["episodes"] == 256.0 and pr["episodes"] == [256.0]
    assert out["episodes"]["mean_length"] == 770.0
    tr = out["timed_region"]
    assert tr["exchange"] == "overlap" and tr["previous_interval_sums"][2] == 256.0 and tr["previous_interval_sums"][1] == 256.0 * 770
    assert np.isfinite(pr["exchange_ms"][0]) and pr["exchange_ms"][0] >= 0
    assert tr["wall_ms"] > 0 and 0.9 <= tr["protocol_efficiency_bound"] <= 1.0, tr
    # round 5's region (the exchange behind the launches, as the closing barrier) stays selectable
    ser = _bench("--gpus", "1", "--force-collective", "--exchange", "serial", "--envs", "256", "--steps", "800", "--warmup", "0",
                 "--no-cpu-baseline", "--rollout-steps", "0", "--closed-loop-steps", "0")
    assert ser["episodes"]["episodes"] == 256.0 and ser["per_rank"]["exchange_ms"][0] > 0
    assert ser["timed_region"]["exchange"] == "serial" and "closing barrier" in ser["per_rank"]["collective"]


def test_bench_driver_protocol_with_collective():
    """the driver's own command line (--steps 20 --warmup 5) through the collective path: the whole timed region is one
    HIP graph (20 step launches + the reduction launch) with the previous interval's all-gather on a side stream; the
    region is within a few per cent of the N = 1 region's (today's bar is 5 %; the test allows for
    run-to-run noise of single 160 us regions); the line carries the closed-loop and sub-batch figures; the gcd protocol
    is still selectable"""
    out = _bench("--gpus", "1", "--force-collective", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rollout-steps", "0",
                 "--rllib-adapter", "0", "--closed-loop-steps", "500", "--sub-batches", "2")
    assert out["steps"] == 20 and out["warmup"] == 5
    assert out["config"]["launch"] == "HIP graph of 20 steps per replay"
    assert out["timed_region"]["protocol"].startswith("ONE replay of a HIP graph of 20 step launches + the pcgrl_reduce_episodes")
    assert out["timed_region"]["protocol_efficiency_bound"] >= 0.95, out["timed_region"]
    sb = out["async_sub_batches"]
    assert "error" not in sb and [r["sub_batches"] for r in sb["rows"]] == [1, 2]
    g = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rollout-steps", "0", "--rllib-adapter", "0",
               "--closed-loop-steps", "0", "--sub-batches", "", "--short-protocol", "gcd")
    assert g["timed_region"]["protocol"].startswith("1 untimed + 4 timed replays of one HIP graph of 5 steps")
    # the default N = 1 region (no process group) against the same region with the overlapped exchange: best of three each
    def best(*extra):
        return min(_bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rollout-steps", "0", "--rllib-adapter", "0",
                          "--closed-loop-steps", "0", "--sub-batches", "", *extra)["timed_region"]["wall_ms"] for _ in range(3))
    plain, coll = best(), best("--force-collective")
    assert coll <= 1.15 * plain, (plain, coll)
    cl = out["closed_loop_device_actions"]
    assert "error" not in cl, cl
    assert cl["steps"] == 500 and cl["us_per_step"] > 0
    assert "first_replay_of_one_graph_ms_per_step" in out["per_rank"]
    rf = out["roofline"]
    assert abs(rf["frac"] - rf["algorithmic_bytes_per_launch"] / (out["ms_per_step"] * 1e-3) / 1e9 / rf["peak"]) < 1e-9


# ------------------------------------------------------------------------------------- device-side action sampling
@pytest.mark.parametrize("problem,rep,shape,kw", [("binary", "narrow", (16, 16), {}), ("zelda", "turtle", (16, 16), {}),
                                                  ("sokoban", "wide", (16, 16), {}),
                                                  ("binary", "narrow", (16, 16), dict(act_window=[3, 3])),
                                                  ("minecraft_3D_maze", "narrow", (7, 7, 7), {})])
def test_sample_actions_range_determinism_uniformity(problem, rep, shape, kw):
    n = 1500
    a = _vec(problem, rep, shape, n, seeds=np.arange(n), **kw)
    b = _vec(problem, rep, shape, n, seeds=np.arange(n), **kw)
    n_act = a.spec.n_tiles if a.act_window else a.num_actions
    assert a._L.pcgrl_num_actions(a._h) == n_act
    draws = []
    for k in range(6):
        x, y = a.sample_actions(seed=7), b.sample_actions(seed=7)
        assert x.dtype == torch.int32 and x.numel() == n * a.action_entries
        assert torch.equal(x, y), "same (seed, draw counter) -> same actions on another engine"
        assert int(x.min()) >= 0 and int(x.max()) < n_act
        draws.append(x.clone())
    assert not torch.equal(draws[0], draws[1]), "the draw counter advances"
    assert not torch.equal(a.sample_actions(seed=8), b.sample_actions(seed=9)), "other seed, other actions"
    if n_act <= 16:  # chi-square against the uniform distribution (df <= 15: 99.99 % quantile < 45)
        cnt = torch.bincount(torch.cat([d.flatten() for d in draws]).long(), minlength=n_act).double().cpu().numpy()
        exp = cnt.sum() / n_act
        assert ((cnt - exp) ** 2 / exp).sum() < 60, cnt


@pytest.mark.parametrize("problem,rep,shape,n", [("binary", "narrow", (16, 16), 1024), ("zelda", "turtle", (16, 16), 300),
                                                ("minecraft_3D_maze", "narrow", (7, 7, 7), 64)])
def test_closed_loop_graph_with_device_actions_vs_oracle(problem, rep, shape, n):
    """[pcgrl_sample_actions -> pcgrl_step] pairs captured in ONE HIP graph and replayed: every replay draws new actions;
    the actions the steps consumed (logged by a copy inside the graph) drive the oracle to the same stats / rewards /
    dones / final state, across auto-resets"""
    G, R = 16, 60  # 960 steps: past the 770-step episode of the 16 x 16 maps
    env = _vec(problem, rep, shape, n, seeds=500 + np.arange(n), auto_reset=True)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=500 + np.arange(n), threads=8)
    env.reset()
    orc.reset()
    dev = env.device
    act = torch.zeros(n, dtype=torch.int32, device=dev)
    log = torch.zeros((G, n), dtype=torch.int32, device=dev)
    rew = torch.zeros((G, n), dtype=torch.float32, device=dev)
    done = torch.zeros((G, n), dtype=torch.uint8, device=dev)
    stats = torch.zeros((G, n, env.n_stats), dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(G):
                env.sample_actions(seed=99, out=act)
                log[k].copy_(act)
                _, r, d, _, info = env.step(act)
                rew[k].copy_(r)
                done[k].copy_(d)
                stats[k].copy_(info["stats"])
    torch.cuda.current_stream(dev).wait_stream(side)
    seen = set()
    n_done = 0
    for rep_i in range(R):
        g.replay()
        torch.cuda.synchronize(dev)
        L = log.cpu().numpy()
        key = L.tobytes()
        assert key not in seen, "a replay drew the same actions again"
        seen.add(key)
        for k in range(G):
            _, orew, odone, ostats = orc.step(L[k], auto_reset=True, want_obs=False)
            t = rep_i * G + k
            assert np.array_equal(stats[k].cpu().numpy(), ostats), f"stats @ {t}"
            assert np.max(np.abs(rew[k].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL, f"reward @ {t}"
            assert np.array_equal(done[k].cpu().numpy().astype(bool), odone), f"done @ {t}"
            n_done += int(odone.sum())
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"])
    assert n_done >= (n if shape == (16, 16) else 0)
    env.check_errors()


# ------------------------------------------------------------------------------------- checkpoints
def test_state_dict_survives_set_static_and_eval_mode():
    """pcgrl_set_static moves static_prob / n_static_walls / eval mode at run time (the reference's
    set_static_prob / set_n_static_walls / set_eval_mode, reps/wrappers.py:256-263); a checkpoint taken afterwards loads
    into a fresh engine built from the ORIGINAL config, which takes the exporter's parameters over and continues
    bit-exactly through auto-resets that draw static tiles with them"""
    n = 64
    kw = dict(seeds=900 + np.arange(n), auto_reset=True, change_percentage=0.04, static_prob=0.1, n_static_walls=1)
    env = _vec("binary", "narrow", (16, 16), n, **kw)
    env.reset()
    g = torch.Generator().manual_seed(3)
    acts = torch.randint(0, env.num_actions, (200, n), generator=g, dtype=torch.int32).to(env.device)
    for t in range(30):
        env.step(acts[t])
    env.set_static(static_prob=0.45, n_static_walls=5, eval_mode=True)  # curriculum step + evaluation mode
    for t in range(30, 60):
        env.step(acts[t])
    sd = env.state_dict()
    want = []
    for t in range(60, 200):
        obs, rew, done, _, info = env.step(acts[t])
        want.append((obs.clone(), rew.clone(), done.clone(), info["stats"].clone(), env.get_static().clone()))
    assert sum(int(w[2].sum()) for w in want) > n, "episodes ended after the checkpoint (static tiles were redrawn)"
    other = _vec("binary", "narrow", (16, 16), n, **dict(kw, seeds=np.zeros(n, np.int64)))
    other.reset()
    other.load_state_dict(sd)
    for t in range(60, 200):
        obs, rew, done, _, info = other.step(acts[t])
        w = want[t - 60]
        assert torch.equal(info["stats"], w[3]) and torch.equal(rew, w[1]) and torch.equal(done, w[2]), f"@ {t}"
        assert torch.equal(obs, w[0]), f"obs @ {t}"
        assert torch.equal(other.get_static(), w[4]), f"static mask @ {t}"
    # a masked import leaves the engine-wide parameters alone: a third engine keeps drawing with ITS parameters
    third = _vec("binary", "narrow", (16, 16), n, **dict(kw, seeds=np.zeros(n, np.int64)))
    third.reset()
    m = torch.zeros(n, dtype=torch.uint8)
    m[: n // 2] = 1
    third.load_state_dict(sd, mask=m)
    st = third.get_state()
    assert torch.equal(st.grids[: n // 2], sd["grids"][: n // 2])
    # still built with another problem's / budget's config: refused
    wrong = _vec("binary", "narrow", (16, 16), n, **dict(kw, change_percentage=0.5))
    with pytest.raises(ValueError, match="another config"):
        wrong.load_state_dict(sd)
    other.check_errors()


def test_export_state_captured_in_a_hip_graph():
    """pcgrl_export_state copies its header from pinned memory the engine owns, so the call is capturable; every
    replay writes a complete, importable image of the state at that moment (and of the static parameters in force)"""
    n = 48
    kw = dict(seeds=70 + np.arange(n), auto_reset=True, change_percentage=0.05, static_prob=0.2, n_static_walls=2)
    env = _vec("zelda", "narrow", (16, 16), n, **kw)
    env.reset()
    dev = env.device
    blob = torch.zeros(int(env._L.pcgrl_state_bytes(env._h)), dtype=torch.uint8, device=dev)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = env._L.pcgrl_export_state(env._h, blob.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream)
            assert rc == 0
    torch.cuda.current_stream(dev).wait_stream(side)
    gen = torch.Generator().manual_seed(11)
    acts = torch.randint(0, env.num_actions, (120, n), generator=gen, dtype=torch.int32).to(dev)
    for t in range(40):
        env.step(acts[t])
    g.replay()
    first = blob.clone()
    env.set_static(static_prob=0.5, n_static_walls=4)
    for t in range(40, 80):
        env.step(acts[t])
    g.replay()  # second replay: new state, new header contents
    torch.cuda.synchronize(dev)
    assert not torch.equal(first, blob)
    want = []
    for t in range(80, 120):
        obs, rew, done, _, info = env.step(acts[t])
        want.append((obs.clone(), rew.clone(), done.clone(), info["stats"].clone()))
    other = _vec("zelda", "narrow", (16, 16), n, **dict(kw, seeds=np.zeros(n, np.int64)))
    other.reset()
    other.load_state_dict({"blob": blob, "maybe_stale": 0})
    for t in range(80, 120):
        obs, rew, done, _, info = other.step(acts[t])
        w = want[t - 80]
        assert torch.equal(info["stats"], w[3]) and torch.equal(rew, w[1]) and torch.equal(done, w[2]) and torch.equal(obs, w[0]), f"@ {t}"
    other.check_errors()


# ------------------------------------------------------------------------------------- sokoban solver pool
def _solvable_rooms(n, seed, shape=(16, 16)):
    rng = np.random.default_rng(seed)
    H, W = shape
    g = np.ones((n, H, W), np.uint8)
    for i in range(n):
        h, w = int(rng.integers(2, 6)), int(rng.integers(3, 7))
        y0, x0 = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
        g[i, y0:y0 + h, x0:x0 + w] = 0
        k = int(rng.integers(1, 4)) if h * w >= 8 else 1
        cells = rng.permutation(h * w)[:1 + 2 * k]
        for c, t in zip(cells, [2] + [3] * k + [4] * k):
            g[i, y0 + c // w, x0 + c % w] = t
    return g


def test_solver_pool_reserve_and_report():
    """pcgrl_reserve_solver_pool / pcgrl_solver_pool_slots: the pool can be sized at a moment the caller chooses (no
    synchronous growth inside a later step), the implicit growth can be switched off, and results do not depend on it"""
    n = 1024
    env = _vec("sokoban", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=False, solver_power=1500)
    slots, full, failed = env.solver_pool_slots()
    assert (slots, full, failed) == (64, 256, False)
    g = _solvable_rooms(n, 5)
    want = po.stats_for_grids("sokoban", g, solver_power=1500)
    # (a) implicit growth off: the pool stays at 64 slots however often the solver runs
    assert env.reserve_solver_pool(-1, allow_lazy_growth=False) == 64
    for _ in range(3):
        env.reset(init_grids=torch.as_tensor(g))
        torch.cuda.synchronize()
    assert np.array_equal(env.get_state().stats.cpu().numpy(), want)
    assert env.solver_pool_slots()[0] == 64
    # (b) explicit reservation, full size
    assert env.reserve_solver_pool(0) == 256
    env.reset(init_grids=torch.as_tensor(g))
    assert np.array_equal(env.get_state().stats.cpu().numpy(), want)
    assert env.solver_pool_slots() == (256, 256, False)
    # (c) not for engines without a solver
    b = _vec("binary", "narrow", (16, 16), 4)
    with pytest.raises(ValueError):
        b.reserve_solver_pool(0)
    assert b.solver_pool_slots()[0] == 0
    env.check_errors()


# ------------------------------------------------------------------------------------- b6: threads x streams x handles
def test_two_handles_two_streams_two_host_threads_vs_oracle():
    """include/pcgrl_amd.h "Conventions": one handle is not re-entrant, several handles may coexist.  Two host threads,
    each with its own engine and its own HIP stream, step concurrently: thread A a sokoban batch whose solver FIRES (so
    the lazy pool growth -- a synchronous hipMalloc inside pcgrl_step -- happens while B is launching) with handle-less
    pcgrl_stats_for_grids calls (the hidden shared engines) in between; thread B zelda-turtle with stats_for_grids on
    its own handle in between.  Everything is compared with the oracle afterwards."""
    T = 160
    nA, nB = 512, 700
    rooms = _solvable_rooms(nA, 21)
    extra = _solvable_rooms(200, 22)
    want_extra = po.stats_for_grids("sokoban", extra, solver_power=1200)
    zmaps = np.random.default_rng(4).integers(0, 8, size=(300, 16, 16)).astype(np.uint8)
    want_z = po.stats_for_grids("zelda", zmaps)
    res, errs = {}, []
    start = threading.Barrier(2)

    def run_a():
        try:
            dev = torch.device("cuda:0")
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                env = _vec("sokoban", "narrow", (16, 16), nA, seeds=3000 + np.arange(nA), auto_reset=True, solver_power=1200,
                           change_percentage=0.2)
                env.reset(init_grids=torch.as_tensor(rooms))
                g = torch.Generator().manual_seed(1)
                acts = torch.randint(0, 2, (T, nA), generator=g, dtype=torch.int32)  # floor / wall: rooms stay playable for a while
                out, hl = [], []
                ex = torch.as_tensor(extra).to(dev)
                start.wait()
                for t in range(T):
                    a = acts[t].to(dev)
                    _, rew, done, _, info = env.step(a)
                    out.append((info["stats"].clone(), rew.clone(), done.clone()))
                    if t % 20 == 5:  # handle-less entry point: hidden engines shared between the threads of the process
                        so = torch.empty((len(extra), 7), dtype=torch.int32, device=dev)
                        from control_pcgrl_amd import _lib
                        cfg = env.cfg
                        _lib.check(env._L.pcgrl_stats_for_grids(cfg, len(extra), ex.data_ptr(), so.data_ptr(), 0, s.cuda_stream), "sfg")
                        hl.append(so)
                s.synchronize()
                env.check_errors()
                res["a"] = (acts.numpy(), [(x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy()) for x, y, z in out],
                            [h.cpu().numpy() for h in hl], env.solver_pool_slots())
        except Exception as exc:  # noqa: BLE001
            errs.append(("a", repr(exc)))
            try:
                start.abort()
            except Exception:
                pass

    def run_b():
        try:
            dev = torch.device("cuda:0")
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                env = _vec("zelda", "turtle", (16, 16), nB, seeds=4000 + np.arange(nB), auto_reset=True, change_percentage=0.3)
                env.reset()
                g = torch.Generator().manual_seed(2)
                acts = torch.randint(0, env.num_actions, (T, nB), generator=g, dtype=torch.int32)
                zm = torch.as_tensor(zmaps).to(dev)
                out, own = [], []
                start.wait()
                for t in range(T):
                    a = acts[t].to(dev)
                    _, rew, done, _, info = env.step(a)
                    out.append((info["stats"].clone(), rew.clone(), done.clone()))
                    if t % 16 == 3:
                        own.append(env.stats_for_grids(zm))
                s.synchronize()
                env.check_errors()
                res["b"] = (acts.numpy(), [(x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy()) for x, y, z in out],
                            [h.cpu().numpy() for h in own])
        except Exception as exc:  # noqa: BLE001
            errs.append(("b", repr(exc)))
            try:
                start.abort()
            except Exception:
                pass

    ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
    ta.start(); tb.start()
    ta.join(900); tb.join(900)
    assert not errs, errs
    assert not ta.is_alive() and not tb.is_alive()
    # thread A against the oracle
    acts, out, hl, pool = res["a"]
    orc = po.OracleVecEnv("sokoban", "narrow", (16, 16), nA, seeds=3000 + np.arange(nA), threads=8, solver_power=1200,
                          change_percentage=0.2)
    orc.reset(init_grids=rooms)
    fired = 0
    for t in range(T):
        _, orew, odone, ostats = orc.step(acts[t], auto_reset=True, want_obs=False)
        assert np.array_equal(out[t][0], ostats), f"A stats @ {t}"
        assert np.max(np.abs(out[t][1].astype(np.float64) - orew)) <= REW_TOL and np.array_equal(out[t][2].astype(bool), odone), f"A @ {t}"
        fired += int((ostats[:, 4] != 8192).sum())
    assert fired > 300, "the device solver ran inside the step launches"
    assert pool[0] >= 64 and not pool[2]
    for h in hl:
        assert np.array_equal(h, want_extra)
    assert len(hl) == 8
    # thread B
    acts, out, own = res["b"]
    orc = po.OracleVecEnv("zelda", "turtle", (16, 16), nB, seeds=4000 + np.arange(nB), threads=8, change_percentage=0.3)
    orc.reset()
    for t in range(T):
        _, orew, odone, ostats = orc.step(acts[t], auto_reset=True, want_obs=False)
        assert np.array_equal(out[t][0], ostats), f"B stats @ {t}"
        assert np.max(np.abs(out[t][1].astype(np.float64) - orew)) <= REW_TOL and np.array_equal(out[t][2].astype(bool), odone), f"B @ {t}"
    for h in own:
        assert np.array_equal(h, want_z)
    assert len(own) == 10


def test_one_thread_two_handles_interleaved_on_two_streams():
    """two engines of one process launched alternately on two streams without any synchronisation between them (the
    sub-batch pattern of tools/sub_batch_chains.py): each half equals the oracle's run of its own envs"""
    n, T = 600, 200
    dev = torch.device("cuda:0")
    s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    envs, acts, outs = [], [], [[], []]
    for i in range(2):
        with torch.cuda.stream(s[i]):
            e = _vec("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=100 + 1000 * i + np.arange(n), auto_reset=True)
            e.reset()
            envs.append(e)
        g = torch.Generator().manual_seed(i)
        acts.append(torch.randint(0, 2, (T, n), generator=g, dtype=torch.int32))
    dacts = [a.to(dev) for a in acts]
    torch.cuda.synchronize(dev)
    for t in range(T):
        for i in range(2):
            with torch.cuda.stream(s[i]):
                _, rew, done, _, info = envs[i].step(dacts[i][t])
                outs[i].append((info["stats"].clone(), rew.clone(), done.clone()))
    torch.cuda.synchronize(dev)
    for i in range(2):
        orc = po.OracleVecEnv("minecraft_3D_maze", "narrow", (7, 7, 7), n, seeds=100 + 1000 * i + np.arange(n), threads=8)
        orc.reset()
        for t in range(T):
            _, orew, odone, ostats = orc.step(acts[i][t].numpy(), auto_reset=True, want_obs=False)
            assert np.array_equal(outs[i][t][0].cpu().numpy(), ostats), f"engine {i} stats @ {t}"
            assert np.max(np.abs(outs[i][t][1].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL
            assert np.array_equal(outs[i][t][2].cpu().numpy().astype(bool), odone)
        envs[i].check_errors()


# ------------------------------------------------------------------------------------- sub-batch chains
@pytest.mark.parametrize("problem,rep,shape,n,k", [("binary", "narrow", (40, 48), 96, 4), ("minecraft_3D_maze", "narrow", (7, 7, 7), 128, 2),
                                                  ("zelda", "turtle", (16, 16), 120, 3)])
def test_sub_batched_env_equals_one_batch_vs_oracle(problem, rep, shape, n, k):
    """make_vec_env(..., sub_batches=k): k engines on k streams behind one step(); eager and as k parallel branches of a
    captured HIP graph; double-buffered step_async / wait.  Same results as the oracle's single batch of n envs."""
    from control_pcgrl_amd import make_vec_env
    cfg = {"task": {"problem": problem, "map_shape": list(shape), "obs_window": None, "weights": None}, "representation": rep,
           "change_percentage": 0.1}
    env = make_vec_env(cfg, n, seeds=700 + np.arange(n), sub_batches=k)
    assert env.k == k and len(env.envs) == k
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=700 + np.arange(n), threads=8, change_percentage=0.1)
    obs, _ = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(5)
    T = 150
    acts = torch.randint(0, env.num_actions, (T, n), generator=g, dtype=torch.int32)
    dacts = acts.to(env.device)

    def check(t, obs, rew, done, stats, want_obs):
        oobs, orew, odone, ostats = orc.step(acts[t].numpy(), auto_reset=True, want_obs=want_obs)
        assert np.array_equal(stats.cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL, f"reward @ {t}"
        assert np.array_equal(done.cpu().numpy().astype(bool), odone), f"done @ {t}"
        if want_obs:
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"

    for t in range(50):  # (a) eager, all sub-batches behind one step()
        obs, rew, done, _, info = env.step(dacts[t])
        check(t, obs, rew, done, info["stats"], t % 7 == 0)
    # (b) one HIP graph with k parallel branches, replayed with new actions in the same buffer
    buf = torch.zeros(n, dtype=torch.int32, device=env.device)
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            env.step(buf)
    torch.cuda.current_stream(env.device).wait_stream(side)
    for t in range(50, 100):
        buf.copy_(dacts[t])
        gr.replay()
        obs, rew, done, _, info = env._step_out
        check(t, obs, rew, done, info["stats"], t % 7 == 0)
    # (c) double buffering: sub-batch i steps while the others' results are read
    n_sub = env.n_sub
    for t in range(100, T):
        outs = [env.step_async(i, dacts[t, i * n_sub:(i + 1) * n_sub]) for i in range(k)]
        for i in range(k):
            env.wait(i)
        obs = torch.cat([o[0] for o in outs]); rew = torch.cat([o[1] for o in outs])
        done = torch.cat([o[2] for o in outs]); stats = torch.cat([o[4]["stats"] for o in outs])
        check(t, obs, rew, done, stats, t % 7 == 0)
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"])
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"]) and np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"])
    # checkpoint of the split env into a fresh one with the same split: continues with the oracle
    other = make_vec_env(cfg, n, seeds=np.zeros(n, np.int64), sub_batches=k)
    other.reset()
    other.load_state_dict(env.state_dict())
    a = env.sample_actions(seed=3)
    assert a.shape[0] == n and int(a.max()) < env.num_actions
    for t in range(5):
        act = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        obs, rew, done, _, info = other.step(act.to(env.device))
        oobs, orew, odone, ostats = orc.step(act.numpy(), auto_reset=True)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats) and np.array_equal(obs.cpu().numpy(), oobs), f"restored split env @ {t}"
    env.check_errors()
    other.check_errors()


# ------------------------------------------------------------------------------------- batches beyond 2^31 bytes of output
@pytest.mark.parametrize("problem,rep,n", [("binary", "narrow", 720_000), ("zelda", "turtle", 250_000)])
def test_batch_whose_observations_exceed_2_gib_vs_oracle(problem, rep, n):
    """288 GB of HBM invite large batches: 720 000 binary envs write 2.2 GB of observations per launch (250 000 zelda envs
    2.3 GB), so every byte offset beyond env 699 050 (233 016) needs more than 31 bits.  Reset, a few steps and the
    observations of the LAST envs of the batch against the oracle."""
    seeds = 5 + np.arange(n)
    env = _vec(problem, rep, (16, 16), n, seeds=seeds, auto_reset=True)
    orc = po.OracleVecEnv(problem, rep, (16, 16), n, seeds=seeds, threads=16)
    obs, _ = env.reset()
    oobs = orc.reset()
    tail = slice(n - 3000, n)
    assert np.array_equal(obs[tail].cpu().numpy(), oobs[tail]), "reset observation of the last envs"
    g = torch.Generator().manual_seed(1)
    for t in range(4):
        a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=True)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL
        assert np.array_equal(obs[tail].cpu().numpy(), oobs[tail]), f"observation of the last envs @ {t}"
        assert np.array_equal(obs[:2000].cpu().numpy(), oobs[:2000])
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"])
    env.check_errors()
