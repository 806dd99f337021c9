#!/usr/bin/env python3
"""Differential fuzzer: random supported configurations of the HIP engine against the CPU oracle, step by step.

Every case draws a problem, representation, map shape, observation window, change budget, optional representation
wrappers (static tiles / action patch) and optional controls from the space the engine accepts
(csrc/pcgrl_engine.hip validate_config), runs both sides on the same seeds and actions across auto-resets and compares
stats / reward / done at every step and observations / full state every few steps (bit-exact; rewards to 1e-6 / 1e-9).
A failing case prints its one-line description, which `--case '<json>'` replays.

Checker-side script (it imports oracle/): lives under tests/, is not collected by pytest; tests/test_gpu_parity.py runs
a short fixed-seed sweep of it.  On the GPU box:  python tests/fuzz_parity.py --cases 200 --seed 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

N_TILES = {"binary": 2, "zelda": 8, "sokoban": 5, "minecraft_3D_maze": 2}


STAT_KEYS = {"binary": ["regions", "path-length"],
             "zelda": ["player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length"],
             "sokoban": ["player", "crate", "target", "regions", "dist-win", "sol-length", "ratio"],
             "minecraft_3D_maze": ["regions", "path-length", "n_jump"]}


def draw_case(rng):
    """one random configuration (a JSON-able dict) the engine accepts"""
    case = _draw_case(rng)
    kw = case["kw"]
    if rng.random() < 0.25:
        kw["weights"] = {k: int(rng.integers(0, 6)) for k in STAT_KEYS[case["problem"]]}
    if rng.random() < 0.25:
        kw["max_board_scans"] = int(rng.integers(1, 4))
    case["auto_reset"] = bool(rng.random() < 0.8)
    if kw.get("controls") and rng.random() < 0.35:  # device-side target resampling at every reset (pcgrl_set_target_resampling)
        case["resample"] = int(rng.integers(0, 1 << 40))
    r = rng.random()
    if r < 0.1:
        case["mode"] = "adapter"
    elif r < 0.15:
        case["mode"] = "gym"
    r = rng.random()
    case["seed_kind"] = "same" if r < 0.05 else ("huge" if r < 0.1 else "range")
    # asynchronous stepping (pcgrl_step_ready): sokoban without wrappers / controls, maps whose levels fit one stage workspace
    if (case["problem"] == "sokoban" and not any(k in kw for k in ("controls", "static_prob", "n_static_walls", "act_window"))
            and case["shape"][0] * case["shape"][1] <= 256 and rng.random() < 0.45):
        case["mode"] = "ready"
        case["budget"] = int(rng.choice([1, 3, 8, 16, 64, 400, 100000]))
    return case


def _draw_case(rng):
    problem = str(rng.choice(["binary", "zelda", "sokoban", "minecraft_3D_maze"], p=[0.35, 0.3, 0.15, 0.2]))
    case = dict(problem=problem, kw={})
    kw = case["kw"]
    if problem == "minecraft_3D_maze":
        case["rep"] = "narrow"
        big = rng.random() < 0.25
        hi = 16 if big else 8
        shape = [int(rng.integers(1, hi + 1)) for _ in range(3)]
        if rng.random() < 0.2:  # BASELINE's shape: the compile-time 7^3 / 14^3 kernels
            shape = [7, 7, 7]
        case["shape"] = shape
        if rng.random() < 0.4:  # any window whose volume is a multiple of 4
            ow = [int(rng.integers(1, (3 * s + 6) if rng.random() < 0.1 else (2 * s + 3))) for s in shape]
            ow[2] += (-ow[2]) % 4 if (ow[0] * ow[1] * ow[2]) % 4 else 0
            kw["obs_window"] = ow
        cells = shape[0] * shape[1] * shape[2]
        case["n_envs"] = int(rng.integers(1, max(2, min(200, 40000 // max(cells, 1)))))
        if rng.random() < 0.7 or cells > 600:
            kw["change_percentage"] = float(rng.choice([0.02, 0.05, 0.1, 0.3]))
        if rng.random() < 0.3:
            kw["controls"] = [str(k) for k in rng.choice(["regions", "path-length", "n_jump"], size=int(rng.integers(1, 3)), replace=False)]
        case["steps"] = int(rng.integers(30, 160 if cells <= 600 else 60))
        case["bias"] = bool(rng.random() < 0.5)
        case["mode"] = "mixed" if rng.random() < 0.5 else "step"
        return case
    rep = str(rng.choice(["narrow", "turtle", "wide"], p=[0.45, 0.35, 0.2]))
    case["rep"] = rep
    wmax = 62 if problem == "sokoban" else 64
    r = rng.random()
    if rep == "wide":  # square maps only; obs_window == map_shape
        s = int(rng.integers(1, (wmax if r < 0.2 else 20) + 1))
        shape = [s, s]
    elif r < 0.15:
        shape = [int(rng.integers(1, (62 if problem == "sokoban" else 64) + 1)), int(rng.integers(1, wmax + 1))]
    else:
        shape = [int(rng.integers(1, 25)), int(rng.integers(1, min(wmax, 34) + 1))]
    headline = rng.random() < 0.2  # BASELINE's 16 x 16 maps with the default window: the compile-time (FAST) kernels
    if headline:
        shape = [16, 16]
    case["shape"] = shape
    H, W = shape
    if rep != "wide" and rng.random() < 0.5 and not headline:
        kw["obs_window"] = [int(rng.integers(1, 2 * H + 4)), int(rng.integers(1, 2 * W + 4))]
        if rng.random() < 0.1:  # far beyond the map on every side (rows of padding only)
            kw["obs_window"] = [int(rng.integers(2 * H, 3 * H + 8)), int(rng.integers(2 * W, 3 * W + 8))]
    if rng.random() < 0.5:
        kw["change_percentage"] = float(rng.choice([0.001, 0.05, 0.2, 0.5, 1.0]))
    if rep != "wide" and rng.random() < 0.3:
        kw["static_prob"] = float(rng.choice([0.0, 0.1, 0.3, 0.7, 1.0]))
        kw["n_static_walls"] = int(rng.integers(0, 12 if rng.random() < 0.2 else 6)) if H >= 3 and W >= 3 else 0
        if rng.random() < 0.15:
            kw["static_eval"] = True
    if rep == "narrow" and rng.random() < 0.25:
        kw["act_window"] = [int(rng.integers(1, min(H, 5) + 1)), int(rng.integers(1, min(W, 5) + 1))]
    if rng.random() < 0.25:
        keys = {"binary": ["regions", "path-length"],
                "zelda": ["nearest-enemy", "enemies", "player", "key", "door", "regions", "path-length"],
                "sokoban": ["player", "crate", "ratio", "dist-win", "sol-length", "regions"]}[problem]
        kw["controls"] = [str(k) for k in rng.choice(keys, size=int(rng.integers(1, 3)), replace=False)]
    if problem == "sokoban":
        kw["solver_power"] = int(rng.choice([50, 500, 3000, 10000]))
    cells = H * W
    case["n_envs"] = int(rng.integers(1, max(2, min(600, 60000 // cells))))
    if rng.random() < 0.05:  # several workgroups per CU, partial last workgroup
        case["n_envs"] = int(rng.integers(600, max(601, min(9000, 600000 // cells))))
    case["steps"] = int(rng.integers(30, 200))
    case["bias"] = bool(problem == "binary" and rng.random() < 0.2)
    case["mode"] = "mixed" if rng.random() < 0.5 else "step"
    return case


def random_grids(rng, problem, m, shape):
    """maps of several kinds: uniform noise, sparse noise (long paths), and -- sokoban / zelda -- playable levels (exactly
    one player, matching crates and targets / one key and one door), which is where the solver and the path searches run"""
    n_tiles = N_TILES[problem]
    g = rng.integers(0, n_tiles, size=(m,) + tuple(shape), dtype=np.uint8)
    kind = rng.random()
    if kind < 0.35:
        g = np.where(rng.random(g.shape) < rng.choice([0.5, 0.8, 0.95]), np.uint8(0), g).astype(np.uint8)
    elif kind < 0.7 and problem in ("sokoban", "zelda") and int(np.prod(shape)) >= 6:
        cells = int(np.prod(shape))
        flat = g.reshape(m, cells)
        for i in range(m):
            row = np.where(rng.random(cells) < rng.choice([0.3, 0.6, 0.9]), 0, 1).astype(np.uint8)  # empty / solid
            if problem == "sokoban":  # tiles: 0 empty 1 solid 2 player 3 crate 4 target
                k = int(rng.integers(1, min(4, (cells - 1) // 2) + 1))
                where = rng.permutation(cells)[:1 + 2 * k]
                row[where] = [2] + [3] * k + [4] * k
            else:  # zelda tiles: 0 empty 1 solid 2 player 3 key 4 door 5.. enemies
                ne = int(rng.integers(0, min(4, cells - 3) + 1))
                where = rng.permutation(cells)[:3 + ne]
                row[where] = [2, 3, 4] + [int(rng.integers(5, 8)) for _ in range(ne)]
            flat[i] = row
        g = flat.reshape((m,) + tuple(shape))
    return g


def run_case(case, seed, verbose=False):
    import torch
    import pcgrl_oracle as po  # (checker)
    from control_pcgrl_amd import VecPcgrlEnv

    problem, rep, shape, n, T = case["problem"], case["rep"], tuple(case["shape"]), case["n_envs"], case["steps"]
    kw = {k: (tuple(v) if k == "obs_window" else v) for k, v in case["kw"].items()}
    controls = kw.get("controls")
    auto = bool(case.get("auto_reset", True))
    seeds = seed + np.arange(n)
    if case.get("seed_kind") == "same":
        seeds = np.full(n, seed)
    elif case.get("seed_kind") == "huge":
        seeds = (np.uint64(0xFFFFFFFFFFFFFF00) - np.arange(n, dtype=np.uint64) * np.uint64(0x123456789ABCD)).astype(np.uint64)
    ekw = dict(kw)
    if controls:
        ekw["reward_dtype"] = torch.float64

    def make_engine():
        return VecPcgrlEnv(problem, rep, shape, n, seeds=seeds, auto_reset=auto, **ekw)

    try:
        env = make_engine()
    except ValueError as e:  # a configuration the reference cannot run either: the oracle must refuse it as well
        try:
            po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8, **kw)
        except ValueError:
            return -1
        raise AssertionError(f"the engine refuses ({e}) what the oracle accepts")
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8, **kw)
    rng = np.random.default_rng(seed)
    bounds = po.cond_bounds(problem, shape) if controls else None
    g = torch.Generator().manual_seed(seed)
    tol = 1e-9 if controls else 1e-6
    n_tiles = N_TILES[problem]

    def queue():
        trg = {k: rng.random(n) * (bounds[k][1] - bounds[k][0]) + bounds[k][0] for k in controls}
        env.queue_targets({k: torch.as_tensor(v) for k, v in trg.items()})
        orc.queue_targets(trg)

    # device-side target resampling: the engine draws every env's control targets at each of its resets from a counter-based
    # stream (trg_resampled, csrc/pcgrl_kernels2d.h); the oracle is handed the same values through its queue before every call
    # that may reset an env, and `draws` counts each env's resets
    resample = case.get("resample") if controls and case.get("mode") not in ("adapter", "gym") else None
    draws = np.zeros(n, np.int64)
    M64 = (1 << 64) - 1

    def _mix(z):
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
        return z ^ (z >> 31)

    def next_draws():
        out = {}
        for j, k in enumerate(controls):
            lo, hi = bounds[k]
            vals = np.empty(n)
            for i in range(n):
                r_ = _mix(_mix((resample + int(draws[i]) * 0x9e3779b97f4a7c15) & M64) ^ ((i * 0xd1b54a32d192ed03 + (j + 1) * 0x8cb92ba72f3d8dd7) & M64))
                vals[i] = float(r_ >> 11) * (1.0 / 9007199254740992.0) * (hi - lo) + lo
            out[k] = vals
        return out

    def before_resets():
        if resample is not None:
            orc.queue_targets(next_draws())

    def after_resets(which):
        if resample is not None:
            draws[np.asarray(which, dtype=bool)] += 1

    def draw_actions(t, lead=()):
        size = tuple(lead) + ((n, env.action_entries) if env.action_entries > 1 else (n,))
        if case.get("bias"):
            return (torch.rand(size, generator=g) < 0.5 + 0.45 * np.sin(t / 23.0)).to(torch.int32)
        return torch.randint(0, env.num_actions, size, generator=g, dtype=torch.int32)

    def check_stats(got, want, odone, what):
        if np.array_equal(got, want):
            return True
        bad = np.nonzero((got != want).any(axis=1))[0]
        e = int(bad[0])
        if problem == "sokoban":  # a level with more crates than the device solver holds is REPORTED (error bit 2)
            try:
                env.check_errors()
            except NotImplementedError as ex:
                if "solver" in str(ex):
                    return False
                raise
        grid = orc.get_state()["grids"][e].reshape(shape)
        raise AssertionError(f"stats {what}: {len(bad)} envs, first {e}: got {got[e].tolist()} want {want[e].tolist()} "
                             f"(done {None if odone is None else bool(odone[e])}) oracle grid afterwards:\n{grid}")

    def check_state(what):
        st, ost = env.get_state(), orc.get_state()
        assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"]), f"grids {what}"
        if rep != "wide":
            assert np.array_equal(st.pos.cpu().numpy()[:, :len(shape)], ost["pos"][:, :len(shape)]), f"pos {what}"
        assert np.array_equal(st.iteration.cpu().numpy(), ost["iteration"]), f"iteration {what}"
        assert np.array_equal(st.changes.cpu().numpy(), ost["changes"]), f"changes {what}"
        assert np.allclose(st.ep_return.cpu().numpy(), ost["ep_return"], atol=1e-6), f"ep_return {what}"
        if env.static_tiles:
            assert np.array_equal(env.get_static().cpu().numpy(), orc.static_tiles()), f"static mask {what}"

    def check_ctrl(what):
        if controls:
            assert np.allclose(env.ctrl_obs.cpu().numpy(), orc.ctrl_obs(), rtol=1e-6, atol=1e-7), f"ctrl_obs {what}"

    totals = np.zeros(3 + len(STAT_KEYS[problem]))

    def orc_step(a, want_obs):
        before_resets()
        r = orc.step(a, auto_reset=auto, want_obs=want_obs)
        d = r[2]
        if auto:
            after_resets(d)
        if auto and d.any():  # what pcgrl_reduce_episodes sums: the episodes that ended by auto-reset
            le = orc.last_episode()
            totals[0] += le["ep_return"][d].sum()
            totals[1] += le["ep_len"][d].sum()
            totals[2] += d.sum()
            totals[3:] += le["final_stats"][d].sum(0)
        return r

    if controls:
        queue()
    if resample is not None:
        env.set_target_resampling(True, resample)
    before_resets()
    obs, info = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset()), "reset observation"
    after_resets(np.ones(n, bool))
    check_ctrl("after reset")
    full_every = int(rng.integers(3, 30))
    mixed = case.get("mode") == "mixed"
    t = 0
    trace = case.setdefault("_trace", [])
    cur_static = None
    while t < T:
        if controls and rng.random() < 0.05:
            queue()
        ev = "step"
        if mixed:
            ev = str(rng.choice(["step", "rollout", "update_refresh", "update_step", "swap", "masked_reset", "inject", "observe",
                                 "sfg", "set_static", "graph"], p=[0.26, 0.21, 0.08, 0.08, 0.07, 0.08, 0.09, 0.03, 0.04, 0.03, 0.03]))
            if ev == "set_static" and not env.static_tiles:
                ev = "step"
        what = f"@ {t} ({ev})"
        trace.append(f"{t}:{ev}")
        if ev == "step":
            for _ in range(int(rng.integers(1, 8)) if mixed else 1):
                what = f"@ {t} ({ev})"
                a = draw_actions(t)
                obs, rew, done, _, info = env.step(a.to(env.device))
                want = t % full_every == 0 or t >= T - 1
                oobs, orew, odone, ostats = orc_step(a.numpy(), want)
                if not check_stats(info["stats"].cpu().numpy(), ostats, odone, what):
                    return -2
                assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= tol, f"reward {what}"
                assert np.array_equal(done.cpu().numpy(), odone), f"done {what}"
                check_ctrl(what)
                if want:
                    assert np.array_equal(obs.cpu().numpy(), oobs), f"obs {what}"
                    check_state(what)
                t += 1
        elif ev == "rollout":
            K = int(rng.integers(2, 13)) if rng.random() < 0.85 else int(rng.integers(13, 41))
            want = str(rng.choice(["all", "last", "none"]))
            a = draw_actions(t, lead=(K,))
            # pcgrl_rollout picks its form by map size (one launch / n step launches); both forms stay under test on every shape
            form = str(rng.choice(["auto", "1", "0", "2"]))
            if env._L.pcgrl_set_rollout_form(env._h, -1 if form == "auto" else int(form)) != 0:  # (form 2: 16 x 16 plain configs only)
                form = "auto"
                env._L.pcgrl_set_rollout_form(env._h, -1)
            try:
                obs, rew, done, stats = env.rollout(a.to(env.device), want_obs=want)
            finally:
                env._L.pcgrl_set_rollout_form(env._h, -1)
            case.setdefault("_trace", []).append(f"rollout_form={form}")
            rew, done, stats = rew.cpu().numpy().astype(np.float64), done.cpu().numpy(), stats.cpu().numpy()
            obs = None if obs is None else obs.cpu().numpy()
            for k in range(K):
                oobs, orew, odone, ostats = orc_step(a[k].numpy(), True)
                if not check_stats(stats[k], ostats, odone, f"{what} step {k}/{K}"):
                    return -2
                assert np.max(np.abs(rew[k] - orew)) <= tol, f"reward {what} step {k}/{K}"
                assert np.array_equal(done[k], odone), f"done {what} step {k}/{K}"
                if want == "all":
                    assert np.array_equal(obs[k], oobs), f"obs {what} step {k}/{K}"
            if want == "last":
                assert np.array_equal(obs, oobs), f"last obs {what}"
            check_ctrl(what)
            check_state(what)
            t += K
        elif ev in ("update_refresh", "update_step"):
            for _ in range(int(rng.integers(1, 5))):
                a = draw_actions(t)
                assert np.array_equal(env.update(a.to(env.device)).cpu().numpy(), orc.update(a.numpy())), f"update obs {what}"
            if ev == "update_refresh":
                if not check_stats(env.refresh_stats().cpu().numpy(), orc.refresh_stats(), None, what):
                    return -2
            t += 1
        elif ev == "swap":  # checkpoint into a fresh engine and go on with that one
            sd = env.state_dict()
            env2 = make_engine()
            if cur_static is not None:  # (host-side settings are the caller's to carry over, like the constructor's)
                env2.set_static(static_prob=cur_static[0], n_static_walls=cur_static[1], eval_mode=cur_static[2])
            if resample is not None:  # (engine-wide run-time parameters are the caller's to carry over, too)
                env2.set_target_resampling(True, resample)
            env2.load_state_dict(sd)
            env.check_errors()
            env.close()
            env = env2
            assert np.array_equal(env.observe().cpu().numpy(), orc.observe()), f"obs {what}"
            check_state(what)
            t += 1
        elif ev == "masked_reset":
            mask = (rng.random(n) < rng.random()).astype(np.uint8)
            before_resets()
            obs, _ = env.reset(mask=mask)
            assert np.array_equal(obs.cpu().numpy(), orc.reset(mask=mask)), f"obs {what}"
            after_resets(mask)
            check_ctrl(what)
            check_state(what)
            t += 1
        elif ev == "inject":
            mask = (rng.random(n) < rng.random()).astype(np.uint8)
            grids = random_grids(rng, problem, n, shape)
            pos = np.stack([rng.integers(0, s, size=n) for s in shape], axis=1).astype(np.int32)
            before_resets()
            obs, _ = env.reset(mask=mask, init_grids=grids, init_pos=pos)
            assert np.array_equal(obs.cpu().numpy(), orc.reset(mask=mask, init_grids=grids, init_pos=pos)), f"obs {what}"
            after_resets(mask)
            if not check_stats(env.get_state().stats.cpu().numpy(), orc.get_state()["stats"], None, what):
                return -2
            check_state(what)
            t += 1
        elif ev == "graph":  # pcgrl_step captured in a HIP graph (static action buffer), replayed with fresh actions
            static_a = draw_actions(t).to(env.device)
            env.step(static_a)  # (eager warm-up; the oracle follows)
            oobs, orew, odone, ostats = orc_step(static_a.cpu().numpy(), False)
            graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    obs, rew, done, _, info = env.step(static_a)
            torch.cuda.current_stream().wait_stream(side)
            R = int(rng.integers(2, 12))
            for r in range(R):
                a = draw_actions(t + r)
                static_a.copy_(a)
                graph.replay()
                oobs, orew, odone, ostats = orc_step(a.numpy(), True)
                if not check_stats(info["stats"].cpu().numpy(), ostats, odone, f"{what} replay {r}/{R}"):
                    return -2
                assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= tol, f"reward {what} replay {r}/{R}"
                assert np.array_equal(done.cpu().numpy(), odone), f"done {what} replay {r}/{R}"
                assert np.array_equal(obs.cpu().numpy(), oobs), f"obs {what} replay {r}/{R}"
            check_ctrl(what)
            check_state(what)
            del graph
            t += R + 1
        elif ev == "sfg":  # pcgrl_stats_for_grids: any batch size, its own scratch engine
            m = int(rng.integers(1, 300))
            grids = random_grids(rng, problem, m, shape)
            got = env.stats_for_grids(torch.as_tensor(grids).to(env.device)).cpu().numpy()
            want_s = po.stats_for_grids(problem, grids, solver_power=kw.get("solver_power", 10000))
            if not np.array_equal(got, want_s):
                bad = np.nonzero((got != want_s).any(axis=1))[0]
                if problem == "sokoban":
                    try:
                        env.check_errors()
                    except NotImplementedError as ex:
                        if "solver" in str(ex):
                            return -2
                        raise
                raise AssertionError(f"stats_for_grids {what}: {len(bad)} of {m} grids, first {int(bad[0])}: got "
                                     f"{got[bad[0]].tolist()} want {want_s[bad[0]].tolist()}\n{grids[bad[0]]}")
            t += 1
        elif ev == "set_static":  # takes effect at the next reset
            sp, nw = float(rng.choice([0.0, 0.2, 0.6])), int(rng.integers(0, 4)) if min(shape) >= 3 else 0
            ev_mode = bool(rng.random() < 0.3)
            cur_static = (sp, nw, ev_mode)
            env.set_static(static_prob=sp, n_static_walls=nw, eval_mode=ev_mode)
            orc.set_static(static_prob=sp, n_static_walls=nw, eval_mode=ev_mode)
            t += 1
        else:
            assert np.array_equal(env.observe().cpu().numpy(), orc.observe()), f"obs {what}"
            t += 1
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"]), "episode counts"
    assert np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"]), "final stats"
    assert np.array_equal(le.ep_len.cpu().numpy(), ole["ep_len"]), "episode lengths"
    if auto:
        red = env.reduce_episodes().cpu().numpy()
        assert np.allclose(red, totals, rtol=1e-12, atol=1e-6), f"reduce_episodes {red.tolist()} vs {totals.tolist()}"
    try:
        env.check_errors()
    except NotImplementedError as ex:  # > 128 crates in a level whose statistics happened to agree anyway
        if problem == "sokoban" and "solver" in str(ex):
            return -2
        raise
    return int(ole["n_episodes"].sum())


def run_ready_case(case, seed):
    """sokoban through pcgrl_set_solver_budget / pcgrl_step_ready: the device solver works to a budget per launch and parks
    what it could not finish; busy envs ignore their actions.  The oracle steps an env exactly when the engine reports an
    emitted transition, with the action the env consumed; resets (masked, injected playable levels) abandon steps in flight;
    a checkpoint into a fresh engine loses the parked searches (they restart) but not the pending steps."""
    import torch
    import pcgrl_oracle as po  # (checker)
    from control_pcgrl_amd import VecPcgrlEnv

    problem, rep, shape, n, T = case["problem"], case["rep"], tuple(case["shape"]), case["n_envs"], case["steps"]
    kw = {k: (tuple(v) if k == "obs_window" else v) for k, v in case["kw"].items()}
    auto, budget = bool(case.get("auto_reset", True)), int(case["budget"])
    seeds = seed + np.arange(n)

    def make_engine():
        e = VecPcgrlEnv(problem, rep, shape, n, seeds=seeds, auto_reset=auto, **kw)
        e.set_solver_budget(budget)
        return e

    env = make_engine()
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8, **kw)
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    obs, _ = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset()), "reset observation"
    busy = env.env_busy().cpu().numpy().astype(bool)
    pend, has_pend = np.zeros(n, np.int32), np.zeros(n, bool)
    trace = case.setdefault("_trace", [])
    emitted_total, t = 0, 0
    while t < T:
        ev = str(rng.choice(["step", "inject", "masked_reset", "swap"], p=[0.8, 0.1, 0.05, 0.05]))
        what = f"@ {t} ({ev})"
        trace.append(f"{t}:{ev}")
        if ev == "step":
            for _ in range(int(rng.integers(1, 12))):
                a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32).numpy()
                if rng.random() < 0.5:  # mostly floor / wall edits: playable levels stay playable for a while
                    a = (a // 5) * 5 + (a % 2)
                consume = ~busy
                pend[consume] = a[consume]
                has_pend[consume] = True
                obs, rew, done, _, info = env.step_ready(torch.as_tensor(a).to(env.device))
                status = info["status"].cpu().numpy()
                emitted = (status & 1) != 0
                assert not (emitted & ~has_pend).any(), f"a transition without a consumed action {what}"
                oobs, orew, odone, ostats = orc.step_masked(emitted, pend, auto_reset=auto)
                if emitted.any():
                    got = info["stats"].cpu().numpy()
                    if not np.array_equal(got[emitted], ostats[emitted]):
                        e = int(np.nonzero(emitted & (got != ostats).any(axis=1))[0][0])
                        raise AssertionError(f"stats {what}: env {e} got {got[e].tolist()} want {ostats[e].tolist()} (budget {budget})")
                    assert np.abs(rew.cpu().numpy()[emitted] - orew[emitted]).max() <= 1e-6, f"reward {what}"
                    assert np.array_equal(done.cpu().numpy()[emitted], odone[emitted]), f"done {what}"
                    assert np.array_equal(obs.cpu().numpy()[emitted], oobs[emitted]), f"obs {what}"
                has_pend[emitted] = False
                busy = (status & 2) != 0
                assert not (~busy & has_pend).any(), f"an idle env still owes a transition {what}"
                emitted_total += int(emitted.sum())
                t += 1
        elif ev in ("inject", "masked_reset"):
            mask = (rng.random(n) < rng.random()).astype(np.uint8)
            if ev == "inject":
                grids = random_grids(rng, problem, n, shape)
                pos = np.stack([rng.integers(0, s_, size=n) for s_ in shape], axis=1).astype(np.int32)
                obs, _ = env.reset(mask=mask, init_grids=grids, init_pos=pos)
                want = orc.reset(mask=mask, init_grids=grids, init_pos=pos)
            else:
                obs, _ = env.reset(mask=mask)
                want = orc.reset(mask=mask)
            assert np.array_equal(obs.cpu().numpy(), want), f"obs {what}"
            has_pend[mask != 0] = False  # steps in flight are abandoned with the old map
            busy = env.env_busy().cpu().numpy().astype(bool)
            t += 1
        else:  # checkpoint into a fresh engine: pending steps travel, parked searches do not (they restart: same results)
            sd = env.state_dict()
            env2 = make_engine()
            env2.load_state_dict(sd)
            env.check_errors()
            env.close()
            env = env2
            assert np.array_equal(env.env_busy().cpu().numpy().astype(bool), busy), f"busy flags {what}"
            assert np.array_equal(env.observe().cpu().numpy(), orc.observe()), f"obs {what}"
            t += 1
        st, ost = env.get_state(), orc.get_state()
        assert np.array_equal(st.grids.cpu().numpy().reshape(n, -1), ost["grids"]), f"grids {what}"
        idle = ~busy
        assert np.array_equal(st.stats.cpu().numpy()[idle], ost["stats"][idle]), f"state stats of idle envs {what}"
        assert np.array_equal(st.iteration.cpu().numpy(), ost["iteration"]), f"iteration {what}"
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"]), "episode counts"
    assert np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"]), "final stats"
    env.check_errors()
    env.close()
    return emitted_total


def _final_check(vec, problem):
    """pcgrl_poll_error at the end of a case; False: a sokoban level beyond the device solver's 128 crates was met
    (reported, the statistics of that step kept their solver-less values)"""
    try:
        vec.check_errors()
    except NotImplementedError as ex:
        if problem == "sokoban" and "solver" in str(ex):
            return False
        raise
    return True


def run_adapter_case(case, seed):
    """the same configuration through PcgrlVectorEnv (ray.rllib VectorEnv call shape): vector_reset / vector_step / reset_at
    per finished env, per-env targets through the sub-env handles, against the oracle without auto-reset"""
    import pcgrl_oracle as po  # (checker)
    from control_pcgrl_amd import PcgrlVectorEnv

    problem, rep, shape, T = case["problem"], case["rep"], tuple(case["shape"]), min(case["steps"], 80)
    n = min(case["n_envs"], 48)
    kw = {k: (tuple(v) if k == "obs_window" else v) for k, v in case["kw"].items() if k not in ("solver_power", "static_eval")}
    controls = kw.get("controls") or []
    seeds = seed + np.arange(n)
    cfg = {"task": {"problem": problem, "map_shape": list(shape), "obs_window": kw.get("obs_window"), "weights": kw.get("weights")},
           "representation": rep, "change_percentage": kw.get("change_percentage"), "max_board_scans": kw.get("max_board_scans", 3),
           "controls": controls or None, "act_window": kw.get("act_window"), "static_prob": kw.get("static_prob"),
           "n_static_walls": kw.get("n_static_walls")}
    try:
        env = PcgrlVectorEnv(cfg, num_envs=n, seeds=[int(x) for x in seeds])
    except ValueError:
        try:
            po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, **kw)
        except ValueError:
            return -1
        raise AssertionError("the adapter refuses what the oracle accepts")
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8, **kw)
    rng = np.random.default_rng(seed)
    K2 = 2 * len(controls)
    bounds = po.cond_bounds(problem, shape) if controls else None
    subs = env.get_sub_environments()
    keys = STAT_KEYS[problem]

    def new_targets(envs):
        for i in envs:
            trg = {k: float(rng.random() * (bounds[k][1] - bounds[k][0]) + bounds[k][0]) for k in controls}
            subs[int(i)].set_trgs(trg)
            m = np.zeros(n, np.uint8)
            m[int(i)] = 1
            orc.queue_targets(trg, mask=m)

    def check_obs(o, want, ctrl, what):
        assert np.array_equal(np.asarray(o)[..., K2:].astype(np.uint8), want), f"obs {what}"
        if K2:
            planes = np.asarray(o)[..., :K2].reshape(-1, K2)
            assert np.all(planes == planes[0]) and np.allclose(planes[0], ctrl, rtol=1e-6, atol=1e-7), f"control planes {what}"

    if controls:
        new_targets(range(n))
    obs, infos = env.vector_reset()
    oobs = orc.reset()
    octrl = orc.ctrl_obs() if controls else [None] * n
    for i in range(n):
        check_obs(obs[i], oobs[i], octrl[i], f"after vector_reset env {i}")
    entries = env.vec.action_entries
    hi = env.vec.spec.n_tiles if entries > 1 else env.vec.num_actions
    for t in range(T):
        a = rng.integers(0, hi, size=(n, entries) if entries > 1 else (n,))
        obs, rew, term, trunc, infos = env.vector_step([x for x in a] if entries > 1 else a.tolist())
        oobs, orew, odone, ostats = orc.step(a, auto_reset=False, want_obs=True)
        octrl = orc.ctrl_obs() if controls else [None] * n
        assert term == odone.tolist() and trunc == term, f"done @ {t}"
        assert np.max(np.abs(np.asarray(rew, np.float64) - orew)) <= 1e-5 * max(1.0, float(np.max(np.abs(orew)))), f"reward @ {t}"
        for i in range(n):
            check_obs(obs[i], oobs[i], octrl[i], f"@ {t} env {i}")
        for i in rng.integers(0, n, size=3):
            info = infos[int(i)]
            assert [info[k] for k in keys] == ostats[int(i)].tolist(), f"info @ {t} env {i}"
            assert subs[int(i)].unwrapped._rep_stats == dict(zip(keys, ostats[int(i)].tolist())), f"sub-env stats @ {t}"
        fin = np.nonzero(odone)[0]
        if len(fin):
            if controls:
                new_targets(fin)
            m = np.zeros(n, np.uint8)
            m[fin] = 1
            oobs = orc.reset(mask=m)
            octrl = orc.ctrl_obs() if controls else [None] * n
            for i in rng.permutation(fin):
                o, info = env.reset_at(int(i))
                assert info == {}
                check_obs(o, oobs[int(i)], octrl[int(i)], f"reset_at({i}) @ {t}")
    ok = _final_check(env.vec, problem)
    env.close()
    return int(orc.last_episode()["n_episodes"].sum()) if ok else -2


def run_gym_case(case, seed):
    """the same configuration as ONE env behind make_env(cfg), the reference's gym call shape (rl/envs.py:28-81)"""
    from types import SimpleNamespace as NS
    import pcgrl_oracle as po  # (checker)
    from control_pcgrl_amd import make_env

    problem, rep, shape, T = case["problem"], case["rep"], tuple(case["shape"]), min(case["steps"], 60)
    kw = {k: (tuple(v) if k == "obs_window" else v) for k, v in case["kw"].items() if k not in ("solver_power", "static_eval")}
    controls = kw.get("controls") or []
    cfg = NS(representation=rep, max_board_scans=kw.get("max_board_scans", 3), change_percentage=kw.get("change_percentage"),
             controls=controls or None, act_window=kw.get("act_window"), static_prob=kw.get("static_prob"),
             n_static_walls=kw.get("n_static_walls"),
             task=NS(problem=problem, map_shape=shape, obs_window=kw.get("obs_window"), weights=kw.get("weights")),
             multiagent=NS(n_agents=0))
    try:
        env = make_env(cfg)
    except ValueError:
        try:
            po.OracleVecEnv(problem, rep, shape, 1, seeds=[seed], **kw)
        except ValueError:
            return -1
        raise AssertionError("make_env refuses what the oracle accepts")
    env.unwrapped.seed(int(seed))
    orc = po.OracleVecEnv(problem, rep, shape, 1, seeds=[seed], **kw)
    rng = np.random.default_rng(seed)
    K2 = 2 * len(controls)
    bounds = po.cond_bounds(problem, shape) if controls else None
    keys = STAT_KEYS[problem]

    def new_targets():
        trg = {k: float(rng.random() * (bounds[k][1] - bounds[k][0]) + bounds[k][0]) for k in controls}
        env.set_trgs(trg)
        orc.queue_targets(trg)

    def check_obs(o, want, what):
        assert o.dtype == np.float32 and np.array_equal(o[..., K2:].astype(np.uint8), want), f"obs {what}"
        if K2:
            planes = o[..., :K2].reshape(-1, K2)
            assert np.all(planes == planes[0]) and np.allclose(planes[0], orc.ctrl_obs()[0], rtol=1e-6, atol=1e-7), f"control planes {what}"

    if controls:
        new_targets()
    obs, info = env.reset()
    assert info == {}
    check_obs(obs, orc.reset()[0], "after reset")
    entries = env._vec.action_entries
    hi = env._vec.spec.n_tiles if entries > 1 else env._vec.num_actions
    for t in range(T):
        a = rng.integers(0, hi, size=(1, entries) if entries > 1 else (1,))
        obs, r, d, tr, info = env.step(a[0] if entries > 1 else int(a[0]))
        oobs, orew, odone, ostats = orc.step(a, auto_reset=False, want_obs=True)
        check_obs(obs, oobs[0], f"@ {t}")
        assert abs(r - orew[0]) <= 1e-5 * max(1.0, abs(orew[0])) and d == bool(odone[0]) and tr == d, f"reward / done @ {t}"
        assert [info[k] for k in keys] == ostats[0].tolist() and env.unwrapped._rep_stats == dict(zip(keys, ostats[0].tolist())), f"stats @ {t}"
        ost = orc.get_state()
        assert info["iterations"] == int(ost["iteration"][0]) and info["changes"] == int(ost["changes"][0]), f"counters @ {t}"
        if d:
            if controls:
                new_targets()
            obs, info = env.reset()
            check_obs(obs, orc.reset()[0], f"reset after {t}")
    ok = _final_check(env._vec, problem)
    env.close()
    return int(orc.last_episode()["n_episodes"].sum()) if ok else -2


def sweep(n_cases, seed, verbose=True, stop_on_fail=True, budget_s=None, only=None):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    failures, refused = [], []
    for i in range(n_cases):
        case = draw_case(rng)
        cs = int(rng.integers(0, 1 << 30))
        if only and case["problem"] not in only:
            continue
        line = json.dumps(dict(case, seed=cs))
        t1 = time.time()
        try:
            eps = {"adapter": run_adapter_case, "gym": run_gym_case, "ready": run_ready_case}.get(case.get("mode"), run_case)(case, cs)
            case.pop("_trace", None)
            if verbose:
                print(f"ok   {i:4d} {time.time() - t1:6.1f}s eps={eps:5d} {line}", flush=True)
        except NotImplementedError as e:
            # The engine REFUSED (PCGRL_EUNSUPPORTED: a stated limit of the device solver / the 3-D search was met at run time,
            # reported, never silent).  Not a parity failure -- nothing wrong was handed out -- and counted on its own: the
            # generator only draws configurations pcgrl_create accepts, so a refusal is a limit worth knowing about.
            refused.append((line, str(e)))
            print(f"REFUSED {i:4d} {line}\n     {e}", flush=True)
        except AssertionError as e:
            failures.append((line, str(e)))
            print(f"FAIL {i:4d} {line}\n     {e}\n     events: {' '.join(case.get('_trace', [])[-12:])}", flush=True)
            if stop_on_fail:
                break
        if budget_s is not None and time.time() - t0 > budget_s:
            print(f"time budget reached after {i + 1} cases", flush=True)
            break
    sweep.refused = refused  # (read by the command line below and by the tests)
    return failures


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--budget-s", type=float, default=None)
    ap.add_argument("--keep-going", action="store_true")
    ap.add_argument("--case", type=str, default=None, help="replay one case (the JSON a failure printed)")
    ap.add_argument("--dry", action="store_true", help="only print the drawn cases (no GPU needed)")
    ap.add_argument("--only", type=str, default=None, help="comma-separated problems to keep (the others are drawn and skipped)")
    ap.add_argument("--max-refused", type=int, default=-1,
                    help="refusals (PCGRL_EUNSUPPORTED at run time) tolerated before the campaign FAILS (exit code 2); default: "
                         "max(2, cases / 500) -- since round 4 nothing the generator draws is refused, so a kernel that newly reports "
                         "ring overflows or solver limits on configurations it used to handle must not pass as '0 failures'")
    a = ap.parse_args()
    if a.dry:
        r = np.random.default_rng(a.seed)
        for _ in range(a.cases):
            c = draw_case(r)
            print(json.dumps(dict(c, seed=int(r.integers(0, 1 << 30)))))
        sys.exit(0)
    if a.case:
        c = json.loads(a.case)
        s = c.pop("seed")
        try:
            print("episodes:", {"adapter": run_adapter_case, "gym": run_gym_case, "ready": run_ready_case}.get(c.get("mode"), run_case)(c, s))
        finally:
            print("events:", " ".join(c.get("_trace", [])))
        sys.exit(0)
    f = sweep(a.cases, a.seed, stop_on_fail=not a.keep_going, budget_s=a.budget_s, only=a.only.split(",") if a.only else None)
    limit = a.max_refused if a.max_refused >= 0 else max(2, a.cases // 500)
    print(f"{len(f)} failure(s), {len(sweep.refused)} refused (reported limits, see REFUSED lines; tolerated: {limit})")
    sys.exit(1 if f else (2 if len(sweep.refused) > limit else 0))
