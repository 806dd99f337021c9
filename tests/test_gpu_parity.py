"""GPU parity: the HIP engine (through the C ABI, via control_pcgrl_amd) against
  (a) golden vectors captured from the reference (tests/golden/), and
  (b) the CPU oracle on identical seeded inputs, at BASELINE sizes.
Bars: bit-exact grids / positions / counters / stats / done / observations; |reward - reference| <= 1e-6
(the engine returns float32; the values are integers, so the comparison is in fact exact).
Run on the GPU box:  python -m pytest tests -m gpu
"""
import glob
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pcgrl_oracle as po  # noqa: E402  (checker only)
from conftest import GOLDEN  # noqa: E402

EPISODES = [p for p in sorted(glob.glob(os.path.join(GOLDEN, "episode_*.npz"))) if "mc3d" not in p]
REW_TOL = 1e-6


def _vec(*a, **k):
    from control_pcgrl_amd import VecPcgrlEnv
    return VecPcgrlEnv(*a, **k)


@pytest.mark.parametrize("path", EPISODES, ids=[os.path.basename(p)[8:-4] for p in EPISODES])
def test_golden_episode_replay(path):
    z = np.load(path)
    problem, rep = str(z["problem"]), str(z["representation"])
    shape = tuple(int(s) for s in z["map_shape"])
    env = _vec(problem, rep, shape, 1, seeds=[int(z["seed"])], auto_reset=False)
    T, ep_len = len(z["action"]), int(z["episode_len"])
    obs_steps = {int(s): i for i, s in enumerate(z["obs_steps"])}

    def check_reset(k):
        obs, _ = env.reset()
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["reset_grid"][k]), "reset grid (RNG stream)"
        if rep != "wide":
            assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["reset_pos"][k])
        assert np.array_equal(st.stats[0].cpu().numpy(), z["reset_stats"][k])
        assert np.array_equal(obs[0].cpu().numpy().ravel(), z["reset_obs"][k])

    check_reset(0)
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for t in range(T):
        obs, rew, done, _, info = env.step(acts[t:t + 1])
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["grid"][t]), f"grid @ {t}"
        if rep != "wide":
            assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["pos"][t]), f"pos @ {t}"
        got = info["stats"][0].cpu().numpy()
        assert np.array_equal(got, z["stats"][t]), f"stats @ {t}: {got} vs {z['stats'][t]}"
        assert abs(float(rew[0]) - z["reward"][t]) <= REW_TOL, f"reward @ {t}"
        assert bool(done[0]) == bool(z["done"][t]), f"done @ {t}"
        assert int(st.changes[0]) == z["changes"][t] and int(st.iteration[0]) == z["iterations"][t]
        o = obs[0].cpu().numpy()
        assert zlib.crc32(o.tobytes()) == int(z["obs_crc"][t]), f"obs crc @ {t}"
        if t in obs_steps:
            assert np.array_equal(o.ravel(), z["obs_full"][obs_steps[t]])
        if t == ep_len - 1:
            check_reset(1)
    env.check_errors()


@pytest.mark.parametrize("problem,fname", [("binary", "stats_binary.npz"), ("zelda", "stats_zelda.npz"),
                                           ("sokoban", "stats_sokoban.npz"),
                                           ("sokoban", "stats_sokoban_solver.npz")])
def test_golden_stats_known_answers(problem, fname):
    z = np.load(os.path.join(GOLDEN, fname))
    env = _vec(problem, "narrow", z["grids"].shape[1:], 1, auto_reset=False)
    got = env.stats_for_grids(torch.as_tensor(z["grids"])).cpu().numpy()
    bad = np.nonzero((got != z["stats"]).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} grids differ, first {bad[:5]}: got {got[bad[:3]]} want {z['stats'][bad[:3]]}"


def _rollout_vs_oracle(problem, rep, shape, n_envs, n_steps, seed0=100, full_every=97, threads=8, **kw):
    env = _vec(problem, rep, shape, n_envs, seeds=seed0 + np.arange(n_envs), auto_reset=True, **kw)
    orc = po.OracleVecEnv(problem, rep, shape, n_envs, seeds=seed0 + np.arange(n_envs), threads=threads, **kw)
    obs, _ = env.reset()
    oobs = orc.reset()
    assert np.array_equal(obs.cpu().numpy(), oobs), "reset observation"
    g = torch.Generator(device="cpu").manual_seed(seed0)
    n_done = 0
    for t in range(n_steps):
        a = torch.randint(0, env.num_actions, (n_envs, env.action_entries) if env.action_entries > 1 else (n_envs,),
                          generator=g, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        want_obs = (t % full_every == 0) or t == n_steps - 1
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=want_obs)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL, f"reward @ {t}"
        assert np.array_equal(done.cpu().numpy(), odone), f"done @ {t}"
        n_done += int(odone.sum())
        if want_obs:
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"
            st, ost = env.get_state(), orc.get_state()
            assert np.array_equal(st.grids.cpu().numpy().reshape(n_envs, -1), ost["grids"]), f"grids @ {t}"
            if rep != "wide":
                assert np.array_equal(st.pos.cpu().numpy()[:, :len(shape)], ost["pos"][:, :len(shape)])
            assert np.array_equal(st.iteration.cpu().numpy(), ost["iteration"])
            assert np.array_equal(st.changes.cpu().numpy(), ost["changes"])
            assert np.allclose(st.ep_return.cpu().numpy(), ost["ep_return"], atol=REW_TOL)
            if env.static_tiles:
                assert np.array_equal(env.get_static().cpu().numpy(), orc.static_tiles()), f"static mask @ {t}"
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"])
    assert np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"])
    assert np.array_equal(le.ep_len.cpu().numpy(), ole["ep_len"])
    assert np.allclose(le.ep_return.cpu().numpy(), ole["ep_return"], atol=REW_TOL)
    env.check_errors()
    return n_done


MC3D_EPISODES = sorted(glob.glob(os.path.join(GOLDEN, "episode_mc3dmaze_*.npz")))


@pytest.mark.parametrize("path", MC3D_EPISODES, ids=[os.path.basename(p)[8:-4] for p in MC3D_EPISODES])
def test_golden_mc3dmaze_episode_replay(path):
    """BASELINE configs[4] against the reference's ControlWrapper(PcgrlEnv3D) trace (the reference's image wrappers
    crash on this problem): grid, pos, stats, reward, done bit-exact; the observation is checked against a
    re-derivation from the reference's obs dict (map with path overlay + pos)."""
    z = np.load(path)
    shape = tuple(int(s) for s in z["map_shape"])
    env = _vec("minecraft_3D_maze", "narrow", shape, 1, seeds=[int(z["seed"])], auto_reset=False)
    T, ep_len = len(z["action"]), int(z["episode_len"])

    def expected_obs(overlay_map, pos):
        m = overlay_map.reshape(shape).astype(np.int64) + 1
        ow = tuple(2 * s for s in shape)
        padded = np.pad(m, [(w // 2, w // 2) for w in ow], constant_values=0)
        sl = tuple(slice(int(p), int(p) + w) for p, w in zip(pos, ow))
        return np.eye(4, dtype=np.uint8)[padded[sl]]

    def check_reset(k):
        obs, _ = env.reset()
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["reset_grid"][k]), "reset grid (RNG stream)"
        assert np.array_equal(st.pos[0].cpu().numpy(), z["reset_pos"][k])
        assert np.array_equal(st.stats[0].cpu().numpy(), z["reset_stats"][k])
        assert np.array_equal(obs[0].cpu().numpy(), expected_obs(z["reset_obs"][k], z["reset_pos"][k]))

    check_reset(0)
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for t in range(T):
        obs, rew, done, _, info = env.step(acts[t:t + 1])
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["grid"][t]), f"grid @ {t}"
        assert np.array_equal(st.pos[0].cpu().numpy(), z["pos"][t]), f"pos @ {t}"
        got = info["stats"][0].cpu().numpy()
        assert np.array_equal(got, z["stats"][t]), f"stats @ {t}: {got} vs {z['stats'][t]}"
        assert abs(float(rew[0]) - z["reward"][t]) <= REW_TOL, f"reward @ {t}"
        assert bool(done[0]) == bool(z["done"][t]), f"done @ {t}"
        assert np.array_equal(obs[0].cpu().numpy(), expected_obs(z["overlay"][t], z["pos"][t])), f"obs/overlay @ {t}"
        if t == ep_len - 1:
            check_reset(1)
    env.check_errors()


def test_golden_mc3dmaze_stats_known_answers():
    z = np.load(os.path.join(GOLDEN, "stats_mc3dmaze.npz"))
    env = _vec("minecraft_3D_maze", "narrow", (7, 7, 7), 1, auto_reset=False)
    got = env.stats_for_grids(torch.as_tensor(z["grids"])).cpu().numpy()
    bad = np.nonzero((got != z["stats"]).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} grids differ, first {bad[:5]}: got {got[bad[:3]]} want {z['stats'][bad[:3]]}"


def test_mc3dmaze_1024_envs_vs_oracle():
    """BASELINE configs[4]: minecraft_3D_maze-narrow 7x7x7, 1024 envs/GPU, across the auto-reset (1031 steps)."""
    n_done = _rollout_vs_oracle("minecraft_3D_maze", "narrow", (7, 7, 7), 1024, 1050, full_every=211)
    assert n_done == 1024


def test_mc3dmaze_other_shapes_vs_oracle():
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (5, 6, 7), 33, 300, full_every=41)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (8, 7, 9), 17, 300, full_every=41)


SHAPES3D = sorted(glob.glob(os.path.join(GOLDEN, "shape3d_*.npz")))


def _expected_obs_3d(shape, overlay_map, pos):
    m = overlay_map.reshape(shape).astype(np.int64) + 1
    ow = tuple(2 * s for s in shape)
    padded = np.pad(m, [(w // 2, w // 2) for w in ow], constant_values=0)
    sl = tuple(slice(int(p), int(p) + w) for p, w in zip(pos, ow))
    return np.eye(4, dtype=np.uint8)[padded[sl]]


@pytest.mark.parametrize("path", SHAPES3D, ids=[os.path.basename(p)[8:-4] for p in SHAPES3D])
def test_golden_shape3d_episode_replay(path):
    """minecraft_3D_maze at the reference's stock map size 15 x 15 x 15 (configs/config.py:153-157) and at 10 x 10 x 10
    against reference episodes: reset from the seed alone, every step's grid, position, stats, reward, done, counters;
    the observation (incl. the path overlay) where the fixture holds the reference's obs dict in full."""
    z = np.load(path)
    shape = tuple(int(s) for s in z["map_shape"])
    env = _vec("minecraft_3D_maze", "narrow", shape, 1, seeds=[int(z["seed"])], auto_reset=False,
               change_percentage=float(z["change_percentage"]))
    T, ep_len = len(z["action"]), int(z["episode_len"])
    full = {int(s): i for i, s in enumerate(z["full_steps"])}

    def check_reset(k):
        obs, _ = env.reset()
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["reset_grid"][k]), "reset grid (RNG stream)"
        assert np.array_equal(st.pos[0].cpu().numpy(), z["reset_pos"][k])
        assert np.array_equal(st.stats[0].cpu().numpy(), z["reset_stats"][k])
        assert np.array_equal(obs[0].cpu().numpy(), _expected_obs_3d(shape, z["reset_overlay"][k], z["reset_pos"][k]))

    check_reset(0)
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for t in range(T):
        obs, rew, done, _, info = env.step(acts[t:t + 1])
        st = env.get_state()
        assert zlib.crc32(st.grids[0].cpu().numpy().astype(np.uint8).tobytes()) == int(z["grid_crc"][t]), f"grid @ {t}"
        assert np.array_equal(st.pos[0].cpu().numpy(), z["pos"][t]), f"pos @ {t}"
        got = info["stats"][0].cpu().numpy()
        assert np.array_equal(got, z["stats"][t]), f"stats @ {t}: {got} vs {z['stats'][t]}"
        assert abs(float(rew[0]) - z["reward"][t]) <= REW_TOL, f"reward @ {t}"
        assert bool(done[0]) == bool(z["done"][t]), f"done @ {t}"
        assert int(st.changes[0]) == z["changes"][t] and int(st.iteration[0]) == z["iterations"][t]
        if t in full:
            assert np.array_equal(obs[0].cpu().numpy(), _expected_obs_3d(shape, z["overlay_full"][full[t]], z["pos"][t])), f"obs/overlay @ {t}"
        if t == ep_len - 1:
            check_reset(1)
    env.check_errors()


def test_golden_mc3dmaze_big_stats_known_answers():
    z = np.load(os.path.join(GOLDEN, "stats_mc3dmaze_big.npz"))
    for key in ("15x15x15", "10x10x10"):
        grids = z["grids_" + key]
        env = _vec("minecraft_3D_maze", "narrow", grids.shape[1:], 1, auto_reset=False)
        got = env.stats_for_grids(torch.as_tensor(grids)).cpu().numpy()
        bad = np.nonzero((got != z["stats_" + key]).any(axis=1))[0]
        assert len(bad) == 0, f"{key}: {len(bad)} grids differ, first {bad[:5]}: got {got[bad[:3]]} want {z['stats_' + key][bad[:3]]}"
        env.check_errors()


def test_golden_controllable_3d_episode_replay():
    """controls n_jump / path-length on the 3-D maze (configs/config.py:195-205) against the reference's rewards for
    float targets set through ControlWrapper.set_trgs, and the control observation by the reference's formula."""
    z = np.load(os.path.join(GOLDEN, "control3d_mc3dmaze_narrow_s11.npz"))
    shape = tuple(int(s) for s in z["map_shape"])
    controls = [str(c) for c in z["controls"]]
    env = _vec("minecraft_3D_maze", "narrow", shape, 1, seeds=[int(z["seed"])], auto_reset=False, controls=controls,
               reward_dtype=torch.float64)
    n, t = int(z["steps_per_episode"]), 0
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for ep in range(len(z["reset_at"])):
        env.queue_targets({k: float(v) for k, v in zip(controls, z["reset_trg"][ep])})
        obs, info = env.reset()
        assert np.array_equal(env.get_state().stats[0].cpu().numpy(), z["reset_stats"][ep])
        assert np.allclose(info["ctrl_obs"][0].cpu().numpy(), z["reset_ctrl"][ep], rtol=1e-6, atol=1e-7)
        for _ in range(n):
            obs, rew, done, _, info = env.step(acts[t:t + 1])
            assert np.array_equal(info["stats"][0].cpu().numpy(), z["stats"][t]), f"stats @ {t}"
            assert abs(float(rew[0]) - z["reward"][t]) <= 1e-9, f"reward @ {t}"
            assert np.allclose(info["ctrl_obs"][0].cpu().numpy(), z["ctrl"][t], rtol=1e-6, atol=1e-7), f"ctrl @ {t}"
            t += 1
    env.check_errors()


def test_mc3dmaze_stock_size_batch_vs_oracle():
    """15 x 15 x 15 and 10 x 10 x 10 (planes of more than 64 cells: the multi-word kernels) in batches across
    auto-resets, against the oracle: every step's stats / reward / done, full state and observations now and then."""
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (15, 15, 15), 48, 420, full_every=59, change_percentage=0.03)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (10, 10, 10), 65, 500, full_every=71, change_percentage=0.1)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (16, 16, 16), 9, 150, full_every=37, change_percentage=0.01)


def test_mc3dmaze_odd_observation_windows_vs_oracle():
    """observation windows that are not twice the map: small ones, ones with more rows than the row-table encoder's
    scratch holds (cell-by-cell path), ones wider than 32 cells"""
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (6, 6, 6), 21, 90, full_every=7, obs_window=(20, 18, 6))
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (5, 5, 5), 9, 60, full_every=5, obs_window=(4, 6, 40))
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (7, 7, 7), 30, 80, full_every=9, obs_window=(6, 10, 14))
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (12, 12, 12), 6, 60, full_every=11, obs_window=(36, 34, 8), change_percentage=0.02)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (3, 3, 3), 40, 70, full_every=3)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (2, 4, 4), 12, 50, full_every=3)
    # rows of one or two cells (a 16-byte chunk then spans four / two window rows)
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (4, 4, 2), 84, 60, full_every=3, obs_window=(8, 2, 1))
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (6, 6, 5), 33, 60, full_every=5, obs_window=(8, 12, 1))
    _rollout_vs_oracle("minecraft_3D_maze", "narrow", (5, 5, 3), 33, 60, full_every=5, obs_window=(7, 6, 2))


@pytest.mark.parametrize("shape,n,T,kw", [((7, 7, 7), 1024, 1300, {}), ((15, 15, 15), 64, 500, dict(change_percentage=0.05)),
                                          ((8, 8, 8), 256, 700, {})])
def test_mc3dmaze_biased_actions_vs_oracle(shape, n, T, kw):
    """slowly drifting action bias (long runs of AIR, then of DIRT): big open components, long corridors, deep searches --
    the maps on which the path search's chain trips, 16-entry trips, ring and speculation all get exercised"""
    seeds = 5000 + np.arange(n)
    env = _vec("minecraft_3D_maze", "narrow", shape, n, seeds=seeds, auto_reset=True, **kw)
    orc = po.OracleVecEnv("minecraft_3D_maze", "narrow", shape, n, seeds=seeds, threads=8, **kw)
    obs, _ = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(11)
    for t in range(T):
        a = (torch.rand(n, generator=g) < 0.5 + 0.45 * np.sin(t / 97.0)).to(torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        want = t % 211 == 0 or t == T - 1
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=want)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL and np.array_equal(done.cpu().numpy(), odone)
        if want:
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    env.check_errors()


def test_binary_narrow_4096_envs_vs_oracle():
    """BASELINE configs[1]: binary-narrow 16x16, 4096 envs on one MI355X, bit-exact state check vs CPU,
    across an auto-reset boundary (episode = 770 steps)."""
    n_done = _rollout_vs_oracle("binary", "narrow", (16, 16), 4096, 800)
    assert n_done == 4096


def test_zelda_turtle_4096_envs_vs_oracle():
    """BASELINE configs[2] (shortened rollout, still crosses the auto-reset)."""
    n_done = _rollout_vs_oracle("zelda", "turtle", (16, 16), 4096, 790, full_every=131)
    assert n_done == 4096


def test_sokoban_wide_2048_envs_vs_oracle():
    """BASELINE configs[3] at its batch size (2048 envs per GPU): random wide actions, auto-reset."""
    _rollout_vs_oracle("sokoban", "wide", (16, 16), 2048, 400, seed0=77, full_every=131, threads=16, change_percentage=0.2)


def test_sokoban_wide_2048_envs_as_benchmarked_vs_oracle():
    """BASELINE configs[3] EXACTLY as bench.py runs it: 2048 envs, bench.py's seeds, no change budget, the reference's
    solver_power (10 000, sokoban_prob.py:40), uniform random wide actions, 800 steps across the in-kernel auto-reset
    (episode = 770 steps): statistics, reward and done of every step against the oracle (sokoban_prob.py:160-180); the
    solver runs where the random maps after a reset happen to satisfy its precondition (:172-177)."""
    n_done = _rollout_vs_oracle("sokoban", "wide", (16, 16), 2048, 800, seed0=0x5EED, full_every=131, threads=16)
    assert n_done == 2048


@pytest.mark.parametrize("problem,rep", [("binary", "turtle"), ("binary", "wide"), ("zelda", "narrow"),
                                         ("zelda", "wide"), ("sokoban", "narrow"), ("sokoban", "turtle"),
                                         ("sokoban", "wide")])
def test_other_pairs_vs_oracle(problem, rep):
    _rollout_vs_oracle(problem, rep, (16, 16), 203, 300, full_every=37)


@pytest.mark.parametrize("problem,rep,shape,kw", [
    ("sokoban", "narrow", (32, 32), {}), ("zelda", "narrow", (24, 24), {}), ("binary", "turtle", (64, 64), {}),
    ("zelda", "turtle", (40, 48), {}), ("binary", "narrow", (64, 64), dict(act_window=[3, 3])),
    ("sokoban", "turtle", (24, 40), dict(obs_window=(48, 80)))])
def test_tile_code_observation_configs_vs_oracle(problem, rep, shape, kw):
    """configurations whose one-hot rows would take more than 20 KB of LDS per workgroup: the general kernels compute the
    observation chunks from per-cell tile codes (encode_obs_codes) -- 6 / 9 / 3 channels, 32- and 64-bit row masks, windows
    that hang over every edge of the map, an action patch -- every observation against the oracle"""
    _rollout_vs_oracle(problem, rep, shape, 61, 260, full_every=1, **kw)


@pytest.mark.parametrize("shape", [(8, 8), (5, 7), (12, 16), (16, 32), (32, 32), (20, 24), (64, 32), (40, 16)])
def test_other_map_shapes_vs_oracle(shape):
    """ragged / non-square / maximum sizes: lanes-per-env 8, 16, 32, 64; rows beyond H idle."""
    ow = (2 * shape[0], 32 if shape[1] <= 16 else 64)
    _rollout_vs_oracle("binary", "narrow", shape, 37, 2 * shape[0] * shape[1] * 3 + 40, full_every=53, obs_window=ow)
    _rollout_vs_oracle("zelda", "turtle", shape, 21, 200, full_every=29, obs_window=ow)


@pytest.mark.parametrize("shape,ow", [((1, 16), (2, 32)), ((2, 2), (4, 16)), ((3, 3), (6, 16)), ((1, 1), (2, 16)),
                                      ((64, 1), (16, 16)), ((4, 64), (8, 64)), ((62, 3), (16, 6))])
def test_degenerate_map_shapes_vs_oracle(shape, ow):
    """single-row / single-column / single-cell maps: empty frontiers, one-lane groups, W = 1 and W = 64 masks"""
    _rollout_vs_oracle("binary", "narrow", shape, 19, 3 * shape[0] * shape[1] + 30, full_every=3, obs_window=ow)
    if shape[0] * shape[1] >= 9:  # smaller maps have empty zelda target ranges (the reference fails on them too)
        _rollout_vs_oracle("zelda", "turtle", shape, 19, 60, full_every=3, obs_window=ow)
    if shape[1] > 62 or shape[0] > 62:  # the device solver's level is at most 64 x 64 cells with its border
        with pytest.raises(NotImplementedError):
            _vec("sokoban", "narrow", shape, 4, obs_window=ow)
        return
    _rollout_vs_oracle("sokoban", "narrow", shape, 19, 60, full_every=3, obs_window=ow)


def test_small_obs_window_and_change_percentage():
    _rollout_vs_oracle("binary", "narrow", (16, 16), 64, 200, obs_window=(8, 16), change_percentage=0.2, full_every=7)
    _rollout_vs_oracle("zelda", "turtle", (16, 16), 64, 200, obs_window=(16, 16), full_every=7)


def test_trained_like_maps_stats_vs_oracle():
    """Long corridors (the regime a trained generator reaches): snake / spiral / random walks, path-length > 100."""
    z = np.load(os.path.join(GOLDEN, "stats_binary.npz"))
    rng = np.random.default_rng(5)
    grids = [z["grids"][i] for i in range(13)]
    for _ in range(500):
        g = np.ones((16, 16), np.uint8)
        y, x = rng.integers(16, size=2)
        for _ in range(int(rng.integers(50, 600))):
            g[y, x] = 0
            d = rng.integers(4)
            y = int(np.clip(y + (d == 0) - (d == 1), 0, 15)); x = int(np.clip(x + (d == 2) - (d == 3), 0, 15))
        grids.append(g)
    grids = np.array(grids, np.uint8)
    want = po.stats_for_grids("binary", grids)
    env = _vec("binary", "narrow", (16, 16), 1, auto_reset=False)
    got = env.stats_for_grids(torch.as_tensor(grids)).cpu().numpy()
    assert np.array_equal(got, want)
    assert want[:, 1].max() >= 136


def test_inject_initial_maps_and_masked_reset():
    n = 50
    rng = np.random.default_rng(3)
    grids = rng.integers(0, 8, size=(n, 16, 16), dtype=np.uint8)
    pos = rng.integers(0, 16, size=(n, 2)).astype(np.int32)
    env = _vec("zelda", "turtle", (16, 16), n, seeds=np.arange(n), auto_reset=False)
    orc = po.OracleVecEnv("zelda", "turtle", (16, 16), n, seeds=np.arange(n))
    obs, _ = env.reset(init_grids=grids, init_pos=pos)
    oobs = orc.reset(init_grids=grids, init_pos=pos)
    assert np.array_equal(obs.cpu().numpy(), oobs)
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.stats.cpu().numpy(), ost["stats"])
    assert np.array_equal(st.last_loss.cpu().numpy(), ost["last_loss"])
    # masked reset: only even envs restart (from their RNG streams)
    mask = (np.arange(n) % 2 == 0).astype(np.uint8)
    obs, _ = env.reset(mask=mask)
    oobs = orc.reset(mask=mask)
    assert np.array_equal(obs.cpu().numpy(), oobs)
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])


def test_out_of_range_action_is_reported():
    env = _vec("binary", "narrow", (16, 16), 8, seeds=np.arange(8), auto_reset=False)
    env.reset()
    before = env.get_state().grids.clone()
    a = torch.tensor([0, 1, 2, 0, 1, -1, 0, 1], dtype=torch.int32, device=env.device)
    env.step(a)
    with pytest.raises(ValueError):
        env.check_errors()
    env.check_errors()  # flag is cleared by the poll
    after = env.get_state().grids
    assert torch.equal(before[2], after[2]) and torch.equal(before[5], after[5])  # bad actions edit nothing


def test_gym_adapter_matches_golden():
    """make_env(cfg): the reference's single-env call shape (rl/envs.py:28-81) on top of the engine."""
    from types import SimpleNamespace as NS
    from control_pcgrl_amd import make_env
    z = np.load(os.path.join(GOLDEN, "episode_binary_narrow_s1.npz"))
    cfg = NS(representation="narrow", max_board_scans=3, change_percentage=None, controls=None,
             task=NS(problem="binary", map_shape=(16, 16), obs_window=(32, 32), weights={"path-length": 1, "regions": 1}),
             multiagent=NS(n_agents=0))
    env = make_env(cfg)
    env.unwrapped.seed(int(z["seed"]))
    obs, info = env.reset()
    assert obs.shape == (32, 32, 3) and obs.dtype == np.float32 and info == {}
    assert np.array_equal(obs.astype(np.uint8).ravel(), z["reset_obs"][0])
    for t in range(60):
        obs, r, d, tr, info = env.step(int(z["action"][t]))
        assert r == z["reward"][t] and d == bool(z["done"][t]) and tr == d
        assert env.unwrapped._rep_stats == {"regions": int(z["stats"][t][0]), "path-length": int(z["stats"][t][1])}
        assert info["iterations"] == z["iterations"][t] and info["changes"] == z["changes"][t]
    with pytest.raises(IndexError):
        env.step(2)


CONTROL = sorted(glob.glob(os.path.join(GOLDEN, "control_*.npz")))


@pytest.mark.parametrize("path", CONTROL, ids=[os.path.basename(p)[8:-4] for p in CONTROL])
def test_golden_controllable_episode_replay(path):
    """Controllable generation against the reference's ControlWrapper(ctrl_metrics=...) trace: queued float targets
    applied at reset, float64 reward (tolerance 1e-9: the reference's own sum order is a Python-set order), control
    planes of the observation (float32 here, float64 in the reference: tolerance 1e-6)."""
    z = np.load(path)
    problem, rep = str(z["problem"]), str(z["representation"])
    shape = tuple(int(s) for s in z["map_shape"])
    controls = [str(c) for c in z["controls"]]
    env = _vec(problem, rep, shape, 1, seeds=[int(z["seed"])], auto_reset=False, controls=controls,
               reward_dtype=torch.float64)
    n = int(z["steps_per_episode"])
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    t = 0
    for ep in range(len(z["reset_at"])):
        env.queue_targets({k: float(v) for k, v in zip(controls, z["reset_trg"][ep])})
        obs, info = env.reset()
        assert np.array_equal(env.get_state().stats[0].cpu().numpy(), z["reset_stats"][ep])
        assert zlib.crc32(obs[0].cpu().numpy().tobytes()) == int(z["reset_obs_crc"][ep])
        assert np.allclose(info["ctrl_obs"][0].cpu().numpy(), z["reset_ctrl"][ep], rtol=1e-6, atol=1e-7)
        for _ in range(n):
            obs, rew, done, _, info = env.step(acts[t:t + 1])
            assert np.array_equal(info["stats"][0].cpu().numpy(), z["stats"][t]), f"stats @ {t}"
            assert abs(float(rew[0]) - z["reward"][t]) <= 1e-9, f"reward @ {t}"
            assert zlib.crc32(obs[0].cpu().numpy().tobytes()) == int(z["obs_crc"][t])
            assert np.allclose(info["ctrl_obs"][0].cpu().numpy(), z["ctrl"][t], rtol=1e-6, atol=1e-7), f"ctrl @ {t}"
            t += 1
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,controls", [("binary", "narrow", (16, 16), ["regions", "path-length"]),
                                                        ("zelda", "turtle", (16, 16), ["nearest-enemy", "path-length"]),
                                                        ("minecraft_3D_maze", "narrow", (7, 7, 7), ["n_jump", "path-length"])])
def test_controllable_batch_vs_oracle(problem, rep, shape, controls):
    """random per-env targets re-drawn for every episode (what UniformNoiseyTargets does), auto-reset, vs the oracle"""
    n = 257
    T = 2 * (int(np.prod(shape)) * 3 + 2) + 25
    seeds = 31 + np.arange(n)
    env = _vec(problem, rep, shape, n, seeds=seeds, auto_reset=True, controls=controls, reward_dtype=torch.float64)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, controls=controls, threads=8)
    rng = np.random.default_rng(0)
    bounds = po.cond_bounds(problem, shape)

    def draw():
        return {k: rng.random(n) * (bounds[k][1] - bounds[k][0]) + bounds[k][0] for k in controls}

    trg = draw()
    env.queue_targets({k: torch.as_tensor(v) for k, v in trg.items()})
    orc.queue_targets(trg)
    obs, info = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset())
    assert np.allclose(info["ctrl_obs"].cpu().numpy(), orc.ctrl_obs(), rtol=1e-6, atol=1e-7)
    g = torch.Generator().manual_seed(1)
    for t in range(T):
        if t % 97 == 0:  # keep fresh targets queued: they apply whenever an env resets
            trg = draw()
            env.queue_targets({k: torch.as_tensor(v) for k, v in trg.items()})
            orc.queue_targets(trg)
        a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=(t % 50 == 0))
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy() - orew)) <= 1e-9, f"reward @ {t}"
        assert np.array_equal(done.cpu().numpy(), odone)
        assert np.allclose(info["ctrl_obs"].cpu().numpy(), orc.ctrl_obs(), rtol=1e-6, atol=1e-7), f"ctrl @ {t}"
        if t % 50 == 0:
            assert np.array_equal(obs.cpu().numpy(), oobs)
    env.check_errors()


def test_gym_adapter_controllable_planes():
    from types import SimpleNamespace as NS
    from control_pcgrl_amd import make_env
    z = np.load(os.path.join(GOLDEN, "control_binary_narrow_s7.npz"))
    cfg = NS(representation="narrow", max_board_scans=3, change_percentage=None, controls=["regions", "path-length"],
             task=NS(problem="binary", map_shape=(16, 16), obs_window=(32, 32), weights={"path-length": 1, "regions": 1}),
             multiagent=NS(n_agents=0))
    env = make_env(cfg)
    env.unwrapped.seed(int(z["seed"]))
    env.set_trgs({"regions": float(z["reset_trg"][0][0]), "path-length": float(z["reset_trg"][0][1])})
    obs, _ = env.reset()
    assert obs.shape == (32, 32, 7) and env.observation_space.shape == (32, 32, 7)
    assert np.allclose(obs[5, 9, :4], z["reset_ctrl"][0], rtol=1e-6) and np.all(obs[..., :4] == obs[0, 0, :4])
    for t in range(20):
        obs, r, d, _, _ = env.step(int(z["action"][t]))
        assert abs(r - z["reward"][t]) <= 1e-9
        assert np.allclose(obs[0, 0, :4], z["ctrl"][t], rtol=1e-6)


@pytest.mark.parametrize("shape", [(64, 64), (32, 64), (48, 40), (20, 33), (12, 40), (16, 64), (5, 33)])
def test_maps_wider_than_32_vs_oracle(shape):
    """64-bit row masks (reference task configs binary_bigger / zelda_bigger are 64x64 with a 128x128 window);
    change_percentage keeps episodes short so the RNG reset path is crossed several times."""
    ow = (2 * shape[0], 128 if shape[1] > 32 else 64)
    _rollout_vs_oracle("binary", "narrow", shape, 19, 400, full_every=57, obs_window=ow, change_percentage=0.01)
    _rollout_vs_oracle("zelda", "turtle", shape, 11, 300, full_every=43, obs_window=ow, change_percentage=0.01)
    _rollout_vs_oracle("binary", "turtle", shape, 7, 150, full_every=31, obs_window=ow, change_percentage=0.02)


def test_binary_bigger_wide_vs_oracle():
    _rollout_vs_oracle("binary", "wide", (64, 64), 5, 200, full_every=19, change_percentage=0.01)


@pytest.mark.parametrize("problem,rep,shape", [("binary", "narrow", (16, 16)), ("zelda", "turtle", (16, 16)),
                                               ("sokoban", "wide", (16, 16)), ("minecraft_3D_maze", "narrow", (7, 7, 7))])
def test_update_then_refresh_stats_vs_oracle(problem, rep, shape):
    """evolution-driver pattern (evo/evolve.py:1083-1120): rep.update() K times without PcgrlEnv.step(), then
    get_stats() once; afterwards normal steps (incl. the incremental binary statistics) must still agree."""
    n = 150
    seeds = 5 + np.arange(n)
    env = _vec(problem, rep, shape, n, seeds=seeds, auto_reset=False)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds)
    env.reset(); orc.reset()
    g = torch.Generator().manual_seed(3)
    for t in range(60):
        a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        obs = env.update(a.to(env.device))
        oobs = orc.update(a.numpy())
        if t % 13 == 0:
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"
    st0 = env.get_state()
    assert int(st0.iteration.max()) == 0 and int(st0.changes.max()) == 0  # counters untouched
    assert np.array_equal(env.refresh_stats().cpu().numpy(), orc.refresh_stats())
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    for t in range(40):
        a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a.to(env.device))
        oobs, orew, odone, ostats = orc.step(a.numpy())
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy() - orew)) <= REW_TOL
    assert np.array_equal(obs.cpu().numpy(), oobs)
    env.check_errors()


# ---------------------------------------------------------------- representation wrappers
EXT = sorted(glob.glob(os.path.join(GOLDEN, "ext_*.npz")))


def _ext_kwargs(z):
    kw = {}
    if float(z["static_prob"]) >= 0:
        kw["static_prob"] = float(z["static_prob"])
    if int(z["n_static_walls"]) >= 0:
        kw["n_static_walls"] = int(z["n_static_walls"])
    if int(z["act_window"][0]) > 0:
        kw["act_window"] = [int(a) for a in z["act_window"]]
    if "obs_window" in z.files:  # (fixtures off the 16 x 16 point carry the task config's window)
        kw["obs_window"] = tuple(int(a) for a in z["obs_window"])
    return kw


@pytest.mark.parametrize("path", EXT, ids=[os.path.basename(p)[4:-4] for p in EXT])
def test_golden_rep_wrapper_episode_replay(path):
    """StaticTileRepresentation / MultiActionRepresentation traces of the reference (4 episodes x 160 steps each):
    maps, static masks, positions, stats, rewards, change counters and observations, bit for bit."""
    z = np.load(path)
    problem, rep = str(z["problem"]), str(z["representation"])
    shape = tuple(int(s) for s in z["map_shape"])
    env = _vec(problem, rep, shape, 1, seeds=[int(z["seed"])], auto_reset=False, **_ext_kwargs(z))
    assert tuple(env.obs_shape) == tuple(int(x) for x in z["obs_shape"])
    n = int(z["steps_per_episode"])
    full = {int(t): o for t, o in zip(z["obs_steps"], z["obs_full"])}
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for ep, t0 in enumerate(z["reset_at"]):
        obs, _ = env.reset()
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["reset_grid"][ep]), f"reset {ep}: map"
        assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["reset_pos"][ep])
        assert np.array_equal(st.stats[0].cpu().numpy(), z["reset_stats"][ep])
        if env.static_tiles:
            assert np.array_equal(env.get_static()[0].cpu().numpy().ravel(), z["reset_static"][ep]), f"reset {ep}: static"
        assert np.array_equal(obs[0].cpu().numpy().ravel(), z["reset_obs"][ep]), f"reset {ep}: obs"
        for t in range(int(t0), int(t0) + n):
            obs, rew, done, _, info = env.step(acts[t:t + 1])
            st = env.get_state()
            assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["grid"][t]), f"map @ {t}"
            assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["pos"][t]), f"pos @ {t}"
            assert np.array_equal(info["stats"][0].cpu().numpy(), z["stats"][t]), f"stats @ {t}"
            assert abs(float(rew[0]) - z["reward"][t]) <= REW_TOL, f"reward @ {t}"
            assert bool(done[0]) == bool(z["done"][t])
            assert int(st.changes[0]) == int(z["changes"][t]), f"changes @ {t}"
            o = obs[0].cpu().numpy()
            assert zlib.crc32(o.tobytes()) == int(z["obs_crc"][t]), f"obs @ {t}"
            if t in full:
                assert np.array_equal(o.ravel(), full[t])
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,n_envs,n_steps,kw", [
    ("binary", "narrow", (16, 16), 512, 1000, dict(static_prob=0.3, n_static_walls=3)),
    ("zelda", "turtle", (16, 16), 256, 900, dict(static_prob=0.1, n_static_walls=5)),
    ("sokoban", "narrow", (16, 16), 64, 300, dict(n_static_walls=7)),
    ("binary", "narrow", (16, 16), 512, 900, dict(act_window=[3, 3])),
    ("binary", "narrow", (16, 16), 128, 300, dict(act_window=[16, 1])),
    ("zelda", "narrow", (16, 16), 256, 900, dict(act_window=[2, 2], static_prob=0.1, n_static_walls=3)),
    ("binary", "narrow", (12, 20), 64, 800, dict(static_prob=0.2, n_static_walls=5, obs_window=(24, 32))),
    ("binary", "turtle", (8, 8), 100, 300, dict(static_prob=0.5, n_static_walls=2, obs_window=(16, 16))),
    ("zelda", "narrow", (40, 48), 16, 300, dict(act_window=[5, 3], static_prob=0.2, n_static_walls=6, obs_window=(80, 96))),
    ("binary", "turtle", (64, 64), 8, 300, dict(static_prob=0.3, n_static_walls=7, obs_window=(32, 32))),
])
def test_rep_wrappers_batch_vs_oracle(problem, rep, shape, n_envs, n_steps, kw):
    """static tiles / action patches with auto-reset over whole episodes, all row-mask widths and lane groupings"""
    _rollout_vs_oracle(problem, rep, shape, n_envs, n_steps, full_every=41, **kw)


def test_static_setters_and_eval_mode_vs_oracle():
    """set_static_prob / set_n_static_walls / set_eval_mode (reps/wrappers.py:256-263) take effect at the next reset"""
    n = 64
    kw = dict(static_prob=0.1, n_static_walls=1)
    env = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=False, **kw)
    orc = po.OracleVecEnv("binary", "narrow", (16, 16), n, seeds=np.arange(n), **kw)
    for sp, sw, ev in [(None, None, False), (0.7, 5, True), (0.0, 3, False), (0.4, 0, True)]:
        env.set_static(static_prob=sp, n_static_walls=sw, eval_mode=ev)
        orc.set_static(static_prob=sp, n_static_walls=sw, eval_mode=ev)
        obs, _ = env.reset()
        assert np.array_equal(obs.cpu().numpy(), orc.reset())
        assert np.array_equal(env.get_static().cpu().numpy(), orc.static_tiles())
        a = torch.randint(0, 2, (n,), dtype=torch.int32)
        for _ in range(5):
            obs, rew, done, _, info = env.step(a.to(env.device))
            oobs, orew, odone, ostats = orc.step(a.numpy())
            assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(info["stats"].cpu().numpy(), ostats)


def test_rep_wrappers_update_and_injected_maps_vs_oracle():
    n = 48
    kw = dict(act_window=[4, 4], static_prob=0.3, n_static_walls=4)
    env = _vec("zelda", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=False, **kw)
    orc = po.OracleVecEnv("zelda", "narrow", (16, 16), n, seeds=np.arange(n), **kw)
    assert np.array_equal(env.reset()[0].cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(1)
    for t in range(30):  # evolution pattern: rep.update() only, then get_stats once
        a = torch.randint(0, 8, (n, 16), generator=g, dtype=torch.int32)
        assert np.array_equal(env.update(a.to(env.device)).cpu().numpy(), orc.update(a.numpy())), f"update obs @ {t}"
    assert np.array_equal(env.refresh_stats().cpu().numpy(), orc.refresh_stats())
    grids = np.random.default_rng(2).integers(0, 8, size=(n, 16, 16), dtype=np.uint8)
    assert np.array_equal(env.reset(init_grids=grids)[0].cpu().numpy(), orc.reset(init_grids=grids))
    a = torch.randint(0, 8, (n, 16), generator=g, dtype=torch.int32)
    obs, rew, done, _, info = env.step(a.to(env.device))
    oobs, orew, odone, ostats = orc.step(a.numpy())
    assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(info["stats"].cpu().numpy(), ostats)
    bad = a.clone()
    bad[3, 5] = 8
    before = env.get_state().grids.clone()
    env.step(bad.to(env.device))
    with pytest.raises(ValueError):
        env.check_errors()
    assert torch.equal(before[3], env.get_state().grids[3])  # a patch with a bad entry edits nothing


def test_gym_adapter_action_patch_and_static_tiles():
    from types import SimpleNamespace as NS
    from control_pcgrl_amd import make_env
    z = np.load(os.path.join(GOLDEN, "ext_zelda_narrow_aw2x2_sp10_sw3_s30.npz"))
    cfg = NS(representation="narrow", max_board_scans=3, change_percentage=None, controls=None, act_window=[2, 2],
             static_prob=0.1, n_static_walls=3,
             task=NS(problem="zelda", map_shape=(16, 16), obs_window=(32, 32), weights=None), multiagent=NS(n_agents=0))
    env = make_env(cfg)
    env.unwrapped.seed(int(z["seed"]))
    obs, _ = env.reset()
    assert obs.shape == (32, 32, 10) and tuple(env.action_space.nvec) == (8, 8, 8, 8)
    assert np.array_equal(obs.astype(np.uint8).ravel(), z["reset_obs"][0])
    for t in range(40):
        obs, r, d, tr, info = env.step(z["action"][t])
        assert r == z["reward"][t] and info["changes"] == z["changes"][t]
        assert zlib.crc32(obs.astype(np.uint8).tobytes()) == int(z["obs_crc"][t])
    with pytest.raises(IndexError):
        env.step([0, 1, 8, 0])


@pytest.mark.parametrize("problem,rep,n_envs", [("binary", "narrow", 65536), ("zelda", "turtle", 32768)])
def test_full_size_properties_beyond_oracle_sizes(problem, rep, n_envs):
    """Size-independent properties at batch sizes the CPU oracle does not finish in seconds: the incrementally
    maintained statistics equal a from-scratch evaluation of the same maps (pcgrl_stats_for_grids), refresh_stats is
    idempotent, observe() reproduces the observation returned by step(), rewards telescope to the loss difference."""
    env = _vec(problem, rep, (16, 16), n_envs, seeds=7 + np.arange(n_envs), auto_reset=False)
    env.reset()
    loss0 = env.get_state().last_loss.clone()
    g = torch.Generator(device=env.device).manual_seed(3)
    ret = torch.zeros(n_envs, dtype=torch.float64, device=env.device)
    for t in range(300):
        a = torch.randint(0, env.num_actions, (n_envs,), generator=g, device=env.device, dtype=torch.int32)
        obs, rew, done, _, info = env.step(a)
        ret += rew.double()
    st = env.get_state()
    scratch = env.stats_for_grids(st.grids)
    assert torch.equal(scratch, info["stats"]) and torch.equal(scratch, st.stats)
    assert torch.equal(ret, st.ep_return) and torch.allclose(st.last_loss - loss0, ret, atol=1e-9)
    step_obs = obs.clone()
    assert torch.equal(env.observe(), step_obs)
    stats_before = st.stats.clone()
    assert torch.equal(env.refresh_stats(), stats_before) and torch.equal(env.refresh_stats(), stats_before)
    assert torch.equal(env.get_state().last_loss, st.last_loss)
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,n_envs,K", [("binary", "narrow", (16, 16), 512, 900), ("zelda", "turtle", (16, 16), 256, 850),
                                                        ("sokoban", "wide", (16, 16), 64, 120), ("binary", "turtle", (12, 20), 37, 300),
                                                        ("zelda", "narrow", (40, 48), 9, 100)])
def test_rollout_kernel_equals_stepwise_and_oracle(problem, rep, shape, n_envs, K):
    """pcgrl_rollout (K steps in one launch, state in registers, auto-reset inside) == K x pcgrl_step == oracle"""
    kw = {} if shape == (16, 16) else dict(obs_window=(2 * shape[0], 32 if shape[1] <= 16 else 96))
    seeds = 11 + np.arange(n_envs)
    a = torch.randint(0, _vec(problem, rep, shape, 1, **kw).num_actions, (K, n_envs), dtype=torch.int32,
                      generator=torch.Generator().manual_seed(5))
    env = _vec(problem, rep, shape, n_envs, seeds=seeds, auto_reset=True, **kw)
    env.reset()
    chunks = [(0, 7), (7, K - 40), (K - 40, K)]  # several launches: state carried through memory in between
    obs_all, rew, done, stats = [], [], [], []
    for lo, hi in chunks:
        o, r, d, s = env.rollout(a[lo:hi].to(env.device), want_obs="all")
        obs_all.append(o); rew.append(r); done.append(d); stats.append(s)
    obs_all, rew, done, stats = (torch.cat(x) for x in (obs_all, rew, done, stats))
    orc = po.OracleVecEnv(problem, rep, shape, n_envs, seeds=seeds, threads=8, **kw)
    orc.reset()
    for t in range(K):
        oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=True, want_obs=(t % 37 == 0 or t == K - 1))
        assert np.array_equal(stats[t].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew[t].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL, f"reward @ {t}"
        assert np.array_equal(done[t].cpu().numpy(), odone), f"done @ {t}"
        if oobs is not None:
            assert np.array_equal(obs_all[t].cpu().numpy(), oobs), f"obs @ {t}"
    st, ost = env.get_state(), orc.get_state()
    assert np.array_equal(st.grids.cpu().numpy().reshape(n_envs, -1), ost["grids"])
    assert np.array_equal(st.iteration.cpu().numpy(), ost["iteration"]) and np.array_equal(st.changes.cpu().numpy(), ost["changes"])
    le, ole = env.last_episode(), orc.last_episode()
    assert np.array_equal(le.n_episodes.cpu().numpy(), ole["n_episodes"]) and np.array_equal(le.final_stats.cpu().numpy(), ole["final_stats"])
    # and the step-wise path continues from the rollout's state (RNG streams included)
    for t in range(3):
        act = torch.randint(0, env.num_actions, (n_envs,), dtype=torch.int32, generator=torch.Generator().manual_seed(t))
        obs, r, d, _, info = env.step(act.to(env.device))
        oobs, orew, odone, ostats = orc.step(act.numpy(), auto_reset=True)
        assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(info["stats"].cpu().numpy(), ostats)
    o_last, _, _, _ = env.rollout(a[:5].to(env.device), want_obs="last")
    for t in range(5):
        oobs, _, _, _ = orc.step(a[t].numpy(), auto_reset=True)
    assert np.array_equal(o_last.cpu().numpy(), oobs)
    env.check_errors()


@pytest.mark.parametrize("force", ["1", "0"])
def test_rollout_both_forms_on_a_large_and_a_small_map(force):
    """pcgrl_rollout picks its form by configuration (step launches on 2-D maps of more than 16 rows, one launch elsewhere);
    PCGRL_ROLLOUT_KERNEL forces either form, and every combination equals the oracle"""
    os.environ["PCGRL_ROLLOUT_KERNEL"] = force
    try:
        for problem, rep, shape, n, K, kw in (("binary", "narrow", (40, 48), 11, 90, dict(obs_window=(80, 96))),
                                             ("zelda", "turtle", (16, 16), 70, 120, {}),
                                             ("minecraft_3D_maze", "narrow", (10, 10, 10), 5, 60, {}),
                                             ("binary", "narrow", (16, 16), 40, 100, dict(static_prob=0.2, n_static_walls=2))):
            seeds = 21 + np.arange(n)
            env = _vec(problem, rep, shape, n, seeds=seeds, auto_reset=True, change_percentage=0.1, **kw)
            orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8, change_percentage=0.1, **kw)
            env.reset()
            orc.reset()
            a = torch.randint(0, env.num_actions, (K, n), dtype=torch.int32, generator=torch.Generator().manual_seed(6))
            obs, rew, done, stats = env.rollout(a.to(env.device), want_obs="all")
            for t in range(K):
                oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=True)
                assert np.array_equal(stats[t].cpu().numpy(), ostats), f"{problem} {shape} stats @ {t}"
                assert np.max(np.abs(rew[t].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL and np.array_equal(done[t].cpu().numpy(), odone)
                assert np.array_equal(obs[t].cpu().numpy(), oobs), f"{problem} {shape} obs @ {t}"
            o_last, _, _, _ = env.rollout(a[:7].to(env.device), want_obs="last")
            for t in range(7):
                oobs, _, _, _ = orc.step(a[t].numpy(), auto_reset=True)
            assert np.array_equal(o_last.cpu().numpy(), oobs)
            env.check_errors()
    finally:
        del os.environ["PCGRL_ROLLOUT_KERNEL"]


def test_rollout_kernel_3d_equals_oracle():
    n, K, shape = 96, 1100, (7, 7, 7)  # episode length 1031: auto-resets inside the launch
    seeds = 3 + np.arange(n)
    a = torch.randint(0, 2, (K, n), dtype=torch.int32, generator=torch.Generator().manual_seed(9))
    env = _vec("minecraft_3D_maze", "narrow", shape, n, seeds=seeds, auto_reset=True)
    env.reset()
    parts = [env.rollout(a[lo:hi].to(env.device), want_obs="all") for lo, hi in [(0, 3), (3, K)]]
    obs_all, rew, done, stats = (torch.cat([p[i] for p in parts]) for i in range(4))
    orc = po.OracleVecEnv("minecraft_3D_maze", "narrow", shape, n, seeds=seeds, threads=8)
    orc.reset()
    for t in range(K):
        oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=True, want_obs=(t % 29 == 0 or t == K - 1 or 1028 <= t <= 1034))
        assert np.array_equal(stats[t].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew[t].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL and np.array_equal(done[t].cpu().numpy(), odone)
        if oobs is not None:
            assert np.array_equal(obs_all[t].cpu().numpy(), oobs), f"obs @ {t}"
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    act = torch.ones(n, dtype=torch.int32)
    obs, r, d, _, info = env.step(act.to(env.device))
    oobs, orew, odone, ostats = orc.step(act.numpy(), auto_reset=True)
    assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(info["stats"].cpu().numpy(), ostats)
    env.check_errors()


def test_rollout_kernel_without_auto_reset_and_with_change_budget():
    n, K = 128, 90
    kw = dict(change_percentage=0.1)
    a = torch.randint(0, 12, (K, n), dtype=torch.int32, generator=torch.Generator().manual_seed(2))
    env = _vec("zelda", "turtle", (16, 16), n, seeds=np.arange(n), auto_reset=False, **kw)
    orc = po.OracleVecEnv("zelda", "turtle", (16, 16), n, seeds=np.arange(n), **kw)
    env.reset(); orc.reset()
    obs, rew, done, stats = env.rollout(a.to(env.device), want_obs="last")
    for t in range(K):
        oobs, orew, odone, ostats = orc.step(a[t].numpy(), auto_reset=False)
        assert np.array_equal(stats[t].cpu().numpy(), ostats) and np.array_equal(done[t].cpu().numpy(), odone), f"@ {t}"
        assert np.max(np.abs(rew[t].cpu().numpy().astype(np.float64) - orew)) <= REW_TOL
    assert done[-1].any() and not done[0].any()  # the change budget ended some episodes; without auto-reset they stay done
    assert np.array_equal(obs.cpu().numpy(), oobs)
    assert np.array_equal(env.get_state().changes.cpu().numpy(), orc.get_state()["changes"])


def test_golden_reference_test3d_maps():
    """the reference's own 3-D known-answer maps (test3D.py) through pcgrl_stats_for_grids, one shape at a time"""
    z = np.load(os.path.join(GOLDEN, "stats_mc3dmaze_test3d.npz"))
    for i, sh in enumerate(z["shapes"]):
        sh = tuple(int(s) for s in sh)
        g = z["grids"][i, : int(np.prod(sh))].reshape((1,) + sh)
        env = _vec("minecraft_3D_maze", "narrow", sh, 1, auto_reset=False, obs_window=(2, 2, 2))
        got = env.stats_for_grids(torch.as_tensor(g)).cpu().numpy()[0]
        assert np.array_equal(got, z["stats"][i]), f"{z['names'][i]} {sh}: got {got}, reference {z['stats'][i]}"
        env.close()


@pytest.mark.parametrize("rep", ["narrow", "turtle"])
def test_steps_without_observation_output_and_mixed_call_sequences(rep):
    """pcgrl_step with d_obs = NULL (no observe wave), mixed with normal steps, updates, refreshes, rollouts and masked
    resets: the hand-over plane of the pre-flooded component must never be used stale"""
    n = 256
    env = _vec("binary", rep, (16, 16), n, seeds=5 + np.arange(n), auto_reset=True, change_percentage=0.4)
    orc = po.OracleVecEnv("binary", rep, (16, 16), n, seeds=5 + np.arange(n), threads=8, change_percentage=0.4)
    assert np.array_equal(env.reset()[0].cpu().numpy(), orc.reset())
    g = torch.Generator().manual_seed(4)
    sp = torch.cuda.current_stream().cuda_stream
    for t in range(400):
        a = torch.randint(0, env.num_actions, (n,), generator=g, dtype=torch.int32)
        ad = a.to(env.device)
        kind = t % 7
        if kind in (0, 1, 2):  # normal step
            obs, rew, done, _, info = env.step(ad)
            oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True)
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"
        elif kind in (3, 4):  # no observation output
            _lib_check = env._L.pcgrl_step(env._h, ad.data_ptr(), 1, None, env._ptrs[1], env._ptrs[2], env._ptrs[3], sp)
            assert _lib_check == 0
            info = {"stats": env._stats}
            oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=False)
        elif kind == 5:  # rep.update only, then a stats refresh
            env.update(ad)
            orc.update(a.numpy())
            assert np.array_equal(env.refresh_stats().cpu().numpy(), orc.refresh_stats())
            continue
        else:  # a short rollout, or a masked reset
            if t % 14 == 6:
                acts = torch.randint(0, env.num_actions, (3, n), generator=g, dtype=torch.int32)
                _, _, _, stats = env.rollout(acts.to(env.device), want_obs="none")
                for k in range(3):
                    _, _, _, ostats = orc.step(acts[k].numpy(), auto_reset=True, want_obs=False)
                info = {"stats": stats[-1]}
            else:
                mask = (torch.arange(n) % 3 == 0).to(torch.uint8)
                env.reset(mask=mask)
                orc.reset(mask=mask.numpy())
                continue
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t} (kind {kind})"
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,kw", [
    ("binary", "narrow", (16, 16), dict(controls=["regions", "path-length"])),
    ("zelda", "turtle", (16, 16), dict(controls=["nearest-enemy", "path-length"])),
    ("sokoban", "wide", (16, 16), dict(controls=["crate"])),
    ("binary", "narrow", (12, 20), dict(controls=["path-length"])),
    ("binary", "narrow", (16, 16), dict(static_prob=0.3, n_static_walls=3)),
    ("zelda", "turtle", (16, 16), dict(static_prob=0.2, n_static_walls=2)),
    ("binary", "narrow", (16, 16), dict(act_window=(3, 3))),
    ("zelda", "narrow", (16, 16), dict(act_window=(2, 2), static_prob=0.1, n_static_walls=3)),
    ("minecraft_3D_maze", "narrow", (5, 5, 5), dict(controls=["path-length"])),
])
def test_rollout_kernel_controllable_and_rep_wrappers_equal_stepwise(problem, rep, shape, kw):
    """pcgrl_rollout_ex in the controllable mode (queued per-env targets taken at the resets inside the launch, float64
    rewards, control observation) and with the representation wrappers (static tiles, action patches) == the same steps
    through pcgrl_step_ex, including auto-resets; the step-wise path itself is pinned against the reference fixtures."""
    n, K = 96, 130
    ctrl = "controls" in kw
    mk = lambda: _vec(problem, rep, shape, n, seeds=300 + np.arange(n), auto_reset=True, change_percentage=0.08,
                      reward_dtype=torch.float64 if ctrl else torch.float32, **kw)
    a, b = mk(), mk()
    if ctrl:
        for env in (a, b):
            env.sample_uniform_targets(generator=torch.Generator(device=env.device).manual_seed(4))
    a.reset(); b.reset()
    g = torch.Generator().manual_seed(12)
    shape_a = (K, n, a.action_entries) if a.action_entries > 1 else (K, n)
    hi = a.spec.n_tiles if a.action_entries > 1 else a.num_actions
    acts = torch.randint(0, hi, shape_a, generator=g, dtype=torch.int32).to(a.device)
    if ctrl:  # new targets are queued before the run: they must apply at each env's next reset in both paths
        for env in (a, b):
            env.sample_uniform_targets(generator=torch.Generator(device=env.device).manual_seed(5))
    obs_r, rew_r, done_r, stats_r = a.rollout(acts, want_obs="all")
    n_done = 0
    for k in range(K):
        obs, rew, done, _, info = b.step(acts[k])
        assert torch.equal(info["stats"], stats_r[k]), f"stats @ {k}"
        assert torch.equal(done, done_r[k]), f"done @ {k}"
        assert float((rew.double() - rew_r[k].double()).abs().max()) <= (1e-9 if ctrl else 0.0), f"reward @ {k}"
        assert torch.equal(obs, obs_r[k]), f"obs @ {k}"
        n_done += int(done.sum())
    assert n_done > n // 2
    if ctrl:
        assert torch.equal(a.ctrl_obs, info["ctrl_obs"])
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa.grids, sb.grids) and torch.equal(sa.counters, sb.counters) and torch.equal(sa.stats, sb.stats)
    assert torch.equal(a.get_rng_state(), b.get_rng_state())
    if a.static_tiles:
        assert torch.equal(a.get_static(), b.get_static())
    # ... and continuing step-wise after the rollout stays identical
    more = torch.randint(0, hi, (20,) + tuple(shape_a[1:]), generator=g, dtype=torch.int32).to(a.device)
    for k in range(20):
        oa, ra, da, _, ia = a.step(more[k])
        ob, rb, db, _, ib = b.step(more[k])
        assert torch.equal(oa, ob) and torch.equal(ia["stats"], ib["stats"]) and torch.equal(da, db)
    a.check_errors(); b.check_errors()


def test_rollout_rejects_malformed_actions():
    with pytest.raises(ValueError):
        _vec("binary", "narrow", (16, 16), 4).rollout(torch.zeros((2, 5), dtype=torch.int32))


# ---------------------------------------------------------------- round 2: races, stale stats, reductions, checkpoints
@pytest.mark.parametrize("problem,rep", [("binary", "narrow"), ("zelda", "turtle")])
def test_auto_reset_every_few_steps_4096_envs_vs_oracle(problem, rep):
    """Auto-reset stress: max_changes = 1, so an env resets at its second change -- some envs of every launch reset,
    and both waves of their workgroup replay the env's RNG streams (the observe wave from the copy it took before the
    barrier).  Thousands of launches at the BASELINE batch size, every output against the oracle."""
    n = 4096
    _rollout_vs_oracle(problem, rep, (16, 16), n, 1500, seed0=31, full_every=211, threads=16, change_percentage=0.001)


@pytest.mark.parametrize("problem,rep,shape,kw", [
    ("binary", "narrow", (16, 16), {}), ("binary", "turtle", (20, 24), {}), ("zelda", "narrow", (16, 16), {}),
    ("sokoban", "turtle", (16, 16), {}), ("binary", "narrow", (40, 48), {}), ("minecraft_3D_maze", "narrow", (7, 7, 7), {}),
    ("minecraft_3D_maze", "narrow", (10, 10, 10), {}),
    # static tiles: a build on a static tile is undone but still reported as a change -- the reference then recomputes
    # the statistics of the (updated) map; found by tests/fuzz_parity.py
    ("binary", "turtle", (15, 30), dict(static_prob=0.3, n_static_walls=2)),
    ("zelda", "narrow", (12, 12), dict(static_prob=0.7, n_static_walls=2)),
    ("binary", "narrow", (8, 9), dict(static_prob=0.5, n_static_walls=0, act_window=[1, 3]))])
def test_update_then_step_without_refresh_vs_oracle(problem, rep, shape, kw):
    """pcgrl_update leaves the statistics stale; the next CHANGING pcgrl_step must recompute them from scratch (the
    reference's get_stats, pcgrl_env.py:314-323) -- not incrementally from the stale masks -- and non-changing steps in
    between keep reporting the old values (the reference's _rep_stats)."""
    n = 192
    seeds = 900 + np.arange(n)
    env = _vec(problem, rep, shape, n, seeds=seeds, auto_reset=False, **kw)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, **kw)
    env.reset(); orc.reset()
    g = torch.Generator().manual_seed(5)
    size = (n, env.action_entries) if env.action_entries > 1 else (n,)
    for rnd in range(4):
        for t in range(3 if kw else 25):
            a = torch.randint(0, env.num_actions, size, generator=g, dtype=torch.int32)
            env.update(a.to(env.device), want_obs=(t % 2 == 0))
            orc.update(a.numpy())
        for t in range(30):
            a = torch.randint(0, env.num_actions, size, generator=g, dtype=torch.int32)
            if rnd == 3:  # through the rollout kernel
                _, rew, done, stats = env.rollout(a.to(env.device)[None], want_obs="none")
                rew, stats = rew[0], stats[0]
            else:
                _, rew, done, _, info = env.step(a.to(env.device))
                stats = info["stats"]
            _, orew, odone, ostats = orc.step(a.numpy())
            assert np.array_equal(stats.cpu().numpy(), ostats), f"stats @ round {rnd} step {t}"
            assert np.max(np.abs(rew.cpu().numpy() - orew)) <= REW_TOL, f"reward @ round {rnd} step {t}"
    assert np.array_equal(env.get_state().grids.cpu().numpy().reshape(n, -1), orc.get_state()["grids"])
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,n", [("binary", "narrow", (16, 16), 4096), ("zelda", "turtle", (16, 16), 1000),
                                                 ("minecraft_3D_maze", "narrow", (3, 3, 3), 700)])
def test_reduce_episodes_equals_per_env_totals(problem, rep, shape, n):
    """pcgrl_reduce_episodes = sum over envs and episodes of what pcgrl_get_last_episode latches, bit-reproducible;
    clear restarts the totals."""
    env = _vec(problem, rep, shape, n, seeds=np.arange(n), auto_reset=True, change_percentage=0.02)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(2)
    tot = np.zeros(3 + env.n_stats)
    seen = np.zeros(n, np.int64)
    for t in range(400):
        a = torch.randint(0, env.num_actions, (n,), generator=g, device=env.device, dtype=torch.int32)
        _, _, done, _, _ = env.step(a)
        if done.any():
            le = env.last_episode()
            d = done.cpu().numpy()
            tot[0] += le.ep_return.cpu().numpy()[d].sum()
            tot[1] += le.ep_len.cpu().numpy()[d].sum()
            tot[2] += d.sum()
            tot[3:] += le.final_stats.cpu().numpy()[d].sum(0)
            seen += d
        if t == 250:
            a1 = env.reduce_episodes(clear=False).cpu().numpy()
            a2 = env.reduce_episodes(clear=False).cpu().numpy()
            assert np.array_equal(a1, a2)
            assert np.allclose(a1, tot, rtol=0, atol=1e-6), (a1, tot)
            part = tot.copy()
            assert np.array_equal(env.reduce_episodes(clear=True).cpu().numpy(), a1)
            assert not env.reduce_episodes(clear=False).cpu().numpy().any()
    assert tot[2] > n  # several episodes per env
    assert np.array_equal(env.last_episode().n_episodes.cpu().numpy(), seen)
    assert np.allclose(env.reduce_episodes().cpu().numpy(), tot - part, rtol=0, atol=1e-6)
    env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,xkw", [
    ("binary", "narrow", (16, 16), {}), ("zelda", "turtle", (16, 16), {}), ("sokoban", "wide", (16, 16), {}),
    ("minecraft_3D_maze", "narrow", (7, 7, 7), {}), ("binary", "turtle", (40, 48), {}),
    ("minecraft_3D_maze", "narrow", (10, 10, 10), {}),
    ("binary", "narrow", (16, 16), dict(static_prob=0.3, n_static_walls=3)),
    ("zelda", "narrow", (16, 16), dict(static_prob=0.1, n_static_walls=3, act_window=[2, 2])),
    ("binary", "narrow", (16, 16), dict(controls=["regions", "path-length"])),
    ("minecraft_3D_maze", "narrow", (7, 7, 7), dict(controls=["n_jump", "path-length"]))])
def test_state_dict_round_trip_continues_bit_exactly(problem, rep, shape, xkw):
    """checkpoint / restore (pcgrl_export_state -> pcgrl_import_state): a run restored into ANOTHER engine continues with
    identical outputs, including auto-resets drawn from the restored RNG -- with static tiles / action patches (static
    mask, lagging bordered planes, spare RNG half) and in controllable mode (active and queued targets) too."""
    n = 96
    kw = dict(seeds=40 + np.arange(n), auto_reset=True, change_percentage=0.05, **xkw)
    if "controls" in xkw:
        kw["reward_dtype"] = torch.float64
    env = _vec(problem, rep, shape, n, **kw)
    if "controls" in xkw:
        env.sample_uniform_targets(generator=torch.Generator(device=env.device).manual_seed(1))
    env.reset()
    if "controls" in xkw:  # targets queued but not yet applied when the checkpoint is taken
        env.sample_uniform_targets(generator=torch.Generator(device=env.device).manual_seed(2))
    g = torch.Generator().manual_seed(8)
    ashape = (140, n, env.action_entries) if env.action_entries > 1 else (140, n)
    acts = torch.randint(0, env.spec.n_tiles if env.act_window else env.num_actions, ashape, generator=g, dtype=torch.int32).to(env.device)
    for t in range(60):
        env.step(acts[t])
    sd = env.state_dict()
    want = []
    for t in range(60, 140):
        obs, rew, done, _, info = env.step(acts[t])
        want.append((obs.clone(), rew.clone(), done.clone(), info["stats"].clone()))
    kw["seeds"] = np.zeros(n, np.int64)
    other = _vec(problem, rep, shape, n, **kw)
    other.reset()
    other.load_state_dict(sd)
    st = other.get_state()
    assert torch.equal(st.grids, sd["grids"]) and torch.equal(st.counters[:, :3], sd["counters"][:, :3])
    for t in range(60, 140):
        obs, rew, done, _, info = other.step(acts[t])
        w = want[t - 60]
        assert torch.equal(info["stats"], w[3]), f"stats @ {t}"
        assert torch.equal(rew, w[1]) and torch.equal(done, w[2]), f"reward / done @ {t}"
        assert torch.equal(obs, w[0]), f"obs @ {t}"
    other.check_errors()


def test_state_image_from_another_config_is_refused():
    """pcgrl_import_state checks the image's header: an image of the same byte size from an engine with other weights, another
    change budget or another problem of the same layout (none of which change pcgrl_state_bytes) is refused before anything
    is overwritten; a corrupted header is refused as well"""
    n = 32
    a = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True)
    a.reset()
    a.step(torch.zeros(n, dtype=torch.int32, device=a.device))
    sd = a.state_dict()
    before = a.get_state().grids.clone()
    for kw in (dict(weights={"regions": 2.0, "path-length": 1.0}), dict(change_percentage=0.5), dict(max_board_scans=2)):
        b = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, **kw)
        b.reset()
        keep = b.get_state().grids.clone()
        assert b._L.pcgrl_state_bytes(b._h) == a._L.pcgrl_state_bytes(a._h)
        with pytest.raises(ValueError, match="another config"):
            b.load_state_dict(sd)
        assert torch.equal(b.get_state().grids, keep), "a refused image must not touch the engine"
    bad = dict(sd, blob=sd["blob"].clone())
    bad["blob"][0] ^= 0xFF
    with pytest.raises(ValueError, match="magic"):
        a.load_state_dict(bad)
    a.load_state_dict(sd)  # the right image still imports
    assert torch.equal(a.get_state().grids, before)
    a.check_errors()


def test_stats_for_grids_any_batch_size_and_async():
    """Problem.get_stats through an existing engine's scratch: the number of maps is independent of the engine's batch."""
    z = np.load(os.path.join(GOLDEN, "stats_sokoban.npz"))
    env = _vec("sokoban", "narrow", z["grids"].shape[1:], 3, auto_reset=False)
    got = env.stats_for_grids(torch.as_tensor(z["grids"]))
    assert np.array_equal(got.cpu().numpy(), z["stats"])
    z = np.load(os.path.join(GOLDEN, "stats_mc3dmaze.npz"))
    env3 = _vec("minecraft_3D_maze", "narrow", z["grids"].shape[1:], 2, auto_reset=False)
    assert np.array_equal(env3.stats_for_grids(torch.as_tensor(z["grids"])).cpu().numpy(), z["stats"])
    env.check_errors(); env3.check_errors()


def test_bench_two_ranks_on_one_device():
    """bench.py --gpus 2 started plainly: it launches its own two ranks (here both on cuda:0, gloo collectives), shards
    the envs, and the reduced episode count is the sum of the shards'."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PCGRL_BENCH_SINGLE_DEVICE="1", PCGRL_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "800",
                        "--warmup", "0", "--no-cpu-baseline", "--rollout-steps", "0"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["envs_per_gpu"] == 256 and out["config"]["global_envs"] == 512
    assert out["config"]["seed_ranges"] == [[0x5EED, 0x5EED + 255], [0x5EED + 256, 0x5EED + 511]]
    # 800 steps of a 770-step episode: every env of both shards finished exactly one episode
    assert out["per_rank"]["episodes"] == [256.0, 256.0] and out["episodes"]["episodes"] == 512.0
    assert out["episodes"]["mean_length"] == 770.0
    assert len(out["per_rank"]["env_steps_per_s"]) == 2
    rf = out["roofline"]
    assert abs(rf["frac"] - rf["algorithmic_bytes_per_launch"] / (out["ms_per_step"] * 1e-3) / 1e9 / rf["peak"]) < 1e-9


# ---------------------------------------------------------------- map shapes off the 16x16 point, reference-pinned
SHAPES = sorted(glob.glob(os.path.join(GOLDEN, "shape_*.npz")))


@pytest.mark.parametrize("path", SHAPES, ids=[os.path.basename(p)[6:-4] for p in SHAPES])
def test_golden_shape_episode_replay(path):
    """Reference episodes at the reference's own binary_big (32x32) / binary_bigger, zelda_bigger (64x64, obs 128x128) /
    zelda_big / zelda_small (7x11, obs 22x22: rows of 198 bytes) task configs and at shapes that select every other
    kernel family (8 / 16 / 32 / 64 lanes per env, 32- and 64-bit row masks, non-square maps, sokoban off 16x16)."""
    z = np.load(path)
    problem, rep = str(z["problem"]), str(z["representation"])
    shape = tuple(int(s) for s in z["map_shape"])
    env = _vec(problem, rep, shape, 1, seeds=[int(z["seed"])], auto_reset=False,
               obs_window=tuple(int(s) for s in z["obs_window"]), change_percentage=float(z["change_percentage"]))
    assert env.cfg.max_changes == int(z["max_changes"])
    T, ep_len = len(z["action"]), int(z["episode_len"])
    full = {int(s): i for i, s in enumerate(z["full_steps"])}

    def check_reset(k):
        obs, _ = env.reset()
        st = env.get_state()
        assert np.array_equal(st.grids[0].cpu().numpy().ravel(), z["reset_grid"][k]), "reset grid (RNG stream)"
        if rep != "wide":
            assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["reset_pos"][k])
        assert np.array_equal(st.stats[0].cpu().numpy(), z["reset_stats"][k])
        assert zlib.crc32(obs[0].cpu().numpy().tobytes()) == int(z["reset_obs_crc"][k])

    check_reset(0)
    acts = torch.as_tensor(z["action"], dtype=torch.int32, device=env.device)
    for t in range(T):
        obs, rew, done, _, info = env.step(acts[t:t + 1])
        st = env.get_state()
        g = st.grids[0].cpu().numpy()
        assert zlib.crc32(g.tobytes()) == int(z["grid_crc"][t]), f"grid @ {t}"
        if rep != "wide":
            assert np.array_equal(st.pos[0, :2].cpu().numpy(), z["pos"][t]), f"pos @ {t}"
        got = info["stats"][0].cpu().numpy()
        assert np.array_equal(got, z["stats"][t]), f"stats @ {t}: {got} vs {z['stats'][t]}"
        assert abs(float(rew[0]) - z["reward"][t]) <= REW_TOL, f"reward @ {t}"
        assert bool(done[0]) == bool(z["done"][t]), f"done @ {t}"
        assert int(st.changes[0]) == z["changes"][t] and int(st.iteration[0]) == z["iterations"][t]
        o = obs[0].cpu().numpy()
        assert zlib.crc32(o.tobytes()) == int(z["obs_crc"][t]), f"obs crc @ {t}"
        if t in full:
            assert np.array_equal(g.ravel(), z["grid_full"][full[t]])
            assert np.array_equal(o.ravel(), z["obs_full"][full[t]])
        if t == ep_len - 1:
            check_reset(1)
    env.check_errors()


def test_golden_shape_stats_known_answers():
    z = np.load(os.path.join(GOLDEN, "stats_shapes.npz"))
    for key in sorted(k[6:] for k in z.files if k.startswith("grids_")):
        problem = key.split("_")[0]
        grids = z["grids_" + key]
        env = _vec(problem, "narrow", grids.shape[1:], 1, auto_reset=False)
        got = env.stats_for_grids(torch.as_tensor(grids)).cpu().numpy()
        assert np.array_equal(got, z["stats_" + key]), key
        env.check_errors()


@pytest.mark.parametrize("problem,rep,shape,ow", [("zelda", "turtle", (7, 11), (22, 22)), ("zelda", "narrow", (7, 11), (22, 22)),
                                                  ("binary", "narrow", (9, 9), (11, 13)), ("sokoban", "wide", (10, 10), None),
                                                  ("binary", "turtle", (33, 35), (7, 5)), ("zelda", "wide", (6, 6), None)])
def test_observation_rows_of_odd_size_vs_oracle(problem, rep, shape, ow):
    """observation rows that are not a multiple of 16 bytes (byte-string store path), batch of envs"""
    kw = {} if ow is None else {"obs_window": ow}
    _rollout_vs_oracle(problem, rep, shape, 77, 200, seed0=700, full_every=9, change_percentage=0.3, **kw)


def test_static_tiles_with_odd_observation_rows_vs_oracle():
    _rollout_vs_oracle("zelda", "narrow", (7, 11), 50, 220, seed0=720, full_every=7, obs_window=(22, 22),
                       static_prob=0.2, n_static_walls=2)


@pytest.mark.parametrize("name", ["binary_narrow", "zelda_turtle", "sokoban_wide"])
def test_rllib_vector_env_adapter_matches_golden(name):
    """PcgrlVectorEnv (ray.rllib VectorEnv call shape, duck-typed: no ray here): three reference episodes side by side
    in ONE engine -- vector_reset / vector_step / reset_at / get_sub_environments against the fixtures."""
    from control_pcgrl_amd import PcgrlVectorEnv
    zs = [np.load(os.path.join(GOLDEN, f"episode_{name}_s{s}.npz")) for s in (1, 2, 3)]
    z0 = zs[0]
    cfg = {"task": {"problem": str(z0["problem"]), "map_shape": [int(s) for s in z0["map_shape"]],
                    "obs_window": [int(s) for s in z0["obs_window"]], "weights": None},
           "representation": str(z0["representation"])}
    env = PcgrlVectorEnv(cfg, num_envs=3, seeds=[int(z["seed"]) for z in zs], obs_dtype=np.float32)
    assert env.observation_space.shape == tuple(int(s) for s in z0["obs_shape"]) and env.action_space.n == int(z0["n_actions"])
    obs, infos = env.vector_reset()
    assert len(obs) == 3 and infos == [{}, {}, {}]
    for k, z in enumerate(zs):
        assert obs[k].dtype == np.float32 and np.array_equal(obs[k].astype(np.uint8).ravel(), z["reset_obs"][0])
    ep_len = int(z0["episode_len"])
    T = min(len(z["action"]) for z in zs)
    for t in range(T):
        obs, rew, term, trunc, infos = env.vector_step([int(z["action"][t]) for z in zs])
        assert term == trunc and len(infos) == 3
        for k, z in enumerate(zs):
            assert zlib.crc32(obs[k].astype(np.uint8).tobytes()) == int(z["obs_crc"][t]), f"obs env {k} @ {t}"
            assert abs(rew[k] - z["reward"][t]) <= REW_TOL and term[k] == bool(z["done"][t])
        if t % 97 == 0 or t == ep_len - 1:
            for k, z in enumerate(zs):
                want = dict(zip([str(s) for s in z["stat_keys"]], z["stats"][t].tolist()))
                info = infos[k]
                assert {s: info[s] for s in want} == want and info["iterations"] == int(z["iterations"][t])
                assert env.get_sub_environments()[k].unwrapped._rep_stats == want
        if t == ep_len - 1:
            assert all(term)
            for k in (2, 0, 1):  # RLlib resets each finished env; one masked launch serves all three
                o, info = env.reset_at(k)
                assert info == {} and np.array_equal(o.astype(np.uint8).ravel(), zs[k]["reset_obs"][1])
    with pytest.raises(IndexError):
        env.vector_step([0, 10 ** 6, 0])
    env.close()


@pytest.mark.parametrize("obs_dtype,direct", [(np.float32, None), (np.uint8, False), (np.uint8, True)],
                         ids=["float32", "uint8-copy", "uint8-kernel-writes-host"])
def test_rllib_vector_env_adapter_never_rewrites_what_it_returned(obs_dtype, direct):
    """RLlib's collectors keep references to the observations / infos a call returned and stack them later: neither the
    next vector_step nor a reset_at (which RLlib calls inside its per-env loop, before it has consumed the other envs'
    last observations) may change them -- in every hand-out mode (float32 converted on the device, the engine's uint8
    copied into a pinned block, the kernel writing into the pinned block itself)."""
    from control_pcgrl_amd import PcgrlVectorEnv
    cfg = {"task": {"problem": "binary", "map_shape": [16, 16], "obs_window": [32, 32], "weights": None},
           "representation": "narrow", "change_percentage": 0.02}
    n = 5
    env = PcgrlVectorEnv(cfg, num_envs=n, seeds=list(range(n)), obs_dtype=obs_dtype, direct_host_outputs=direct)
    assert env._direct == (bool(direct) or env._host_convert) and env._host_convert == (obs_dtype is np.float32)
    obs0, _ = env.vector_reset()
    keep0 = [o.copy() for o in obs0]
    rng = np.random.default_rng(0)
    kept = []
    for t in range(60):
        obs, rew, term, trunc, infos = env.vector_step(rng.integers(0, 2, n).tolist())
        kept.append((obs, [o.copy() for o in obs], infos, [dict(infos[i]) for i in range(n)]))
        if any(term):
            for i in np.nonzero(term)[0]:
                o, _ = env.reset_at(int(i))
                # the reset of env i left every array of the last step as it was -- also those of the other finished envs
                assert all(np.array_equal(a, b) for a, b in zip(kept[-1][0], kept[-1][1]))
                assert not np.shares_memory(o, kept[-1][0][int(i)])
    assert all(np.array_equal(a, b) for a, b in zip(obs0, keep0)), "vector_reset observations were overwritten"
    for live, copy, infos, info_copy in kept:
        assert all(np.array_equal(a, b) for a, b in zip(live, copy)), "a later call rewrote a returned observation"
        assert [dict(infos[i]) for i in range(n)] == info_copy, "a later call changed a returned info"
        assert live[0].dtype == np.dtype(obs_dtype)
    # ... and once the caller has dropped them, their pinned blocks are reused: a steady-state loop allocates nothing
    del kept, live, copy, infos, info_copy, obs, obs0, o
    import gc
    gc.collect()
    blocks = len(env._free)
    assert blocks >= 60
    for t in range(50):
        env.vector_step(rng.integers(0, 2, n).tolist())
    assert len(env._free) == blocks, "blocks of dropped results are recycled, none were added"
    env.close()


def test_rllib_vector_env_adapter_hand_out_modes_agree():
    """the three hand-out modes of PcgrlVectorEnv return the same observations / rewards / dones / infos (2 000 envs, past
    an episode end with reset_at)"""
    from control_pcgrl_amd import PcgrlVectorEnv
    cfg = {"task": {"problem": "zelda", "map_shape": [16, 16], "obs_window": [32, 32], "weights": None},
           "representation": "turtle", "change_percentage": 0.05}
    n = 300
    envs = [PcgrlVectorEnv(cfg, num_envs=n, seeds=list(range(n)), obs_dtype=d, direct_host_outputs=k)
            for d, k in ((np.float32, None), (np.uint8, False), (np.uint8, True))]
    first = [e.vector_reset()[0] for e in envs]
    assert all(np.array_equal(np.stack(first[0]).astype(np.uint8), np.stack(f)) for f in first[1:])
    rng = np.random.default_rng(3)
    n_reset = 0
    for t in range(120):
        a = rng.integers(0, 12, n)
        outs = [e.vector_step(a) for e in envs]
        ref = outs[0]
        for o in outs[1:]:
            assert np.array_equal(np.stack(ref[0]).astype(np.uint8), np.stack(o[0])) and ref[1] == o[1] and ref[2] == o[2]
            assert dict(ref[4][7]) == dict(o[4][7])
        for i in np.nonzero(ref[2])[0]:
            rs = [e.reset_at(int(i))[0] for e in envs]
            assert all(np.array_equal(rs[0].astype(np.uint8), r) for r in rs[1:])
            n_reset += 1
    assert n_reset > 0
    for e in envs:
        e.close()


def test_rllib_vector_env_adapter_controllable_and_rep_wrappers():
    """cfg.controls through PcgrlVectorEnv: the 2 * n_ctrl constant planes in front of the one-hot channels, targets set
    per sub-env (control_wrappers.py:168-178, :189-214), against the reference's controllable episode; and static tiles +
    an action patch through the adapter against the reference's representation-wrapper episode."""
    from control_pcgrl_amd import PcgrlVectorEnv
    z = np.load(os.path.join(GOLDEN, "control_binary_narrow_s7.npz"))
    controls = [str(c) for c in z["controls"]]
    cfg = {"task": {"problem": "binary", "map_shape": [16, 16], "obs_window": [32, 32], "weights": {"path-length": 1, "regions": 1}},
           "representation": "narrow", "controls": controls}
    env = PcgrlVectorEnv(cfg, num_envs=2, seeds=[int(z["seed"]), 12345])
    K2 = 2 * len(controls)
    assert env.observation_space.shape == (32, 32, 3 + K2) and float(np.max(env.observation_space.high)) == 1.0
    n, t = int(z["steps_per_episode"]), 0
    for ep in range(len(z["reset_at"])):
        sub = env.get_sub_environments()[0]
        trgs_before = dict(sub.metric_trgs)
        sub.set_trgs({k: float(v) for k, v in zip(controls, z["reset_trg"][ep])})
        assert sub.metric_trgs == trgs_before, "queued targets are not the env's targets before its next reset (control_wrappers.py:174-178)"
        obs = env.vector_reset()[0] if ep == 0 else [env.reset_at(0)[0], env.reset_at(1)[0]]
        assert all(sub.metric_trgs[k] == float(v) for k, v in zip(controls, z["reset_trg"][ep]))
        assert obs[0].shape == (32, 32, 3 + K2) and np.all(obs[0][..., :K2] == obs[0][0, 0, :K2])
        assert np.allclose(obs[0][3, 4, :K2], z["reset_ctrl"][ep], rtol=1e-6, atol=1e-7)
        assert zlib.crc32(obs[0][..., K2:].astype(np.uint8).tobytes()) == int(z["reset_obs_crc"][ep])
        for _ in range(n):
            obs, rew, term, trunc, infos = env.vector_step([int(z["action"][t]), 0])
            assert abs(rew[0] - z["reward"][t]) <= 1e-5, f"reward @ {t}"  # (the adapter hands out float32 rewards)
            assert np.allclose(obs[0][0, 0, :K2], z["ctrl"][t], rtol=1e-6, atol=1e-7), f"ctrl planes @ {t}"
            assert zlib.crc32(obs[0][..., K2:].astype(np.uint8).tobytes()) == int(z["obs_crc"][t]), f"obs @ {t}"
            assert {k: infos[0][k] for k in ("regions", "path-length")} == dict(zip(("regions", "path-length"), z["stats"][t].tolist()))
            t += 1
    env.close()

    z = np.load(os.path.join(GOLDEN, "ext_zelda_narrow_aw2x2_sp10_sw3_s30.npz"))
    cfg = {"task": {"problem": "zelda", "map_shape": [16, 16], "obs_window": [32, 32], "weights": None}, "representation": "narrow",
           "static_prob": float(z["static_prob"]), "n_static_walls": int(z["n_static_walls"]), "act_window": [int(a) for a in z["act_window"]]}
    env = PcgrlVectorEnv(cfg, num_envs=1, seeds=[int(z["seed"])])
    n = int(z["steps_per_episode"])
    for ep, t0 in enumerate(z["reset_at"][:2]):
        obs = env.vector_reset()[0] if ep == 0 else [env.reset_at(0)[0]]
        assert np.array_equal(obs[0].astype(np.uint8).ravel(), z["reset_obs"][ep])
        for t in range(int(t0), int(t0) + n):
            obs, rew, term, trunc, infos = env.vector_step([z["action"][t]])
            assert zlib.crc32(obs[0].astype(np.uint8).tobytes()) == int(z["obs_crc"][t]), f"ext obs @ {t}"
            assert abs(rew[0] - z["reward"][t]) <= REW_TOL
    env.close()


def _solvable_rooms(n, seed, shape=(16, 16)):
    """sokoban maps that meet the solver's precondition (sokoban_prob.py:172-177): one small room carved into solid with
    one player and k crates / k targets -- what a trained generator produces most of the time."""
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


@pytest.mark.parametrize("shape", [(8, 8), (12, 20), (20, 20), (30, 30), (32, 32), (33, 17), (45, 6), (62, 32), (20, 40),
                                   (6, 62), (48, 33), (62, 62)])
def test_sokoban_solver_other_map_shapes_vs_oracle(shape):
    """the device solver with its helper wavefronts behind the lanes-per-env families it supports (8 / 16 / 32 / 64 lanes
    per env, 32- / 64-bit row masks: a bordered level is at most 64 x 64):
    playable rooms inside maps off the 16x16 point, one level per workgroup (pcgrl_stats_for_grids)"""
    g = _solvable_rooms(96, 31 + shape[0], shape)
    want = po.stats_for_grids("sokoban", g, solver_power=2000)
    assert (want[:, 4] != 8192).mean() > 0.9 and (want[:, 5] > 0).mean() > 0.1, "the solver runs, some levels are solved"
    env = _vec("sokoban", "narrow", shape, 4, solver_power=2000)
    got = env.stats_for_grids(torch.as_tensor(g).to(env.device)).cpu().numpy()
    assert np.array_equal(got, want)
    env.check_errors()


def test_golden_sokoban_solver_other_shapes_known_answers():
    """solver-firing levels on 8 x 8, 20 x 20 and 30 x 30 maps: the reference's SokobanCtrlProblem.get_stats answers"""
    for fname, keys in (("stats_sokoban_solver_shapes.npz", ("8x8", "20x20", "30x30")),
                        ("stats_sokoban_solver_wide.npz", ("20x40", "48x33", "62x62"))):  # (wide: 64-bit row masks)
        z = np.load(os.path.join(GOLDEN, fname))
        for key in keys:
            grids = z["grids_" + key]
            env = _vec("sokoban", "narrow", grids.shape[1:], 1, auto_reset=False)
            got = env.stats_for_grids(torch.as_tensor(grids)).cpu().numpy()
            assert np.array_equal(got, z["stats_" + key]), f"{key}: got {got.tolist()} want {z['stats_' + key].tolist()}"
            env.check_errors()


def test_golden_sokoban_solver_more_than_128_pairs():
    """levels with 129 .. 505 crate / target pairs against the reference's answers: more pairs than two registers per lane
    hold, searched with eight (SK_NH_HUGE) on the simulate wave alone, through stats_for_grids (helper waves present) and
    through a reset with injected maps (step-path kernels)"""
    z = np.load(os.path.join(GOLDEN, "stats_sokoban_solver_huge.npz"))
    for g, want, power, shape in zip(z["grids"], z["stats"], z["solver_power"], z["shapes"]):
        h, w = int(shape[0]), int(shape[1])
        grid = np.ascontiguousarray(g[:h, :w])
        env = _vec("sokoban", "narrow", (h, w), 3, auto_reset=False, solver_power=int(power))
        got = env.stats_for_grids(torch.as_tensor(grid[None])).cpu().numpy()
        assert np.array_equal(got[0], want), f"{h}x{w} power {int(power)}: got {got[0].tolist()} want {want.tolist()}"
        env.reset(init_grids=torch.as_tensor(np.repeat(grid[None], 3, 0)))
        st = env.get_state().stats.cpu().numpy()
        assert np.array_equal(st, np.repeat(want[None], 3, 0)), f"{h}x{w} through pcgrl_reset: {st.tolist()}"
        env.check_errors()  # nothing was refused


def test_fuzz_cases_with_more_than_128_pairs_replay():
    """the two cases the round-3 fuzzer reported as REFUSED (random 24 x 21 / 23 x 23 maps that met the solver's
    precondition with more than 128 pairs) replay green against the oracle"""
    import json
    import fuzz_parity as fz
    for line in ('{"problem": "sokoban", "kw": {"obs_window": [42, 28], "solver_power": 50}, "rep": "turtle", "shape": [24, 21], "n_envs": 809, "steps": 54, "bias": false, "mode": "mixed", "auto_reset": true, "seed": 480642379}',
                 '{"problem": "sokoban", "kw": {"change_percentage": 0.001, "solver_power": 500}, "rep": "wide", "shape": [23, 23], "n_envs": 68, "steps": 123, "bias": false, "mode": "adapter", "auto_reset": true, "seed_kind": "huge", "seed": 656225453}'):
        c = json.loads(line)
        seed = c.pop("seed")
        run = {"adapter": fz.run_adapter_case, "gym": fz.run_gym_case}.get(c.get("mode"), fz.run_case)
        assert run(c, seed) >= 0


def test_loss_integer_and_float64_forms_agree():
    """get_loss has an integer form (all static targets integral: every stock problem) and the float64 form; a
    non-integral target on a zero-weighted statistic switches an engine to the float64 form without changing any reward."""
    n, k = 512, 300
    w = {"regions": 1.0, "path-length": 0.0}
    a = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, weights=w)
    b = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, weights=w,
             static_trgs={"path-length": 48.5})
    a.reset()
    b.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.randint(0, 2, (k, n), generator=g, device="cuda", dtype=torch.int32)
    for i in range(k):
        ra = a.step(acts[i])[1].clone()
        rb = b.step(acts[i])[1]
        assert torch.equal(ra, rb), i
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa.last_loss, sb.last_loss) and torch.equal(sa.ep_return, sb.ep_return) and torch.equal(sa.stats, sb.stats)


def test_step_seq_equals_single_steps():
    """pcgrl_step_seq = the same launches as a loop over pcgrl_step (action rows taken round-robin from a pool)"""
    n, pool, k = 300, 7, 23
    a = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, change_percentage=0.05)
    b = _vec("binary", "narrow", (16, 16), n, seeds=np.arange(n), auto_reset=True, change_percentage=0.05)
    a.reset()
    b.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    rows = torch.randint(0, 2, (pool, n), generator=g, device="cuda", dtype=torch.int32)
    sp = torch.cuda.current_stream().cuda_stream
    assert a.step_seq_raw(rows.data_ptr(), n, pool, 3, k, sp) == 0
    for i in range(k):
        out = b.step(rows[(3 + i) % pool])
    sa, sb = a.get_state(), b.get_state()
    for f in ("grids", "pos", "counters", "stats", "last_loss", "ep_return"):
        assert torch.equal(getattr(sa, f), getattr(sb, f)), f
    assert torch.equal(a._obs, b._obs) and torch.equal(a._reward, b._reward) and torch.equal(a._done, b._done)
    a.check_errors()


def test_sokoban_solver_long_searches_vs_oracle():
    """Unsolvable open rooms (one target sits at the far end of an L-shaped alcove no crate can be pushed into) with 3-6
    crates at the largest accepted solver_power (16 000): all four stages run to the cap, the A* open lists grow past the
    part the helper waves keep in LDS (8 192 entries), so the sifts cross from LDS into the HBM part of the heap; the
    visited table fills to half (group overflow); the levels with six crates use hashed instead of exact keys."""
    rng = np.random.default_rng(11)
    n = 12
    g = np.ones((n, 16, 16), np.uint8)
    for i in range(n):
        h, w = int(rng.integers(8, 12)), int(rng.integers(9, 14))
        y0, x0 = int(rng.integers(3, 16 - h)), int(rng.integers(1, 16 - w))
        g[i, y0:y0 + h, x0:x0 + w] = 0
        g[i, y0 - 2:y0, x0 + 2] = 0
        g[i, y0 - 2, x0 + 3] = 4
        k = 3 + i % 4
        inner = [(y, x) for y in range(y0 + 1, y0 + h - 1) for x in range(x0 + 1, x0 + w - 1)]
        pick = rng.permutation(len(inner))[:2 * k]
        for c, t in zip(pick, [2] + [3] * k + [4] * (k - 1)):
            g[i, inner[c][0], inner[c][1]] = t
    want = po.stats_for_grids("sokoban", g, solver_power=16000)
    assert (want[:, 4] != 8192).all() and (want[:, 5] == 0).all(), "the solver runs on every level and never wins"
    env = _vec("sokoban", "narrow", (16, 16), 4, solver_power=16000)
    got = env.stats_for_grids(torch.as_tensor(g).to(env.device)).cpu().numpy()
    assert np.array_equal(got, want), (got, want)
    env.check_errors()


def test_sokoban_wide_2048_envs_solver_inside_episodes_vs_oracle():
    """BASELINE configs[3] at its batch size with the solver FIRING inside step launches: playable levels are injected,
    the agent edits floor / wall cells, and dist-win / sol-length move the reward; hundreds of waves hold workspace slots
    at the same time.  Every step's stats, reward and done against the oracle."""
    n, power = 2048, 400
    env = _vec("sokoban", "wide", (16, 16), n, seeds=np.arange(n), auto_reset=True, solver_power=power)
    orc = po.OracleVecEnv("sokoban", "wide", (16, 16), n, seeds=np.arange(n), threads=16, solver_power=power)
    maps = _solvable_rooms(n, 5)
    env.reset(init_grids=torch.as_tensor(maps))
    orc.reset(init_grids=maps)
    st0 = env.get_state().stats.cpu().numpy()
    assert np.array_equal(st0, orc.get_state()["stats"])
    assert (st0[:, 4] != 8192).all() and (st0[:, 5] > 0).mean() > 0.15, "every injected level runs the solver, a fifth is solvable"
    rng = np.random.default_rng(9)
    active = solved = 0
    for t in range(24):
        # a generator-like edit that keeps most levels playable: grow the room by one floor cell next to it (a floor cell
        # anywhere else would be a second region), or drop a wall / crate / target somewhere
        grids = orc.get_state()["grids"].reshape(n, 16, 16)
        open_ = np.pad(grids != 1, ((0, 0), (1, 1), (1, 1)))
        near = (open_[:, :-2, 1:-1] | open_[:, 2:, 1:-1] | open_[:, 1:-1, :-2] | open_[:, 1:-1, 2:]) & (grids == 1)
        cell = rng.integers(0, 256, n)
        tile = rng.choice([1, 1, 3, 4], n)
        for i in np.nonzero(rng.random(n) < 0.6)[0]:
            cand = np.flatnonzero(near[i])
            if len(cand):
                r, c_ = divmod(int(rng.choice(cand)), 16)
                cell[i], tile[i] = c_ * 16 + r, 0  # the wide action's cell index is column-major (wide_rep.py:40-45)
        a = (cell * 5 + tile).astype(np.int32)
        obs, rew, done, _, info = env.step(torch.as_tensor(a).to(env.device))
        oobs, orew, odone, ostats = orc.step(a, auto_reset=True, want_obs=(t % 8 == 7))
        got = info["stats"].cpu().numpy()
        assert np.array_equal(got, ostats), f"stats @ {t}: {np.nonzero((got != ostats).any(1))[0][:5]}"
        assert np.max(np.abs(rew.cpu().numpy() - orew)) <= REW_TOL, f"reward @ {t}"
        assert np.array_equal(done.cpu().numpy(), odone)
        if t % 8 == 7:
            assert np.array_equal(obs.cpu().numpy(), oobs)
        active += int((got[:, 4] != 8192).sum())
        solved += int((got[:, 5] > 0).sum())
    assert active > 0.12 * 24 * n and solved > 0.03 * 24 * n, (active, solved)  # thousands of searches inside step launches
    env.check_errors()


@pytest.mark.parametrize("problem,rep", [("binary", "narrow"), ("zelda", "turtle")])
def test_step_graph_replay_vs_oracle(problem, rep):
    """VecPcgrlEnv.step captured once in a HIP graph (static action buffer) and replayed with fresh actions across
    auto-resets equals the oracle step for step; a replay AFTER pcgrl_update -- the captured launch is the compile-time
    16x16 kernel, which has no code for the statistics pcgrl_update leaves stale -- is reported (PCGRL_ESTALE), never a
    silent divergence; after a reset the same graph is good again.  (include/pcgrl_amd.h, "HIP graphs")"""
    n, shape = 192, (16, 16)
    seeds = 40 + np.arange(n)
    env = _vec(problem, rep, shape, n, seeds=seeds, auto_reset=True)
    orc = po.OracleVecEnv(problem, rep, shape, n, seeds=seeds, threads=8)
    env.reset(); orc.reset()
    gen = torch.Generator().manual_seed(3)
    static_a = torch.zeros(n, dtype=torch.int32, device=env.device)

    def draw():
        return torch.randint(0, env.num_actions, (n,), generator=gen, dtype=torch.int32)

    for _ in range(3):  # (eager warm-up before the capture)
        a = draw()
        env.step(a.to(env.device)); orc.step(a.numpy(), auto_reset=True, want_obs=False)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            obs, rew, done, _, info = env.step(static_a)
    torch.cuda.current_stream().wait_stream(side)
    T = 1000  # episode length 770: the replays cross the auto-reset
    for t in range(T):
        a = draw()
        static_a.copy_(a)
        graph.replay()
        want_obs = t % 97 == 0 or t == T - 1
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=want_obs)
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL, f"reward @ {t}"
        assert np.array_equal(done.cpu().numpy(), odone), f"done @ {t}"
        if want_obs:
            assert np.array_equal(obs.cpu().numpy(), oobs), f"obs @ {t}"
    env.check_errors()
    # pcgrl_update leaves statistics stale; the captured launch cannot handle that and must say so
    for _ in range(6):
        a = draw()
        env.update(a.to(env.device), want_obs=False); orc.update(a.numpy())
    for t in range(40):
        static_a.copy_(draw())
        graph.replay()
    with pytest.raises(RuntimeError, match="ESTALE"):
        env.check_errors()
    # ... and a full reset makes the same graph valid again
    env.reset(); orc.reset()
    for t in range(60):
        a = draw()
        static_a.copy_(a)
        graph.replay()
        oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True, want_obs=(t == 59))
        assert np.array_equal(info["stats"].cpu().numpy(), ostats), f"stats after reset @ {t}"
        assert np.max(np.abs(rew.cpu().numpy().astype(np.float64) - orew)) <= REW_TOL
    assert np.array_equal(obs.cpu().numpy(), oobs)
    env.check_errors()


def test_fuzz_sweep_fixed_seed():
    """tests/fuzz_parity.py: 250 random supported configurations (problem x representation x map shape x window x change
    budget x wrappers x controls) against the oracle, every step.  Longer sweeps: python tests/fuzz_parity.py --cases N"""
    import fuzz_parity
    failures = fuzz_parity.sweep(250, 20261002, verbose=False, stop_on_fail=False)
    assert not failures, failures[:3]
    assert not fuzz_parity.sweep.refused, fuzz_parity.sweep.refused[:3]  # nothing the generator draws is refused at run time


def test_injected_maps_with_action_patch_ignore_init_pos():
    """with an action patch the edit position is a function of the step counter alone: a masked reset that injects maps
    AND positions starts at the first patch centre, like the oracle (found by tests/fuzz_parity.py)"""
    n, shape, kw = 45, (20, 28), dict(act_window=[4, 3])
    seeds = 77 + np.arange(n)
    env = _vec("binary", "narrow", shape, n, seeds=seeds, auto_reset=True, **kw)
    orc = po.OracleVecEnv("binary", "narrow", shape, n, seeds=seeds, **kw)
    assert np.array_equal(env.reset()[0].cpu().numpy(), orc.reset())
    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(3)
    for rnd in range(3):
        mask = (rng.random(n) < 0.6).astype(np.uint8)
        grids = rng.integers(0, 2, size=(n,) + shape, dtype=np.uint8)
        pos = np.stack([rng.integers(0, s, size=n) for s in shape], axis=1).astype(np.int32)
        obs, _ = env.reset(mask=mask, init_grids=grids, init_pos=pos)
        assert np.array_equal(obs.cpu().numpy(), orc.reset(mask=mask, init_grids=grids, init_pos=pos)), f"obs after inject {rnd}"
        for t in range(12):
            a = torch.randint(0, 2, (n, 12), generator=g, dtype=torch.int32)
            obs, rew, done, _, info = env.step(a.to(env.device))
            oobs, orew, odone, ostats = orc.step(a.numpy(), auto_reset=True)
            assert np.array_equal(info["stats"].cpu().numpy(), ostats) and np.array_equal(obs.cpu().numpy(), oobs), (rnd, t)
    assert np.array_equal(env.get_state().pos.cpu().numpy()[:, :2], orc.get_state()["pos"][:, :2])
    env.check_errors()


def test_stats_for_grids_large_sokoban_batch_on_a_small_engine():
    """a batch much larger than the engine's own env count grows the solver's workspace pool once (one level per workgroup,
    up to 256 searches at a time); answers equal the oracle's before and after the growth"""
    g = _solvable_rooms(700, 5)
    want = po.stats_for_grids("sokoban", g, solver_power=1500)
    env = _vec("sokoban", "narrow", (16, 16), 1, solver_power=1500)
    small = env.stats_for_grids(torch.as_tensor(g[:3]).to(env.device)).cpu().numpy()
    assert np.array_equal(small, want[:3])
    for _ in range(2):
        got = env.stats_for_grids(torch.as_tensor(g).to(env.device)).cpu().numpy()
        assert np.array_equal(got, want)
    env.reset()
    a = torch.zeros(1, dtype=torch.int32, device=env.device)
    env.step(a)  # the engine's own step path uses the grown pool, too
    env.check_errors()
