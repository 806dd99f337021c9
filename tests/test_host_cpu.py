"""CPU-side checks (no GPU): C-ABI library loads and exports every declared symbol, host tables agree with the
oracle's independent tables, config building mirrors the reference's derived constants, and the multi-process
episodic-return reduction works over gloo (world_size 2)."""
import os
import re
import subprocess
import time
import sys

import numpy as np
import pytest

import pcgrl_oracle as po
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from control_pcgrl_amd import _lib
    _lib.build()
    header = open(os.path.join(ROOT, "include", "pcgrl_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pcgrl_[a-z_]+)\s*\(", header))
    assert declared >= {"pcgrl_create", "pcgrl_step", "pcgrl_reset", "pcgrl_destroy", "pcgrl_get_state"}
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/pcgrl_amd.h but not exported"
    assert set(_lib.SYMBOLS) == declared
    assert b"gfx950" in L.pcgrl_version()


def test_code_object_targets_gfx950():
    from control_pcgrl_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_create_rejects_bad_configs_without_gpu():
    """validation happens before any HIP call"""
    import ctypes as C
    from control_pcgrl_amd import _lib
    from control_pcgrl_amd.vec_env import build_config
    L = _lib.lib()
    h = C.c_void_p()
    cfg, _, _ = build_config("binary", "wide", (16, 16), obs_window=(32, 32))
    assert L.pcgrl_create(C.byref(cfg), 4, 0, C.byref(h)) == 1  # EINVAL: wide needs obs_window == map_shape
    assert b"obs_window" in L.pcgrl_last_error()
    cfg, _, _ = build_config("sokoban", "narrow", (16, 63))
    assert L.pcgrl_create(C.byref(cfg), 4, 0, C.byref(h)) == 2  # EUNSUPPORTED: the device solver's level is at most 64 x 64 with its border
    cfg, _, _ = build_config("minecraft_3D_maze", "narrow", (17, 7, 7))
    assert L.pcgrl_create(C.byref(cfg), 4, 0, C.byref(h)) == 2  # EUNSUPPORTED: 3-D maps up to 16 x 16 x 16
    with pytest.raises(ValueError):
        build_config("binary", "cellular", (16, 16))
    with pytest.raises(ValueError):
        build_config("smb", "narrow", (16, 16))


def test_round5_entry_points_check_their_arguments_without_gpu():
    """the entry points added in 0.3 validate before any HIP call (null handles / pointers are errors, not crashes)"""
    from control_pcgrl_amd import _lib
    L = _lib.lib()
    assert L.pcgrl_num_actions(None) == -1
    assert L.pcgrl_sample_actions(None, None, 0, None) == 1 and b"pcgrl_sample_actions" in L.pcgrl_last_error()
    assert L.pcgrl_reserve_solver_pool(None, 0, 1) == 1
    assert L.pcgrl_solver_pool_slots(None, None, None) == 0
    assert L.pcgrl_copy_to_host(None, None, 16, None) == 1
    assert L.pcgrl_graph_upload(None, None) == 1
    import control_pcgrl_amd
    assert control_pcgrl_amd.__version__.encode() in L.pcgrl_version()  # one version number, two places


def test_round6_entry_points_check_their_arguments_without_gpu():
    """asynchronous stepping / target resampling / rollout form: null handles and pointers are errors, not crashes"""
    from control_pcgrl_amd import _lib
    L = _lib.lib()
    assert L.pcgrl_set_solver_budget(None, 16) == 1 and b"pcgrl_set_solver_budget" in L.pcgrl_last_error()
    assert L.pcgrl_get_solver_budget(None) == -1
    assert L.pcgrl_step_ready(None, None, 1, None, None, None, None, None, None) == 1
    assert L.pcgrl_env_busy(None, None, None) == 1
    assert L.pcgrl_set_target_resampling(None, 1, 0, None, None) == 1
    assert L.pcgrl_set_rollout_form(None, 0) == 1


def test_sub_batched_env_validates_the_split():
    from control_pcgrl_amd import SubBatchedVecEnv
    with pytest.raises(ValueError, match="multiple of sub_batches"):
        SubBatchedVecEnv("binary", "narrow", (16, 16), 10, 4)


@pytest.mark.parametrize("problem,shape", [("binary", (16, 16)), ("binary", (32, 32)), ("binary", (10, 14)),
                                           ("zelda", (16, 16)), ("zelda", (32, 32)), ("sokoban", (16, 16)),
                                           ("minecraft_3D_maze", (7, 7, 7))])
def test_host_tables_match_oracle_tables(problem, shape):
    from control_pcgrl_amd.vec_env import build_config
    for rep in ("narrow", "turtle", "wide"):
        if problem == "minecraft_3D_maze" and rep != "narrow":
            continue
        c, spec, ow = build_config(problem, rep, shape)
        o = po.make_config(problem, rep, shape)
        for f in ("problem", "representation", "ndim", "max_iterations", "max_changes", "n_stats", "solver_power"):
            assert getattr(c, f) == getattr(o, f), f
        for f in ("dims", "obs_window"):
            assert list(getattr(c, f)) == list(getattr(o, f)), f
        for f in ("has_trg", "weights", "trg_lo", "trg_hi"):
            assert list(getattr(c, f)) == list(getattr(o, f)), f
        assert spec.stat_keys == po.STAT_KEYS[problem]


def test_reference_derived_constants():
    """values the survey probed from the reference"""
    from control_pcgrl_amd.problems import problem_spec, target_interval
    assert problem_spec("binary", (16, 16)).static_trgs["path-length"] == 136
    z = problem_spec("zelda", (16, 16)).static_trgs
    assert z["path-length"] == 271 and target_interval(z["nearest-enemy"]) == (5.0, 143.0)
    assert target_interval(z["enemies"]) == (2.0, 4.0)
    s = problem_spec("sokoban", (16, 16)).static_trgs
    assert s["sol-length"] == 20 and target_interval(s["crate"]) == (2.0, 2.0) and "target" not in s
    m = problem_spec("minecraft_3D_maze", (7, 7, 7)).static_trgs
    assert m["path-length"] == 12700 and m["n_jump"] == 5
    from control_pcgrl_amd.vec_env import build_config
    c, _, _ = build_config("binary", "narrow", (16, 16), change_percentage=0.2)
    assert c.max_iterations == 769 and c.max_changes == 51


def test_shard_ranges_cover_everything():
    from control_pcgrl_amd.dist import shard_env_range, shard_seeds
    for total, ws in ((4096 * 8, 8), (10, 3), (7, 8)):
        ranges = [shard_env_range(total, r, ws) for r in range(ws)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(ws - 1))
    assert shard_seeds(100, 10, 1, 3) == [104, 105, 106]


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from control_pcgrl_amd.dist import EpisodeStatsReducer, shard_env_range
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
lo, hi = shard_env_range(10, rank, 2)
red = EpisodeStatsReducer(2, "cpu")
n = hi - lo
done = torch.tensor([(g % 2 == 0) for g in range(lo, hi)])
ret = torch.tensor([float(g) for g in range(lo, hi)])
ln = torch.full((n,), 770, dtype=torch.int32)
fs = torch.stack([torch.arange(lo, hi), torch.arange(lo, hi) * 2], 1).to(torch.int32)
red.update(done, ret, ln, fs)
out = red.reduce()
assert out["episodes"] == 5.0, out
assert abs(out["mean_return"] - (0 + 2 + 4 + 6 + 8) / 5) < 1e-12, out
assert out["mean_length"] == 770.0
assert out["mean_final_stats"] == [4.0, 8.0], out
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_episode_stats_reduction_gloo_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


def test_bench_bfs_active_maps_have_one_player_key_door():
    """bench.py's SURVEY 8(d) "BFS-active" zelda workload: exactly one player / key / door per injected map, and the
    restricted action set never places one"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import numpy as np
    g = bench.bfs_active_maps(64, 3)
    assert g.shape == (64, 16, 16)
    for t in (2, 3, 4):
        assert ((g == t).sum(axis=(1, 2)) == 1).all()
    assert not {4 + 2, 4 + 3, 4 + 4} & set(bench.BFS_ACTIONS)
    assert set(bench.WORKLOADS) == set(bench.ALGO_BYTES)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (gloo rendezvous on 127.0.0.1), shards the
    envs with contiguous seed ranges, and rank 0 alone prints the line; --gpus N under a mismatching WORLD_SIZE fails."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--envs", "100"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["global_envs"] == 200
    assert out["first_last_seed_per_rank"] == [[0x5EED, 0x5EED + 99], [0x5EED + 100, 0x5EED + 199]]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_rank_that_hangs_exits_nonzero():
    """a rank that never reaches the rendezvous ends itself with exit code 3 after PCGRL_BENCH_RANK_TIMEOUT (never a re-exec,
    never an in-process fallback), the launcher stops the other rank and the run fails -- instead of hanging in a collective"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PCGRL_BENCH_TEST_HANG_RANK="1", PCGRL_BENCH_TEST_HANG_TIMEOUT="4", PCGRL_BENCH_RANK_TIMEOUT="120")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--envs", "100"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0, r.stdout + r.stderr
    assert "rank 1 did not finish within" in r.stderr and "rank 1 exited with code 3" in r.stderr, r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no line from a failed run"
    assert time.time() - t0 < 120


def test_bench_solver_active_maps_are_playable():
    """bench.py's "solver-active" sokoban workload: every injected level meets the solver's precondition (one player,
    crates == targets > 0, one region), and the action pool only places floor / wall on cells in and around the room"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    maps, cells = bench.solver_active_maps(96, 5)
    st = po.stats_for_grids("sokoban", maps, solver_power=300)
    assert (st[:, 0] == 1).all() and (st[:, 1] == st[:, 2]).all() and (st[:, 1] > 0).all() and (st[:, 3] == 1).all()
    assert (st[:, 4] != 8192).all() and (st[:, 5] > 0).any()
    a = bench.solver_active_actions(cells, 32, 1)
    assert a.shape == (32, 96) and (a % 5 <= 1).all() and a.min() >= 0 and a.max() < 1280
    for i in range(96):
        assert set((a[:, i] // 5).tolist()) <= set(cells[i].tolist())


def test_empty_static_target_range_is_refused_like_the_reference():
    """zelda's nearest-enemy target is the range (5, ceil(w / 2 + 1) * h): on a 1 x 5 map np.arange(5, 4) is empty and the
    reference's get_loss raises (min of an empty array, control_wrappers.py:339) -- so do the host config builder and the
    oracle, instead of inventing an interval (found by tests/fuzz_parity.py)."""
    from control_pcgrl_amd.vec_env import build_config
    with pytest.raises(ValueError, match="empty target range"):
        build_config("zelda", "narrow", (1, 5))
    with pytest.raises(ValueError, match="empty target range"):
        po.make_config("zelda", "narrow", (1, 5))
    build_config("zelda", "narrow", (2, 5))  # (5, 8): fine
    po.make_config("zelda", "narrow", (2, 5))


def test_fuzzer_draws_only_configurations_the_engine_validates():
    """tests/fuzz_parity.py's generator stays inside the accepted configuration space: every drawn case builds a config on
    the host and passes pcgrl_create's validation (which runs before any HIP call: without a GPU the call then fails
    with PCGRL_EHIP, never EINVAL / EUNSUPPORTED), or is refused by host and oracle alike (empty target ranges)."""
    import ctypes as C
    import fuzz_parity
    from control_pcgrl_amd import _lib
    from control_pcgrl_amd.vec_env import build_config
    L = _lib.lib()
    rng = np.random.default_rng(12345)
    refused = 0
    for i in range(400):
        case = fuzz_parity.draw_case(rng)
        kw = {k: (tuple(v) if k == "obs_window" else v) for k, v in case["kw"].items()}
        try:
            cfg, _, _ = build_config(case["problem"], case["rep"], tuple(case["shape"]), **kw)
        except ValueError:
            with pytest.raises(ValueError):
                po.make_config(case["problem"], case["rep"], tuple(case["shape"]), **kw)
            refused += 1
            continue
        po.make_config(case["problem"], case["rep"], tuple(case["shape"]), **kw)
        h = C.c_void_p()
        rc = L.pcgrl_create(C.byref(cfg), case["n_envs"], 0, C.byref(h))
        assert rc not in (1, 2), (case, L.pcgrl_last_error())
        if rc == 0:  # (a GPU is present after all)
            L.pcgrl_destroy(h)
    assert refused < 40


def test_bench_workload_tables_are_consistent():
    """every bench workload has its algorithmic bytes (action + map read / 1 B write + uint8 one-hot window +
    reward / done / stats / pos) and they follow from the shapes; the launcher's GPU count needs no GPU runtime"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.ALGO_BYTES) == set(bench.WORKLOADS)
    n_stats = {"binary": 2, "zelda": 7, "sokoban": 7, "minecraft_3D_maze": 3}
    n_tiles = {"binary": 2, "zelda": 8, "sokoban": 5, "minecraft_3D_maze": 2}
    for w in ("binary-narrow", "zelda-turtle", "sokoban-wide", "minecraft_3D_maze-narrow", "binary_big-narrow", "binary_bigger-narrow",
              "zelda_big-turtle", "zelda_bigger-turtle", "minecraft_3D_maze-narrow-15"):
        problem, rep, shape, _ = bench.WORKLOADS[w][:4]
        cells = int(np.prod(shape))
        if rep == "wide":
            obs, pos = cells * n_tiles[problem], 0
        elif len(shape) == 3:
            obs, pos = int(np.prod([2 * s for s in shape])) * 4, 3  # out of bounds, AIR, DIRT, path overlay
        else:
            obs, pos = int(np.prod([2 * s for s in shape])) * (n_tiles[problem] + 1), 2
        assert bench.ALGO_BYTES[w] == 4 + cells + 1 + obs + 4 + 1 + 4 * n_stats[problem] + pos, w
    n = bench.gpus_without_runtime()
    assert n is None or (isinstance(n, int) and n >= 0)


def test_adapter_lease_returns_its_block_when_the_last_array_is_gone():
    """PcgrlVectorEnv hands out numpy arrays over pinned blocks: a block may only come back to the free list when the last
    array (view, or view of a view) over it has been garbage-collected -- the mechanism, on a plain CPU tensor"""
    import gc
    torch = pytest.importorskip("torch")
    from control_pcgrl_amd.rllib_env import _Lease
    free = []
    block = torch.arange(64, dtype=torch.uint8)
    lease = _Lease(block, free)
    h = np.asarray(lease)
    assert h.shape == (64,) and h[5] == 5 and not h.flags.owndata
    obs = h[:32].reshape(2, 4, 4)
    rows = list(obs)           # what vector_step returns: one view per env
    rew = h[32:48].view(np.float32)
    del lease, h, obs
    gc.collect()
    assert free == [], "views are alive: the block is still lent"
    keep = rows[1][2:]         # a view of a view keeps it, too
    del rows, rew
    gc.collect()
    assert free == []
    block[16 + 8 + 0] = 200     # (the memory really is the block's)
    assert keep[0, 0] == 200
    del keep
    gc.collect()
    assert len(free) == 1 and free[0] is block


# ------------------------------------------------------------------------------------- the bench line's profile echoes
def test_bench_echoes_only_profiles_taken_on_these_kernels(monkeypatch):
    """roofline.traffic / kernel_mean_us are echoes of committed rocprofv3 summaries: accepted when the summary's
    kernel_sources_sha16 (per kernel family) equals the checkout's, refused -- with the reason -- when it does not."""
    import bench

    fam2, fam3 = bench.kernel_sources_sha16("2d"), bench.kernel_sources_sha16("3d")
    assert bench.kernel_sources_sha16() == {"2d": fam2, "3d": fam3} and fam2 != fam3 and len(fam2) == 16
    # the committed round-6 summary belongs to the committed kernels (tools/finish_round.py wrote both)
    import json
    import os
    s = json.load(open(os.path.join(bench.ROOT, "profiles", "r06_summary.json")))
    assert s["kernel_sources_sha16"] == {"2d": fam2, "3d": fam3}, "profiles/r06_summary.json was not taken on the kernels of this checkout"
    assert isinstance(s.get("profiled_at_head"), str) and len(s["profiled_at_head"]) >= 7
    traffic, source, head = bench.profiled_traffic("binary-narrow", 4096)
    assert traffic and traffic > 13e6 and "r06_summary.json" in source and head == s["profiled_at_head"]
    assert bench.profiled_kernel_mean_us("binary-narrow", 4096) > 4.0
    t3, src3, head3 = bench.profiled_traffic("minecraft_3D_maze-narrow", 1024)
    assert t3 and "r06_summary.json" in src3 and head3 == s["profiled_at_head"]
    # other kernels than the profiled ones: nothing is echoed, and the line says why
    real = bench.kernel_sources_sha16
    monkeypatch.setattr(bench, "kernel_sources_sha16", lambda family=None: "0" * 16 if family else {"2d": "0" * 16, "3d": "0" * 16})
    traffic, source, head = bench.profiled_traffic("binary-narrow", 4096)
    assert traffic is None and head is None and "other kernels" in source and "0" * 16 in source
    assert bench.profiled_kernel_mean_us("binary-narrow", 4096) is None
    monkeypatch.setattr(bench, "kernel_sources_sha16", real)
    # a size that was never profiled: no echo, no reason needed
    assert bench.profiled_traffic("binary-narrow", 12345)[0] is None


def test_kernel_hash_ignores_the_host_side_and_splits_the_families(tmp_path, monkeypatch):
    """pcgrl_engine.hip (argument checks, allocation, launches) is not part of the kernels' identity; the 3-D header is part of
    the 3-D family's only."""
    import shutil
    import bench

    root = tmp_path / "repo"
    shutil.copytree(os.path.join(bench.ROOT, "control_pcgrl_amd", "csrc"), root / "control_pcgrl_amd" / "csrc",
                    ignore=shutil.ignore_patterns("*.so", "*.o", "_obj*"))
    shutil.copytree(os.path.join(bench.ROOT, "include"), root / "include")
    monkeypatch.setattr(bench, "ROOT", str(root))
    h2, h3 = bench.kernel_sources_sha16("2d"), bench.kernel_sources_sha16("3d")
    with open(root / "control_pcgrl_amd" / "csrc" / "pcgrl_engine.hip", "a") as f:
        f.write("\n// host-side edit\n")
    assert (bench.kernel_sources_sha16("2d"), bench.kernel_sources_sha16("3d")) == (h2, h3)
    with open(root / "control_pcgrl_amd" / "csrc" / "pcgrl_kernels3d.h", "a") as f:
        f.write("\n// 3-D kernel edit\n")
    assert bench.kernel_sources_sha16("2d") == h2 and bench.kernel_sources_sha16("3d") != h3
    with open(root / "control_pcgrl_amd" / "csrc" / "pcgrl_kernels2d.h", "a") as f:
        f.write("\n// 2-D kernel edit\n")
    assert bench.kernel_sources_sha16("2d") != h2
