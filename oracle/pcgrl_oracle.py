"""ctypes binding + problem tables for the CPU oracle (oracle/pcgrl_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (control_pcgrl_amd/) never imports this module.

The tables below restate the reference's per-problem constants independently of the product's
control_pcgrl_amd/problems.py (two statements of the same reference lines, both pinned by golden rewards).
Reference paths are relative to /root/reference/control_pcgrl/.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpcgrl_oracle.so")

PROBLEMS = {"binary": 0, "zelda": 1, "sokoban": 2, "minecraft_3D_maze": 3}
REPRESENTATIONS = {"narrow": 0, "turtle": 1, "wide": 2}
N_TILES = {"binary": 2, "zelda": 8, "sokoban": 5, "minecraft_3D_maze": 2}
STAT_KEYS = {
    "binary": ["regions", "path-length"],
    "zelda": ["player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length"],
    "sokoban": ["player", "crate", "target", "regions", "dist-win", "sol-length", "ratio"],
    "minecraft_3D_maze": ["regions", "path-length", "n_jump"],
}
# cfg.task.weights of the reference's task configs
DEFAULT_WEIGHTS = {
    "binary": {"path-length": 1, "regions": 1},  # configs/task/binary.yaml:5-7
    "zelda": {"player": 3, "key": 3, "door": 3, "regions": 5, "enemies": 1, "nearest-enemy": 2,
              "path-length": 1},  # configs/task/zelda.yaml:8-16
    "sokoban": {"player": 3, "crate": 2, "target": 2, "regions": 5, "ratio": 2, "dist-win": 0,
                "sol-length": 1},  # configs/config.py:94-102
    "minecraft_3D_maze": {"path-length": 100, "n_jump": 100, "regions": 0},  # configs/config.py:161-167
}


def static_targets(problem, map_shape):
    """static_trgs of the reference's problem classes, evaluated for `map_shape`."""
    if problem == "binary":  # probs/binary/binary_prob.py:50, 59-63
        h, w = map_shape
        return {"regions": 1, "path-length": math.ceil(w / 2) * h + math.floor(h / 2)}
    if problem == "zelda":  # probs/zelda/zelda_ctrl_prob.py:19-45, zelda_prob.py:30
        h, w = map_shape
        max_nearest = math.ceil(w / 2 + 1) * h
        max_path = (math.ceil(w / 2) * h + math.floor(h / 2)) * 2 - 1
        return {"enemies": (2, 5), "path-length": max_path, "nearest-enemy": (5, max_nearest), "regions": 1,
                "player": 1, "key": 1, "door": 1}
    if problem == "sokoban":  # sokoban_prob.py:30-31 freezes 5x5 before sokoban_ctrl_prob.py:13, 27-35 (Q10)
        w = h = 5
        max_path = math.ceil(w / 2 + 1) * h
        return {"player": 1, "crate": (2, 3), "regions": 1, "ratio": 0, "dist-win": 0, "sol-length": max_path}
    if problem == "minecraft_3D_maze":  # minecraft_3D_maze_prob.py:33-58: sizes frozen at 15 (Q11)
        length = width = height = 15
        per_floor = math.ceil(width / 2) * length + math.floor(length / 2)
        max_path = 2 * (height // 3) * per_floor
        return {"regions": 1, "path-length": 10 * max_path, "n_jump": 5}
    raise ValueError(problem)


class OrcConfig(C.Structure):
    _fields_ = [
        ("problem", C.c_int32), ("representation", C.c_int32), ("ndim", C.c_int32),
        ("dims", C.c_int32 * 3), ("obs_window", C.c_int32 * 3),
        ("max_iterations", C.c_int32), ("max_changes", C.c_int32), ("n_stats", C.c_int32),
        ("has_trg", C.c_int32 * 8), ("weights", C.c_double * 8), ("trg_lo", C.c_double * 8),
        ("trg_hi", C.c_double * 8), ("solver_power", C.c_int32),
        ("n_ctrl", C.c_int32), ("ctrl_idx", C.c_int32 * 8), ("ctrl_range", C.c_double * 8),
        ("act_window", C.c_int32 * 3), ("static_tiles", C.c_int32), ("n_static_walls", C.c_int32),
        ("static_eval", C.c_int32), ("static_prob", C.c_double),
    ]


def cond_bounds(problem, map_shape):
    """Problem.cond_bounds (only the keys usable as controls), evaluated for `map_shape`."""
    if problem == "binary":  # probs/binary/binary_prob.py:66-84
        h, w = map_shape
        return {"regions": (0, w * math.ceil(h / 2)), "path-length": (0, math.ceil(w / 2) * h + math.floor(h / 2))}
    if problem == "zelda":  # probs/zelda/zelda_ctrl_prob.py:55-73
        h, w = map_shape
        n = w * h
        return {"nearest-enemy": (0, math.ceil(w / 2 + 1) * h), "enemies": (0, n - 2), "player": (0, n - 2),
                "key": (0, n - 2), "door": (0, n - 2), "regions": (0, n / 2),
                "path-length": (0, (math.ceil(w / 2) * h + math.floor(h / 2)) * 2 - 1)}
    if problem == "sokoban":  # probs/sokoban/sokoban_ctrl_prob.py:37-49, frozen at 5x5 (Q10)
        w = h = 5
        mp = math.ceil(w / 2 + 1) * h
        return {"player": (1, w * h), "crate": (1, w * h / 2 - max(w, h)), "target": (1, w * h), "ratio": (0, w * h),
                "dist-win": (0, w * h * (w + h)), "sol-length": (0, 2 * mp), "regions": (0, w * h / 2)}
    if problem == "minecraft_3D_maze":  # minecraft_3D_maze_prob.py:52-58, sizes frozen at 15 (Q11)
        mp = 2 * (15 // 3) * (math.ceil(15 / 2) * 15 + math.floor(15 / 2))
        return {"regions": (0, math.ceil(15 * 15 / 2 * 15)), "path-length": (0, mp), "n_jump": (0, mp // 2)}
    raise ValueError(problem)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("pcgrl_oracle.c", "sokoban_solver.c", "mc3d.c", "pcgrl_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig), C.c_int32]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int32]
        L.orc_seed.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_step_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_observe.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_obs_size.restype = C.c_int64
        L.orc_obs_size.argtypes = [C.c_void_p]
        L.orc_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_get_last_episode.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_stats_for_grids.argtypes = [C.POINTER(OrcConfig), C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_stats_for_grids_mt.argtypes = [C.POINTER(OrcConfig), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_queue_targets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_ctrl_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_refresh_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_static.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_static.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_int32]
        L.orc_rng_probe.argtypes = [C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def make_config(problem, representation, map_shape, obs_window=None, weights=None, max_board_scans=3,
                change_percentage=None, solver_power=10000, controls=None, act_window=None, static_prob=None,
                n_static_walls=None, static_eval=False):
    map_shape = tuple(int(s) for s in map_shape)
    ndim = len(map_shape)
    if obs_window is None:
        obs_window = map_shape if representation == "wide" else tuple(2 * s for s in map_shape)
    if weights is None:
        weights = DEFAULT_WEIGHTS[problem]
    keys = STAT_KEYS[problem]
    trgs = static_targets(problem, map_shape)
    cfg = OrcConfig()
    cfg.problem = PROBLEMS[problem]
    cfg.representation = REPRESENTATIONS[representation]
    cfg.ndim = ndim
    for d in range(3):
        cfg.dims[d] = map_shape[d] if d < ndim else 1
        cfg.obs_window[d] = int(obs_window[d]) if d < ndim else 1
    n_cells = int(np.prod(map_shape))
    cfg.max_iterations = n_cells * max_board_scans + 1  # envs/pcgrl_env.py:241
    cfg.max_changes = -1 if change_percentage is None else max(int(change_percentage * n_cells), 1)  # :235-239
    cfg.n_stats = len(keys)
    for i, k in enumerate(keys):
        cfg.weights[i] = float(weights.get(k, 0))  # control_wrappers.py:41-45
        if k in trgs:
            t = trgs[k]
            cfg.has_trg[i] = 1
            if isinstance(t, tuple):  # control_wrappers.py:337-339: min |arange(lo, hi) - val|
                vals = np.arange(*t)
                if len(vals) == 0:  # the reference raises at its first get_loss (min of an empty array)
                    raise ValueError(f"empty target range {t} for '{k}' on map_shape {map_shape}")
                cfg.trg_lo[i], cfg.trg_hi[i] = float(vals[0]), float(vals[-1])
            else:
                cfg.trg_lo[i] = cfg.trg_hi[i] = float(t)
    cfg.solver_power = solver_power
    controls = list(controls or [])
    cfg.n_ctrl = len(controls)
    bounds = cond_bounds(problem, map_shape)
    for i, k in enumerate(controls):  # control_wrappers.py:66-73
        cfg.ctrl_idx[i] = keys.index(k)
        cfg.ctrl_range[i] = abs(bounds[k][1] - bounds[k][0])
    if act_window is not None:  # reps/wrappers.py:720-722
        assert representation == "narrow" and ndim == 2, "the reference's MultiActionRepresentation only runs on narrow"
        for d in range(ndim):
            cfg.act_window[d] = int(act_window[d])
    # rl/utils.py:308: static_tile_wrapper = static_prob is not None or n_static_walls is not None
    if static_prob is not None or n_static_walls is not None:
        assert representation in ("narrow", "turtle") and ndim == 2, "StaticTileRepresentation: narrow / turtle, 2-D"
        cfg.static_tiles = 1
        cfg.static_prob = float(static_prob or 0)        # reps/wrappers.py:240
        cfg.n_static_walls = int(n_static_walls or 0)    # :242
        cfg.static_eval = int(bool(static_eval))
    return cfg


class OracleVecEnv:
    """Batched CPU env with the same call shape as the HIP engine's VecPcgrlEnv (numpy in/out)."""

    def __init__(self, problem, representation, map_shape, num_envs, seeds=None, threads=1, **kw):
        self.problem, self.representation = problem, representation
        self.map_shape = tuple(map_shape)
        self.cfg = make_config(problem, representation, map_shape, **kw)
        self.n = int(num_envs)
        self.n_cells = int(np.prod(self.map_shape))
        self.n_stats = self.cfg.n_stats
        self.h = lib().orc_create(C.byref(self.cfg), self.n)
        lib().orc_set_threads(self.h, int(threads))
        self.obs_size = int(lib().orc_obs_size(self.h))
        if seeds is not None:
            self.seed(seeds)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    @property
    def obs_shape(self):
        if self.representation == "wide":
            return self.map_shape + (N_TILES[self.problem],)
        ow = tuple(int(self.cfg.obs_window[d]) for d in range(len(self.map_shape)))
        extra = (1 if self.problem == "minecraft_3D_maze" else 0) + (1 if self.cfg.static_tiles else 0)
        return ow + (N_TILES[self.problem] + 1 + extra,)

    def seed(self, seeds):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.n,)))
        lib().orc_seed(self.h, s.ctypes.data)

    def reset(self, mask=None, init_grids=None, init_pos=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        g = None if init_grids is None else np.ascontiguousarray(init_grids, dtype=np.uint8).reshape(self.n, self.n_cells)
        p = None
        if init_pos is not None:
            p = np.zeros((self.n, 3), np.int32)
            ip = np.asarray(init_pos, dtype=np.int32).reshape(self.n, -1)
            p[:, : ip.shape[1]] = ip
        lib().orc_reset(self.h, None if m is None else m.ctypes.data, None if g is None else g.ctypes.data,
                        None if p is None else p.ctypes.data)
        return self.observe()

    def observe(self):
        obs = np.empty((self.n, self.obs_size), np.uint8)
        lib().orc_observe(self.h, obs.ctypes.data)
        return obs.reshape((self.n,) + self.obs_shape)

    def step(self, actions, auto_reset=False, want_obs=True):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        obs = np.empty((self.n, self.obs_size), np.uint8) if want_obs else None
        rew = np.empty(self.n, np.float64)
        done = np.empty(self.n, np.uint8)
        stats = np.empty((self.n, self.n_stats), np.int32)
        lib().orc_step(self.h, a.ctypes.data, int(auto_reset), None if obs is None else obs.ctypes.data,
                       rew.ctypes.data, done.ctypes.data, stats.ctypes.data)
        if obs is not None:
            obs = obs.reshape((self.n,) + self.obs_shape)
        return obs, rew, done.astype(bool), stats

    def step_masked(self, mask, actions, auto_reset=False):
        """step only the envs with mask != 0; returns (obs, reward, done, stats) with the rows of the other envs zero"""
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        a = np.ascontiguousarray(actions, dtype=np.int32)
        obs = np.zeros((self.n, self.obs_size), np.uint8)
        rew = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.uint8)
        stats = np.zeros((self.n, self.n_stats), np.int32)
        lib().orc_step_masked(self.h, m.ctypes.data, a.ctypes.data, int(auto_reset), obs.ctypes.data, rew.ctypes.data,
                              done.ctypes.data, stats.ctypes.data)
        return obs.reshape((self.n,) + self.obs_shape), rew, done.astype(bool), stats

    def update(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        obs = np.empty((self.n, self.obs_size), np.uint8)
        lib().orc_update(self.h, a.ctypes.data, obs.ctypes.data)
        return obs.reshape((self.n,) + self.obs_shape)

    def refresh_stats(self):
        stats = np.empty((self.n, self.n_stats), np.int32)
        lib().orc_refresh_stats(self.h, stats.ctypes.data)
        return stats

    def set_static(self, static_prob=None, n_static_walls=None, eval_mode=False):
        lib().orc_set_static(self.h, -1.0 if static_prob is None else float(static_prob),
                             -1 if n_static_walls is None else int(n_static_walls), int(bool(eval_mode)))

    def static_tiles(self):
        """StaticTileRepresentation.static_tiles, bordered shape [N, H+2, W+2]"""
        h, w = self.map_shape
        out = np.empty((self.n, h + 2, w + 2), np.uint8)
        lib().orc_get_static(self.h, out.ctypes.data)
        return out

    def get_state(self):
        grids = np.empty((self.n, self.n_cells), np.uint8)
        pos = np.empty((self.n, 3), np.int32)
        counters = np.empty((self.n, 4), np.int32)
        stats = np.empty((self.n, self.n_stats), np.int32)
        last_loss = np.empty(self.n, np.float64)
        ep_ret = np.empty(self.n, np.float64)
        lib().orc_get_state(self.h, grids.ctypes.data, pos.ctypes.data, counters.ctypes.data, stats.ctypes.data,
                            last_loss.ctypes.data, ep_ret.ctypes.data)
        return dict(grids=grids, pos=pos, iteration=counters[:, 0], changes=counters[:, 1], n_step=counters[:, 2],
                    ep_len=counters[:, 3], stats=stats, last_loss=last_loss, ep_return=ep_ret)

    def queue_targets(self, trgs, mask=None):
        """set_trgs (control_wrappers.py:168-172): `trgs` = {metric: value or [N] values}; applied at the next reset."""
        keys = STAT_KEYS[self.problem]
        lo = np.zeros((self.n, self.n_stats), np.float64)
        hi = np.zeros((self.n, self.n_stats), np.float64)
        for k, v in trgs.items():
            lo[:, keys.index(k)] = hi[:, keys.index(k)] = np.broadcast_to(np.asarray(v, np.float64), (self.n,))
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_queue_targets(self.h, None if m is None else m.ctypes.data, lo.ctypes.data, hi.ctypes.data)

    def ctrl_obs(self):
        out = np.zeros((self.n, 2 * self.cfg.n_ctrl), np.float64)
        lib().orc_get_ctrl_obs(self.h, out.ctypes.data)
        return out

    def last_episode(self):
        ret = np.empty(self.n, np.float64)
        ln = np.empty(self.n, np.int32)
        fs = np.empty((self.n, self.n_stats), np.int32)
        ne = np.empty(self.n, np.int64)
        lib().orc_get_last_episode(self.h, ret.ctypes.data, ln.ctypes.data, fs.ctypes.data, ne.ctypes.data)
        return dict(ep_return=ret, ep_len=ln, final_stats=fs, n_episodes=ne)


def stats_for_grids(problem, grids, map_shape=None, solver_power=10000, threads=1):
    grids = np.ascontiguousarray(grids, dtype=np.uint8)
    if map_shape is None:
        map_shape = grids.shape[1:]
    cfg = make_config(problem, "narrow", map_shape, solver_power=solver_power)
    n = grids.shape[0]
    out = np.empty((n, cfg.n_stats), np.int32)
    lib().orc_stats_for_grids_mt(C.byref(cfg), n, grids.ctypes.data, out.ctypes.data, int(threads))
    return out


def rng_probe(seed, n):
    st = np.zeros(4, np.uint64)
    d = np.zeros(n, np.float64)
    lib().orc_rng_probe(C.c_uint64(int(seed)), n, st.ctypes.data, d.ctypes.data)
    return st, d
