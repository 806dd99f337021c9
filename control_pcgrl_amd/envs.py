"""Single-environment adapter with the reference's gym call shape.

`make_env(cfg)` mirrors control_pcgrl/rl/envs.py:28-81: it returns an object whose
reset() -> (obs, info) and step(a) -> (obs, reward, done, truncated, info) tuples, observation_space /
action_space and the attributes RLlib callbacks read (`unwrapped._rep_stats`, `metrics`, `ctrl_metrics`,
`metric_trgs`, `cond_bounds`, `static_trgs`; control_wrappers.py:56-76, envs/pcgrl_ctrl_env.py:8-10) match the
reference's wrapped env, while the transition itself runs on the GPU engine (a VecPcgrlEnv of size 1, or a slot
of a shared batch).  Intended for plumbing / drop-in checks; throughput comes from make_vec_env().
"""
import numpy as np
import torch

from .vec_env import VecPcgrlEnv, _cfg_get, make_vec_env

try:  # use the real spaces when gymnasium is installed, else shape/bounds holders with the same attributes
    from gymnasium import spaces as _spaces

    Box, Discrete, MultiDiscrete = _spaces.Box, _spaces.Discrete, _spaces.MultiDiscrete
except Exception:  # pragma: no cover - gymnasium is not in the build image

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:  # array bounds, like gymnasium
                shape = np.shape(low)
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Discrete:
        def __init__(self, n):
            self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape, self.dtype = self.nvec.shape, np.dtype(np.int64)

        def sample(self):
            return (np.random.random(self.nvec.shape) * self.nvec).astype(np.int64)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"


class PcgrlGymEnv:
    """One env with the reference's API on top of the batched engine."""

    metadata = {"render.modes": []}

    def __init__(self, cfg=None, vec: VecPcgrlEnv = None, device="cuda:0", seed=None):
        self._vec = vec if vec is not None else make_vec_env(cfg, 1, device=device, auto_reset=False,
                                                              seeds=None if seed is None else [seed])
        assert self._vec.num_envs == 1 and not self._vec.auto_reset
        v = self._vec
        # wrappers.py:121-123 ToImage: Box(low=0, high=max tile value, shape=(H, W, C)) float32
        # controllable mode prepends 2 * len(controls) constant planes (control_wrappers.py:86-104, :189-214)
        self._n_ctrl_planes = 2 * len(v.controls)
        shape = v.obs_shape[:-1] + (v.obs_shape[-1] + self._n_ctrl_planes,)
        # ToImage takes high = max over the stacked spaces (wrappers.py:113-123); the control planes are declared
        # Box(0, 1) by the reference (control_wrappers.py:96-104) although target / range can leave that interval
        if self._n_ctrl_planes:  # control_wrappers.py:96-104: low / high arrays, zeros / ones for the control planes
            self.observation_space = Box(low=np.zeros(shape, np.float32), high=np.ones(shape, np.float32), dtype=np.float32)
        else:
            self.observation_space = Box(low=0, high=1, shape=shape, dtype=np.float32)
        if v.act_window:  # envs/reps/wrappers.py:434-439: one tile id per cell of the action patch
            self.action_space = MultiDiscrete([v.spec.n_tiles] * v.action_entries)
        else:
            self.action_space = Discrete(v.num_actions)  # narrow_rep.py:65-68, turtle_rep.py:70-71, wrappers.py:297
        self.static_trgs = dict(v.spec.static_trgs)
        self.metric_trgs = self.static_trgs
        self.cond_bounds = dict(v.spec.cond_bounds)
        self.ctrl_metrics = list(v.controls)
        self.metric_weights = {k: float(v.cfg.weights[i]) for i, k in enumerate(v.stat_keys)}
        self.metrics = {k: None for k in self.static_trgs}
        self._rep_stats = None
        self.render_mode = None

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        if seed is not None:
            self._vec.seed([int(seed)])
        return [seed]

    def set_trgs(self, trgs):
        """ControlWrapper.set_trgs (control_wrappers.py:168-172): takes effect at the next reset()."""
        self._vec.queue_targets({k: float(x) if not isinstance(x, tuple) else x for k, x in trgs.items()})
        self.metric_trgs.update(trgs)

    def _with_ctrl_planes(self, obs, info):
        o = obs[0].float().cpu().numpy()
        if not self._n_ctrl_planes:
            return o
        c = info["ctrl_obs"][0].cpu().numpy()
        planes = np.broadcast_to(c, o.shape[:-1] + (self._n_ctrl_planes,)).astype(np.float32)
        return np.concatenate((planes, o), axis=-1)  # control planes first (:210)

    def _stats_dict(self, stats_row):
        vals = stats_row.tolist()
        return {k: int(x) for k, x in zip(self._vec.stat_keys, vals)}

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        obs, info = self._vec.reset()
        st = self._vec.get_state()
        self._rep_stats = self._stats_dict(st.stats[0].cpu())
        self.metrics = self._rep_stats
        return self._with_ctrl_planes(obs, info), {}

    def step(self, action):
        if self._vec.act_window:
            a = np.asarray(action, dtype=np.int64).reshape(-1)  # :478 action.reshape(self.action_size)
            if a.size != self._vec.action_entries or (a < 0).any() or (a >= self._vec.spec.n_tiles).any():
                raise IndexError(f"action {a.tolist()} outside {self.action_space}")
            act = torch.tensor(a.reshape(1, -1), dtype=torch.int32, device=self._vec.device)
        else:
            a = int(action)
            if not 0 <= a < self.action_space.n:
                # the reference raises IndexError from numpy indexing on an out-of-range action
                raise IndexError(f"action {a} outside Discrete({self.action_space.n})")
            act = torch.tensor([a], dtype=torch.int32, device=self._vec.device)
        obs, rew, done, trunc, info = self._vec.step(act)
        stats = info["stats"][0].cpu()
        self._rep_stats = self._stats_dict(stats)
        self.metrics = self._rep_stats
        st = self._vec.get_state()
        d = bool(done[0].item())
        out_info = dict(self._rep_stats)
        out_info.update(iterations=int(st.iteration[0]), changes=int(st.changes[0]),
                        max_iterations=int(self._vec.cfg.max_iterations),
                        max_changes=None if self._vec.cfg.max_changes < 0 else int(self._vec.cfg.max_changes))
        return self._with_ctrl_planes(obs, info), float(rew[0].item()), d, d, out_info

    # StaticTileRepresentation setters, reached in the reference as env.unwrapped._rep.set_* (rl/evaluate.py:128-129)
    def set_static_prob(self, static_prob):
        self._vec.set_static(static_prob=static_prob)

    def set_n_static_walls(self, n_static_walls):
        self._vec.set_static(n_static_walls=n_static_walls)

    def set_eval_mode(self, eval_mode):
        self._vec.set_static(eval_mode=eval_mode)

    @property
    def _rep(self):
        return self

    def get_map(self):
        return self._vec.get_state().grids[0].cpu().numpy()

    def close(self):
        self._vec.close()


def make_env(cfg, device="cuda:0"):
    """Drop-in for control_pcgrl/rl/envs.py:make_env(cfg) on this path."""
    rep = _cfg_get(cfg, "representation")
    if rep not in ("narrow", "turtle", "wide"):
        raise Exception("Unknown representation: {}".format(rep))  # rl/envs.py:65
    return PcgrlGymEnv(cfg, device=device)
