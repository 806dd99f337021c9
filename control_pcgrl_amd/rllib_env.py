"""RLlib-shaped batched adapter: one `VecPcgrlEnv` behind ray.rllib's `VectorEnv` call shape.

The reference trains with RLlib workers that each hold `num_envs_per_worker` gym envs
(control_pcgrl/rl/utils.py:396-415 `ppo_config.rollouts(num_envs_per_worker=...)`, rl/train.py:251
`register_env('pcgrl', make_env)`); RLlib wraps them in a `VectorEnv` and drives it with
`vector_reset / reset_at / vector_step / get_sub_environments`.  `PcgrlVectorEnv` is that object for the
whole worker batch at once: ONE engine, one `pcgrl_step` launch and ONE device->host copy per `vector_step`
(observations, rewards, dones and stats share a packed buffer).  Every call hands out FRESH numpy arrays: RLlib's
collectors keep references to the observations they are given and stack them later, so nothing a call returned is ever
written again.

    register_env("pcgrl", lambda env_config: PcgrlVectorEnv(env_config, num_envs=env_config["num_envs_per_worker"]))

Episode ends follow RLlib's contract exactly: `vector_step` returns the true last observation of a finished
episode (the engine runs without in-kernel auto-reset here), and RLlib then calls `reset_at(i)` for each
finished env.  All envs that finished in the same step are reset by a single masked `pcgrl_reset` launch on the
first of those calls; the following `reset_at` calls are served from that result (kept in buffers of its own: the
arrays `vector_step` returned stay what they were).

Controllable generation (`cfg.controls`, control_wrappers.py:86-104, :189-214): the observation carries the
2 * len(controls) constant planes (target / range, metric / range) in front of the one-hot channels, like the
reference's ControlWrapper; targets are set per sub-env with `get_sub_environments()[i].set_trgs({...})` (or for the
whole batch with `set_trgs`) and take effect at that env's next reset (:168-178).

ray is not required: with ray installed the class derives from `ray.rllib.env.vector_env.VectorEnv`, otherwise
it is a plain object with the same methods (which is how the tests drive it).
"""
import numpy as np
import torch

from .envs import Box, Discrete, MultiDiscrete
from .vec_env import VecPcgrlEnv, make_vec_env

try:  # pragma: no cover - ray is not in the build image
    from ray.rllib.env.vector_env import VectorEnv as _Base
except Exception:
    _Base = object


class _SubEnv:
    """What RLlib callbacks reach through `base_env.get_sub_environments()[i]` (rl/callbacks.py:91-117 reads
    `env.unwrapped._rep_stats`, `metrics`, `ctrl_metrics`, `static_trgs`)."""

    def __init__(self, parent, index):
        self._p, self._i = parent, index
        self.observation_space, self.action_space = parent.observation_space, parent.action_space
        self.static_trgs = dict(parent.vec.spec.static_trgs)
        self.metric_trgs = dict(self.static_trgs)
        self.cond_bounds = dict(parent.vec.spec.cond_bounds)
        self.ctrl_metrics = list(parent.vec.controls)

    @property
    def unwrapped(self):
        return self

    @property
    def _rep_stats(self):
        return self._p.stats_dict(self._i)

    metrics = _rep_stats

    def set_trgs(self, trgs):
        """ControlWrapper.set_trgs (control_wrappers.py:168-172) for this env: applied at its next reset"""
        self._p.set_trgs(trgs, index=self._i)
        self.metric_trgs.update(trgs)


class _Infos:
    """list-like over a SNAPSHOT of the step's statistics: info dict i is built when somebody reads it (N dicts per step
    would cost more than the step), from values no later call can change"""

    def __init__(self, parent, stats, iters):
        self._p, self._stats, self._iters, self._n = parent, stats, iters, len(iters)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._p._info(self._stats[i], self._iters[i])

    def __iter__(self):
        return (self[i] for i in range(self._n))


class PcgrlVectorEnv(_Base):
    def __init__(self, cfg=None, num_envs=1, device="cuda:0", seeds=None, vec: VecPcgrlEnv = None, obs_dtype=np.float32):
        self.vec = vec if vec is not None else make_vec_env(cfg, num_envs, device=device, seeds=seeds, auto_reset=False)
        v = self.vec
        assert not v.auto_reset, "PcgrlVectorEnv drives resets itself (RLlib calls reset_at)"
        self.num_envs = v.num_envs
        self.n_ctrl_planes = 2 * len(v.controls)
        self.obs_dtype = np.dtype(obs_dtype if not self.n_ctrl_planes else np.float32)  # (control planes are fractions)
        shape = v.obs_shape[:-1] + (v.obs_shape[-1] + self.n_ctrl_planes,)
        # wrappers.py:113-123 ToImage: Box(0, max over the stacked spaces) = Box(0, 1) for the one-hot map; the control
        # planes are declared [0, 1] by the reference too (control_wrappers.py:96-104) although metric / range may exceed it
        if self.n_ctrl_planes:
            self.observation_space = Box(low=np.zeros(shape, np.float32), high=np.ones(shape, np.float32), dtype=np.float32)
        else:
            self.observation_space = Box(low=0, high=1, shape=shape, dtype=np.float32)
        self.action_space = (MultiDiscrete([v.spec.n_tiles] * v.action_entries) if v.act_window
                             else Discrete(v.num_actions))
        if _Base is not object:  # pragma: no cover
            super().__init__(self.observation_space, self.action_space, self.num_envs)
        N, S = self.num_envs, v.n_stats
        ob = int(np.prod(v.obs_shape))
        self._ob_shape = v.obs_shape

        # packed output buffers: [obs N*ob | reward N*4 | stats N*S*4 | done N | ctrl N*2K*4] (16-byte aligned sections),
        # device + pinned host; one set for vector_step, one for the batched reset
        def up(x):
            return (x + 15) & ~15

        o_rew = up(N * ob)
        o_stats = o_rew + up(N * 4)
        o_done = o_stats + up(N * S * 4)
        o_ctrl = o_done + up(N)
        total = o_ctrl + up(N * max(self.n_ctrl_planes, 1) * 4)

        def buffers():
            dev = torch.zeros(total, dtype=torch.uint8, device=v.device)
            host = torch.zeros(total, dtype=torch.uint8).pin_memory()
            h = host.numpy()
            views = dict(obs=h[0:N * ob].reshape((N,) + v.obs_shape), rew=h[o_rew:o_rew + N * 4].view(np.float32),
                         stats=h[o_stats:o_stats + N * S * 4].view(np.int32).reshape(N, S), done=h[o_done:o_done + N].view(np.bool_),
                         ctrl=h[o_ctrl:o_ctrl + N * max(self.n_ctrl_planes, 1) * 4].view(np.float32).reshape(N, -1))
            base = dev.data_ptr()
            ptrs = dict(obs=base, rew=base + o_rew, stats=base + o_stats, done=base + o_done, ctrl=base + o_ctrl)
            return dev, host, views, ptrs

        self._sdev, self._shost, self._s, self._sp = buffers()  # step
        self._rdev, self._rhost, self._r, self._rp = buffers()  # reset
        self._act = torch.zeros((N, v.action_entries), dtype=torch.int32).pin_memory()
        self._act_dev = torch.zeros((N, v.action_entries), dtype=torch.int32, device=v.device)
        self._stats = np.zeros((N, S), np.int32)  # the envs' current statistics (sub-env accessors)
        self._pending = np.zeros(N, np.bool_)   # finished, waiting for RLlib's reset_at
        self._fresh = np.zeros(N, np.bool_)     # already reset by the batched launch, reset_at only hands the obs out
        self._reset_obs = None                  # observations of the last batched reset (fresh array per reset)
        self._iter = np.zeros(N, np.int64)
        self._subs = [_SubEnv(self, i) for i in range(N)]

    # -- helpers -----------------------------------------------------------------------------------
    def _observations(self, views):
        """a NEW array [N, H, W, C (+ control planes)] from the pinned staging buffer"""
        o = views["obs"].astype(self.obs_dtype)
        if not self.n_ctrl_planes:
            return o
        planes = np.broadcast_to(views["ctrl"][:, :self.n_ctrl_planes].reshape((self.num_envs,) + (1,) * (o.ndim - 2) + (-1,)),
                                 o.shape[:-1] + (self.n_ctrl_planes,))
        return np.concatenate((planes.astype(self.obs_dtype), o), axis=-1)  # control planes first (:210)

    def stats_dict(self, i):
        return {k: int(x) for k, x in zip(self.vec.stat_keys, self._stats[i])}

    def _info(self, stats_row, it):
        d = {k: int(x) for k, x in zip(self.vec.stat_keys, stats_row)}
        d.update(iterations=int(it), max_iterations=int(self.vec.cfg.max_iterations),
                 max_changes=None if self.vec.cfg.max_changes < 0 else int(self.vec.cfg.max_changes))
        return d

    def info_dict(self, i):
        return self._info(self._stats[i], self._iter[i])

    def set_trgs(self, trgs, index=None):
        """ControlWrapper.set_trgs for one env (index) or the whole batch: queued, applied at the next reset"""
        mask = None
        if index is not None:
            mask = np.zeros(self.num_envs, np.uint8)
            mask[int(index)] = 1
        self.vec.queue_targets({k: (x if isinstance(x, tuple) else float(x)) for k, x in trgs.items()}, mask=mask)

    def _masked_reset(self, mask):
        """one launch for every env in `mask`; results go to the reset staging buffers and, for the masked envs only, into
        the current statistics"""
        v = self.vec
        m = torch.as_tensor(mask.astype(np.uint8), device=v.device)
        L, s = v._L, v._stream()
        from . import _lib
        _lib.check(L.pcgrl_reset(v._h, m.data_ptr(), None, None, s), "pcgrl_reset")
        _lib.check(L.pcgrl_observe(v._h, self._rp["obs"], s), "pcgrl_observe")
        _lib.check(L.pcgrl_get_state(v._h, None, None, None, self._rp["stats"], None, None, s), "pcgrl_get_state")
        if self.n_ctrl_planes:
            _lib.check(L.pcgrl_ctrl_observe(v._h, self._rp["ctrl"], s), "pcgrl_ctrl_observe")
        self._rhost.copy_(self._rdev, non_blocking=True)
        torch.cuda.current_stream(v.device).synchronize()
        self._reset_obs = self._observations(self._r)
        self._stats[mask] = self._r["stats"][mask]
        self._iter[mask] = 0

    # -- VectorEnv API -----------------------------------------------------------------------------
    def vector_reset(self, *, seeds=None, options=None):
        if seeds is not None and any(s is not None for s in seeds):
            self.vec.seed([0 if s is None else int(s) for s in seeds])
        self._masked_reset(np.ones(self.num_envs, np.bool_))
        self._pending[:] = False
        self._fresh[:] = False
        return list(self._reset_obs), [{} for _ in range(self.num_envs)]

    def reset_at(self, index=None, *, seed=None, options=None):
        i = 0 if index is None else int(index)
        if seed is not None:
            raise NotImplementedError("per-env reseeding goes through vector_reset(seeds=...) / VecPcgrlEnv.seed")
        if not self._fresh[i]:
            mask = self._pending.copy()
            mask[i] = True
            self._masked_reset(mask)  # every env that finished in the last step, in one launch
            self._fresh |= mask
            self._pending[:] = False
        self._fresh[i] = False
        return self._reset_obs[i], {}

    def restart_at(self, index=None):
        return self.reset_at(index)[0]

    def vector_step(self, actions):
        v = self.vec
        a = np.asarray(actions, dtype=np.int64).reshape(self.num_envs, v.action_entries)
        hi = v.spec.n_tiles if v.act_window else v.num_actions
        if (a < 0).any() or (a >= hi).any():
            raise IndexError("action outside the action space")  # the reference raises IndexError from numpy indexing
        self._act.numpy()[...] = a
        self._act_dev.copy_(self._act, non_blocking=True)
        p = self._sp
        if self.n_ctrl_planes:
            rc = v._L.pcgrl_step_ex(v._h, self._act_dev.data_ptr(), 0, p["obs"], p["rew"], None, p["done"], p["stats"], p["ctrl"],
                                    v._stream())
        else:
            rc = v._L.pcgrl_step(v._h, self._act_dev.data_ptr(), 0, p["obs"], p["rew"], p["done"], p["stats"], v._stream())
        if rc:
            from . import _lib
            _lib.check(rc, "pcgrl_step")
        self._shost.copy_(self._sdev, non_blocking=True)  # the one device -> host copy of the call
        torch.cuda.current_stream(v.device).synchronize()
        obs = self._observations(self._s)
        self._stats[...] = self._s["stats"]
        self._iter += 1
        self._pending |= self._s["done"]
        self._fresh[:] = False
        done = self._s["done"].tolist()
        return list(obs), self._s["rew"].tolist(), done, list(done), _Infos(self, self._stats.copy(), self._iter.copy())

    def get_sub_environments(self):
        return self._subs

    def try_render_at(self, index=None):
        return None

    def close(self):
        self.vec.close()
