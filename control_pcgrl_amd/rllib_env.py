"""RLlib-shaped batched adapter: one `VecPcgrlEnv` behind ray.rllib's `VectorEnv` call shape.

The reference trains with RLlib workers that each hold `num_envs_per_worker` gym envs
(control_pcgrl/rl/utils.py:396-415 `ppo_config.rollouts(num_envs_per_worker=...)`, rl/train.py:251
`register_env('pcgrl', make_env)`); RLlib wraps them in a `VectorEnv` and drives it with
`vector_reset / reset_at / vector_step / get_sub_environments`.  `PcgrlVectorEnv` is that object for the
whole worker batch at once: ONE engine, one `pcgrl_step` launch and ONE device->host copy per `vector_step`
(observations, rewards, dones and stats share a packed buffer), numpy views handed to the caller.

    register_env("pcgrl", lambda env_config: PcgrlVectorEnv(env_config, num_envs=env_config["num_envs_per_worker"]))

Episode ends follow RLlib's contract exactly: `vector_step` returns the true last observation of a finished
episode (the engine runs without in-kernel auto-reset here), and RLlib then calls `reset_at(i)` for each
finished env.  All envs that finished in the same step are reset by a single masked `pcgrl_reset` launch on the
first of those calls; the following `reset_at` calls are served from that result.

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
        self.metric_trgs = self.static_trgs
        self.cond_bounds = dict(parent.vec.spec.cond_bounds)
        self.ctrl_metrics = list(parent.vec.controls)

    @property
    def unwrapped(self):
        return self

    @property
    def _rep_stats(self):
        return self._p.stats_dict(self._i)

    metrics = _rep_stats


class _LazyInfos:
    """list-like: info dict i is built when somebody reads it (N dicts per step would cost more than the step)"""

    def __init__(self, parent, n):
        self._p, self._n = parent, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._p.info_dict(i)

    def __iter__(self):
        return (self[i] for i in range(self._n))


class PcgrlVectorEnv(_Base):
    def __init__(self, cfg=None, num_envs=1, device="cuda:0", seeds=None, vec: VecPcgrlEnv = None, obs_dtype=np.float32):
        self.vec = vec if vec is not None else make_vec_env(cfg, num_envs, device=device, seeds=seeds, auto_reset=False)
        v = self.vec
        assert not v.auto_reset, "PcgrlVectorEnv drives resets itself (RLlib calls reset_at)"
        if v.controls:
            raise NotImplementedError("controllable mode: use make_env / VecPcgrlEnv (control planes are per-env scalars)")
        self.num_envs = v.num_envs
        self.obs_dtype = np.dtype(obs_dtype)
        self.observation_space = Box(low=0, high=1, shape=v.obs_shape, dtype=np.float32)  # wrappers.py:121-123
        self.action_space = (MultiDiscrete([v.spec.n_tiles] * v.action_entries) if v.act_window
                             else Discrete(v.num_actions))
        if _Base is not object:  # pragma: no cover
            super().__init__(self.observation_space, self.action_space, self.num_envs)
        N, S = self.num_envs, v.n_stats
        ob = int(np.prod(v.obs_shape))
        # packed output buffer: [obs N*ob | reward N*4 | stats N*S*4 | done N] (16-byte aligned sections), device + pinned host
        def up(x):
            return (x + 15) & ~15
        self._o_obs, self._o_rew = 0, up(N * ob)
        self._o_stats = self._o_rew + up(N * 4)
        self._o_done = self._o_stats + up(N * S * 4)
        total = self._o_done + up(N)
        self._dev = torch.zeros(total, dtype=torch.uint8, device=v.device)
        self._host = torch.zeros(total, dtype=torch.uint8).pin_memory()
        h = self._host.numpy()
        self._obs_u8 = h[self._o_obs:self._o_obs + N * ob].reshape((N,) + v.obs_shape)
        self._rew = h[self._o_rew:self._o_rew + N * 4].view(np.float32)
        self._stats = h[self._o_stats:self._o_stats + N * S * 4].view(np.int32).reshape(N, S)
        self._done = h[self._o_done:self._o_done + N].view(np.bool_)
        self._obs = np.zeros((N,) + v.obs_shape, self.obs_dtype) if self.obs_dtype != np.uint8 else self._obs_u8
        base = self._dev.data_ptr()
        self._ptrs = (base + self._o_obs, base + self._o_rew, base + self._o_done, base + self._o_stats)
        self._act = torch.zeros((N, v.action_entries), dtype=torch.int32).pin_memory()
        self._act_dev = torch.zeros((N, v.action_entries), dtype=torch.int32, device=v.device)
        self._pending = np.zeros(N, np.bool_)   # finished, waiting for RLlib's reset_at
        self._fresh = np.zeros(N, np.bool_)     # already reset by the batched launch, reset_at only hands the obs out
        self._iter = np.zeros(N, np.int64)
        self._subs = [_SubEnv(self, i) for i in range(N)]

    # -- helpers -----------------------------------------------------------------------------------
    def _pull(self):
        self._host.copy_(self._dev, non_blocking=True)  # the one device -> host copy of the call
        torch.cuda.current_stream(self.vec.device).synchronize()
        if self._obs is not self._obs_u8:
            np.copyto(self._obs, self._obs_u8, casting="unsafe")

    def stats_dict(self, i):
        return {k: int(x) for k, x in zip(self.vec.stat_keys, self._stats[i])}

    def info_dict(self, i):
        d = self.stats_dict(i)
        d.update(iterations=int(self._iter[i]), max_iterations=int(self.vec.cfg.max_iterations),
                 max_changes=None if self.vec.cfg.max_changes < 0 else int(self.vec.cfg.max_changes))
        return d

    def _masked_reset(self, mask):
        v = self.vec
        m = torch.as_tensor(mask.astype(np.uint8), device=v.device)
        L, s = v._L, v._stream()
        from . import _lib
        _lib.check(L.pcgrl_reset(v._h, m.data_ptr(), None, None, s), "pcgrl_reset")
        _lib.check(L.pcgrl_observe(v._h, self._ptrs[0], s), "pcgrl_observe")
        _lib.check(L.pcgrl_get_state(v._h, None, None, None, self._ptrs[3], None, None, s), "pcgrl_get_state")
        self._pull()
        self._iter[mask] = 0

    # -- VectorEnv API -----------------------------------------------------------------------------
    def vector_reset(self, *, seeds=None, options=None):
        if seeds is not None and any(s is not None for s in seeds):
            self.vec.seed([0 if s is None else int(s) for s in seeds])
        self._masked_reset(np.ones(self.num_envs, np.bool_))
        self._pending[:] = False
        self._fresh[:] = False
        return list(self._obs), [{} for _ in range(self.num_envs)]

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
        return self._obs[i], {}

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
        rc = v._L.pcgrl_step(v._h, self._act_dev.data_ptr(), 0, self._ptrs[0], self._ptrs[1], self._ptrs[2], self._ptrs[3],
                             v._stream())
        if rc:
            from . import _lib
            _lib.check(rc, "pcgrl_step")
        self._pull()
        self._iter += 1
        self._pending |= self._done
        self._fresh[:] = False
        done = self._done.tolist()
        return list(self._obs), self._rew.tolist(), done, list(done), _LazyInfos(self, self.num_envs)

    def get_sub_environments(self):
        return self._subs

    def try_render_at(self, index=None):
        return None

    def close(self):
        self.vec.close()
