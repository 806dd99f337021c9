"""RLlib-shaped batched adapter: one `VecPcgrlEnv` behind ray.rllib's `VectorEnv` call shape.

The reference trains with RLlib workers that each hold `num_envs_per_worker` gym envs
(control_pcgrl/rl/utils.py:396-415 `ppo_config.rollouts(num_envs_per_worker=...)`, rl/train.py:251
`register_env('pcgrl', make_env)`); RLlib wraps them in a `VectorEnv` and drives it with
`vector_reset / reset_at / vector_step / get_sub_environments`.  `PcgrlVectorEnv` is that object for the
whole worker batch at once: ONE engine, one `pcgrl_step` launch and ONE device->host copy per `vector_step`
(observations, rewards, dones and stats share a packed buffer).  Nothing a call returned is ever written again: RLlib's
collectors keep references to the observations they are given and stack them later.  The arrays of a call live in a
pinned host block of their own, which the device->host copy fills directly (no staging copy, no conversion on the host:
`obs_dtype=np.float32` converts on the device, `obs_dtype=np.uint8` hands out the engine's bytes as they are); the
block goes back to a free list when the last array over it has been garbage-collected, so a steady-state loop
allocates nothing.  Small batches (the reference's 20 envs per worker) skip the copy as well: the step kernel writes
its outputs straight into the pinned block over PCIe (`direct_host_outputs`).

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
import ctypes as C

import numpy as np
import torch

from . import _lib
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
        """ControlWrapper.set_trgs (control_wrappers.py:168-172) for this env: queued, and like the reference's
        _ctrl_trg_queue (:174-178) `metric_trgs` only changes when the env's next reset applies them"""
        self._p.set_trgs(trgs, index=self._i)


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


class _Lease:
    """One hand-out of a pinned host block.  numpy arrays made over it (np.asarray(lease) and every view of that) keep
    the lease alive; when the last of them is gone the block returns to the free list it came from."""
    __slots__ = ("__array_interface__", "block", "ptr", "_free", "__weakref__")

    def __init__(self, block, free):
        self.block, self._free, self.ptr = block, free, block.data_ptr()
        self.__array_interface__ = {"data": (self.ptr, False), "shape": (block.numel(),), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._free.append(self.block)
        except Exception:  # interpreter shutdown
            pass


def _hip_runtime():
    """Device->host copy and stream wait of the HIP runtime libpcgrl_amd.so itself is linked against -- the one that owns
    the stream handles torch hands over (pcgrl_copy_to_host / pcgrl_stream_synchronize; a ctypes call costs ~1 us, the
    torch wrappers ~10).  Never a dlopen of libamdhip64 by name: that could load a second runtime."""
    return _lib.lib()


class PcgrlVectorEnv(_Base):
    DIRECT_MAX_BYTES = 1 << 20  # default: the step kernel writes into host memory itself when a call's outputs are <= 1 MiB
    HOST_CONVERT_MAX_BYTES = 1 << 18  # float32 hand-out: uint8 -> float32 on the host up to this many observation bytes per call

    def __init__(self, cfg=None, num_envs=1, device="cuda:0", seeds=None, vec: VecPcgrlEnv = None, obs_dtype=np.float32,
                 direct_host_outputs=None):
        self.vec = vec if vec is not None else make_vec_env(cfg, num_envs, device=device, seeds=seeds, auto_reset=False)
        v = self.vec
        assert not v.auto_reset, "PcgrlVectorEnv drives resets itself (RLlib calls reset_at)"
        self.num_envs = v.num_envs
        self.n_ctrl_planes = 2 * len(v.controls)
        self.obs_dtype = np.dtype(obs_dtype if not self.n_ctrl_planes else np.float32)  # (control planes are fractions)
        if self.obs_dtype not in (np.dtype(np.float32), np.dtype(np.uint8)):
            raise ValueError("obs_dtype must be float32 (the reference's declared Box dtype) or uint8 (the engine's bytes)")
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
        self._out_shape = (N,) + shape
        ob_out = int(np.prod(shape)) * self.obs_dtype.itemsize  # bytes per env as handed out

        # one packed block per call: [obs | reward N*4 | stats N*S*4 | done N | ctrl N*2K*4], 16-byte aligned sections
        def up(x):
            return (x + 15) & ~15

        self._o_rew = up(N * ob_out)
        self._o_stats = self._o_rew + up(N * 4)
        self._o_done = self._o_stats + up(N * S * 4)
        self._o_ctrl = self._o_done + up(N)
        self._total = self._o_ctrl + up(N * max(self.n_ctrl_planes, 1) * 4)
        self._free = []  # pinned host blocks no array refers to any more
        self._dev = torch.zeros(self._total, dtype=torch.uint8, device=v.device)
        base = self._dev.data_ptr()
        self._dp = dict(obs=base, rew=base + self._o_rew, stats=base + self._o_stats, done=base + self._o_done, ctrl=base + self._o_ctrl)
        self._convert = self.obs_dtype != np.dtype(np.uint8)
        # small batches (the reference's 20 envs per worker): the kernel writes its uint8 outputs into a pinned staging block
        # and the float32 arrays are made on the host -- a device-side conversion launch costs more than converting 60 KB
        self._host_convert = (self._convert and not self.n_ctrl_planes and direct_host_outputs is not False
                              and N * int(np.prod(v.obs_shape)) <= self.HOST_CONVERT_MAX_BYTES)
        if self._host_convert:
            self._o8 = (up(N * int(np.prod(v.obs_shape))), up(N * 4), up(N * S * 4), up(N))  # section sizes of the staging block
            self._stage = torch.zeros(sum(self._o8), dtype=torch.uint8).pin_memory()
            sb, sn = self._stage.data_ptr(), self._stage.numpy()
            o1, o2, o3 = self._o8[0], self._o8[0] + self._o8[1], self._o8[0] + self._o8[1] + self._o8[2]
            self._stage_ptrs = (sb, sb + o1, sb + o2, sb + o3, 0)
            self._stage_views = dict(obs=sn[:N * int(np.prod(v.obs_shape))].reshape((N,) + v.obs_shape), rew=sn[o1:o1 + N * 4].view(np.float32),
                                     stats=sn[o2:o2 + N * S * 4].view(np.int32).reshape(N, S), done=sn[o3:o3 + N].view(np.bool_))
        if self._convert and not self._host_convert:  # the engine's uint8 observation -> float32 (+ control planes in front) on the device
            self._obs_u8 = torch.empty((N,) + v.obs_shape, dtype=torch.uint8, device=v.device)
            self._obs_out = self._dev[:N * ob_out].view(torch.float32).view(self._out_shape)
            self._ctrl_dev = self._dev[self._o_ctrl:self._o_ctrl + N * max(self.n_ctrl_planes, 1) * 4].view(torch.float32).view(N, -1)
            self._engine_obs = self._obs_u8.data_ptr()
        else:
            self._engine_obs = self._dp["obs"]
        if direct_host_outputs is None:
            direct_host_outputs = self._total <= self.DIRECT_MAX_BYTES
        self._direct = (bool(direct_host_outputs) and not self._convert) or self._host_convert
        self._hip = _hip_runtime()
        self._act = torch.zeros((N, v.action_entries), dtype=torch.int32).pin_memory()
        self._act_np = self._act.numpy()
        # small batches: the kernel reads the actions from pinned host memory itself (no host->device copy to issue)
        self._act_dev = self._act if self._direct else torch.zeros((N, v.action_entries), dtype=torch.int32, device=v.device)
        self._stats = np.zeros((N, S), np.int32)  # the envs' current statistics (sub-env accessors)
        self._pending = np.zeros(N, np.bool_)   # finished, waiting for RLlib's reset_at
        self._fresh = np.zeros(N, np.bool_)     # already reset by the batched launch, reset_at only hands the obs out
        self._reset_obs = None                  # observations of the last batched reset (a block of their own)
        self._iter = np.zeros(N, np.int64)
        self._subs = [_SubEnv(self, i) for i in range(N)]
        self._queued_trgs = {}  # env index -> targets set since its last reset

    # -- helpers -----------------------------------------------------------------------------------
    def _lease(self):
        return _Lease(self._free.pop() if self._free else torch.empty(self._total, dtype=torch.uint8).pin_memory(), self._free)

    def _views(self, lease):
        """numpy views of a call's block (they keep the lease, and so the block, alive)"""
        h = np.asarray(lease)
        N, S = self.num_envs, self.vec.n_stats
        return dict(obs=h[:N * int(np.prod(self._out_shape[1:])) * self.obs_dtype.itemsize].view(self.obs_dtype).reshape(self._out_shape),
                    rew=h[self._o_rew:self._o_rew + N * 4].view(np.float32),
                    stats=h[self._o_stats:self._o_stats + N * S * 4].view(np.int32).reshape(N, S),
                    done=h[self._o_done:self._o_done + N].view(np.bool_))

    def _finish(self, lease, stream):
        """device-side conversion if any, the call's one device->host copy (unless the kernels wrote into the block
        themselves), and the wait for it"""
        if self._host_convert:
            rc = self._hip.pcgrl_stream_synchronize(stream)
            if rc:
                _lib.check(rc, "pcgrl_stream_synchronize")
            r, sv = self._views(lease), self._stage_views
            np.copyto(r["obs"], sv["obs"], casting="unsafe")
            r["rew"][...] = sv["rew"]
            r["stats"][...] = sv["stats"]
            r["done"][...] = sv["done"]
            return
        if self._convert:
            K2 = self.n_ctrl_planes
            if K2:
                self._obs_out[..., :K2] = self._ctrl_dev[:, :K2].reshape((self.num_envs,) + (1,) * (len(self._out_shape) - 2) + (K2,))
                self._obs_out[..., K2:] = self._obs_u8  # control planes first (control_wrappers.py:210)
            else:
                self._obs_out.copy_(self._obs_u8)
        if not self._direct:
            rc = self._hip.pcgrl_copy_to_host(lease.ptr, self._dev.data_ptr(), self._total, stream)
            if rc:
                _lib.check(rc, "pcgrl_copy_to_host")
        rc = self._hip.pcgrl_stream_synchronize(stream)
        if rc:
            _lib.check(rc, "pcgrl_stream_synchronize")

    def _out_ptrs(self, lease):
        if self._host_convert:
            return self._stage_ptrs
        if not self._direct:
            return self._engine_obs, self._dp["rew"], self._dp["stats"], self._dp["done"], self._dp["ctrl"]
        b = lease.ptr
        return b, b + self._o_rew, b + self._o_stats, b + self._o_done, b + self._o_ctrl

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
        for i in (range(self.num_envs) if index is None else (int(index),)):
            self._queued_trgs.setdefault(i, {}).update(trgs)

    def _masked_reset(self, mask):
        """one launch for every env in `mask`; the observations go to a block of their own and, for the masked envs only,
        the statistics into the current ones"""
        v = self.vec
        m = torch.as_tensor(mask.astype(np.uint8), device=v.device)
        L, s = v._L, v._stream()
        from . import _lib
        lease = self._lease()
        obs_p, _, stats_p, _, ctrl_p = self._out_ptrs(lease)
        _lib.check(L.pcgrl_reset(v._h, m.data_ptr(), None, None, s), "pcgrl_reset")
        _lib.check(L.pcgrl_observe(v._h, obs_p, s), "pcgrl_observe")
        _lib.check(L.pcgrl_get_state(v._h, None, None, None, stats_p, None, None, s), "pcgrl_get_state")
        if self.n_ctrl_planes:
            _lib.check(L.pcgrl_ctrl_observe(v._h, ctrl_p, s), "pcgrl_ctrl_observe")
        self._finish(lease, s)
        r = self._views(lease)
        self._reset_obs = r["obs"]
        self._stats[mask] = r["stats"][mask]
        self._iter[mask] = 0
        for i in [i for i in self._queued_trgs if mask[i]]:  # the queued targets are the envs' targets from here on
            self._subs[i].metric_trgs.update(self._queued_trgs.pop(i))

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
        if a.min() < 0 or a.max() >= hi:
            raise IndexError("action outside the action space")  # the reference raises IndexError from numpy indexing
        self._act_np[...] = a
        s = v._stream()
        if not self._direct:
            self._act_dev.copy_(self._act, non_blocking=True)
        lease = self._lease()
        obs_p, rew_p, stats_p, done_p, ctrl_p = self._out_ptrs(lease)
        if self.n_ctrl_planes:
            rc = v._L.pcgrl_step_ex(v._h, self._act_dev.data_ptr(), 0, obs_p, rew_p, None, done_p, stats_p, ctrl_p, s)
        else:
            rc = v._L.pcgrl_step(v._h, self._act_dev.data_ptr(), 0, obs_p, rew_p, done_p, stats_p, s)
        if rc:
            from . import _lib
            _lib.check(rc, "pcgrl_step")
        self._finish(lease, s)  # the one device -> host copy of the call
        r = self._views(lease)
        self._stats[...] = r["stats"]
        self._iter += 1
        self._pending |= r["done"]
        self._fresh[:] = False
        done = r["done"].tolist()
        return list(r["obs"]), r["rew"].tolist(), done, list(done), _Infos(self, r["stats"], self._iter.copy())

    def get_sub_environments(self):
        return self._subs

    def try_render_at(self, index=None):
        return None

    def close(self):
        self.vec.close()
