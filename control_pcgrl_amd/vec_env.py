"""Batched PCGRL environment on one MI355X: N envs advanced by one HIP launch per step.

Mirrors, for a batch, the reference's  make_env(cfg) -> ControlWrapper(CroppedImage|ActionMapImage
PCGRLWrapper(PcgrlCtrlEnv))  stack (control_pcgrl/rl/envs.py:28-81): same observation layout
(channel-last one-hot, uint8), same reward, same done rule, same per-problem stats.  All tensors live on the
env's GPU; step() performs no host synchronisation.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .problems import PROBLEMS, REPRESENTATIONS, problem_spec, target_interval


def _cfg_get(cfg, path, default=None):
    cur = cfg
    for part in path.split("."):
        if cur is None:
            return default
        cur = cur.get(part, default) if isinstance(cur, dict) else getattr(cur, part, default)
    return cur


def build_config(problem, representation, map_shape, obs_window=None, weights=None, max_board_scans=3,
                 change_percentage=None, solver_power=10000, static_trgs=None, controls=None, act_window=None,
                 static_prob=None, n_static_walls=None, static_eval=False):
    """cfg fields -> pcgrl_config (include/pcgrl_amd.h)."""
    if representation not in REPRESENTATIONS:
        raise ValueError(f"Unknown representation: {representation}")  # rl/envs.py:65
    spec = problem_spec(problem, map_shape)
    map_shape = tuple(int(s) for s in map_shape)
    ndim = len(map_shape)
    if obs_window is None:
        # rl/utils.py:302-334 validate_config: default obs_window = 2 * map_shape; wide must see the whole map
        obs_window = map_shape if representation == "wide" else tuple(2 * s for s in map_shape)
    obs_window = tuple(int(s) for s in obs_window)
    weights = dict(spec.default_weights if weights is None else weights)
    trgs = dict(spec.static_trgs)
    if static_trgs:
        trgs.update(static_trgs)
    c = _lib.PcgrlConfig()
    c.problem = PROBLEMS[problem]
    c.representation = REPRESENTATIONS[representation]
    c.ndim = ndim
    for d in range(3):
        c.dims[d] = map_shape[d] if d < ndim else 1
        c.obs_window[d] = obs_window[d] if d < ndim else 1
    n_cells = int(np.prod(map_shape))
    c.max_iterations = n_cells * int(max_board_scans) + 1  # envs/pcgrl_env.py:241
    if change_percentage is None:
        c.max_changes = -1
    else:
        assert 0 < change_percentage
        c.max_changes = max(int(change_percentage * n_cells), 1)  # envs/pcgrl_env.py:235-239
    c.n_stats = len(spec.stat_keys)
    for i, k in enumerate(spec.stat_keys):
        c.weights[i] = float(weights.get(k, 0.0))  # control_wrappers.py:41-45
        if k in trgs:  # control_wrappers.py:48-82: all_metrics = static targets when not controllable
            c.has_trg[i] = 1
            c.trg_lo[i], c.trg_hi[i] = target_interval(trgs[k])
    c.solver_power = int(solver_power)
    controls = list(controls or [])
    c.n_ctrl = len(controls)
    for i, k in enumerate(controls):  # control_wrappers.py:66-73: param_ranges[k] = |bounds[1] - bounds[0]|
        if k not in spec.stat_keys or k not in spec.cond_bounds:
            raise ValueError(f"control metric '{k}' is not a metric of problem '{problem}'")
        c.ctrl_idx[i] = spec.stat_keys.index(k)
        c.ctrl_range[i] = abs(spec.cond_bounds[k][1] - spec.cond_bounds[k][0])
    if act_window is not None:  # envs/reps/wrappers.py:720-722 MultiActionRepresentation
        if len(act_window) != ndim or ndim != 2:
            raise ValueError("act_window must have one entry per map dimension (2-D problems)")
        for d in range(ndim):
            c.act_window[d] = int(act_window[d])
    # rl/utils.py:308: static_tile_wrapper = static_prob is not None or n_static_walls is not None
    if static_prob is not None or n_static_walls is not None:
        c.static_tiles = 1
        c.static_prob = float(static_prob or 0)      # envs/reps/wrappers.py:240
        c.n_static_walls = int(n_static_walls or 0)  # :242
        c.static_eval = int(bool(static_eval))
    return c, spec, obs_window


class VecPcgrlEnv:
    """N independent PCGRL envs on one GPU.

    step(actions) -> (obs uint8 [N, *obs_shape], reward f32 [N], done bool [N], truncated bool [N], info)
      info["stats"]  int32 [N, n_stats] in `self.stat_keys` order
    Output tensors are owned by the env and overwritten by the next step()/reset() (clone to keep).
    auto_reset=True (default): finished envs restart inside the same launch (RLlib's convention: the returned
    observation is the first of the new episode); last_episode() exposes what RLlib's callbacks read at
    episode end (rl/callbacks.py:91-117).
    """

    def __init__(self, problem, representation, map_shape, num_envs, device="cuda:0", obs_window=None, weights=None,
                 max_board_scans=3, change_percentage=None, seeds=None, auto_reset=True, solver_power=10000,
                 static_trgs=None, controls=None, reward_dtype=torch.float32, act_window=None, static_prob=None,
                 n_static_walls=None, static_eval=False, _out=None):
        if not torch.cuda.is_available():
            raise RuntimeError("VecPcgrlEnv needs a GPU (ROCm device); there is no CPU fallback in the product path")
        self.device = torch.device(device)
        self.problem, self.representation = problem, representation
        self.map_shape = tuple(int(s) for s in map_shape)
        self.num_envs = int(num_envs)
        self.auto_reset = bool(auto_reset)
        self.cfg, self.spec, self.obs_window = build_config(
            problem, representation, map_shape, obs_window, weights, max_board_scans, change_percentage, solver_power,
            static_trgs, controls, act_window, static_prob, n_static_walls, static_eval)
        self.controls = list(controls or [])
        self.act_window = None if act_window is None else tuple(int(a) for a in act_window)
        self.static_tiles = bool(self.cfg.static_tiles)
        self.stat_keys = list(self.spec.stat_keys)
        self.n_stats = len(self.stat_keys)
        self.n_cells = int(np.prod(self.map_shape))
        L = _lib.lib()
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(L.pcgrl_create(C.byref(self.cfg), self.num_envs, dev_index, C.byref(h)), "pcgrl_create")
        self._h = h
        self._L = L
        self._dev_index = dev_index
        shape = (C.c_int32 * 4)()
        nd = C.c_int32()
        _lib.check(L.pcgrl_obs_shape(h, C.byref(shape), C.byref(nd)), "pcgrl_obs_shape")
        self.obs_shape = tuple(shape[i] for i in range(nd.value))
        n_act = {"narrow": self.spec.n_tiles, "turtle": self.spec.n_tiles + 4,
                 "wide": self.n_cells * self.spec.n_tiles}[representation]
        self.num_actions = n_act
        # with an action patch the action is MultiDiscrete([n_tiles] * prod(act_window)): int32 [N, action_entries]
        self.action_entries = int(np.prod(self.act_window)) if self.act_window else 1
        N, dev = self.num_envs, self.device
        if _out is not None:  # SubBatchedVecEnv: this engine writes its rows of the whole batch's output tensors
            self._obs, self._reward, self._done, self._stats = _out
            assert self._obs.shape == (N,) + self.obs_shape and self._obs.is_contiguous() and self._stats.is_contiguous()
        else:
            self._obs = torch.empty((N,) + self.obs_shape, dtype=torch.uint8, device=dev)
            self._reward = torch.empty(N, dtype=torch.float32, device=dev)
            self._done = torch.empty(N, dtype=torch.uint8, device=dev)
            self._stats = torch.empty((N, self.n_stats), dtype=torch.int32, device=dev)
        self._ptrs = (self._obs.data_ptr(), self._reward.data_ptr(), self._done.data_ptr(), self._stats.data_ptr())
        # controllable mode / float64 rewards go through pcgrl_step_ex
        self._reward64 = torch.empty(N, dtype=torch.float64, device=dev) if reward_dtype == torch.float64 else None
        self._ctrl_obs = torch.zeros((N, 2 * len(self.controls)), dtype=torch.float32, device=dev) if self.controls else None
        self._ex = self._reward64 is not None or self._ctrl_obs is not None
        # step() hands out the same tensors every call (they are overwritten in place): the tuple is built once
        done = self._done.view(torch.bool)
        info = {"stats": self._stats}
        if self._ctrl_obs is not None:
            info["ctrl_obs"] = self._ctrl_obs
        self._step_out = (self._obs, self._reward64 if self._reward64 is not None else self._reward, done, done, info)
        if seeds is not None:
            self.seed(seeds)

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.pcgrl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        """raw hipStream_t of torch's current stream on the env's device (the fast accessor when this torch has it)"""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index)
        except AttributeError:  # pragma: no cover
            return torch.cuda.current_stream(self.device).cuda_stream

    def seed(self, seeds):
        """Env i gets numpy PCG64(SeedSequence(seeds[i])) for both RNG streams (envs/pcgrl_env.py:142-146)."""
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.num_envs,)))
        _lib.check(self._L.pcgrl_seed(self._h, s.ctypes.data), "pcgrl_seed")

    # -- gym-like API ------------------------------------------------------------------------------
    def reset(self, mask=None, init_grids=None, init_pos=None):
        def dev(t, dtype):
            if t is None:
                return None
            return torch.as_tensor(t, device=self.device).to(dtype).contiguous()

        m = dev(mask, torch.uint8)
        g = dev(init_grids, torch.uint8)
        p = None
        if init_pos is not None:
            ip = torch.as_tensor(init_pos, device=self.device).to(torch.int32).reshape(self.num_envs, -1)
            p = torch.zeros((self.num_envs, 3), dtype=torch.int32, device=self.device)
            p[:, : ip.shape[1]] = ip
        if g is not None:
            assert g.numel() == self.num_envs * self.n_cells
        _lib.check(self._L.pcgrl_reset(self._h, m.data_ptr() if m is not None else None,
                                       g.data_ptr() if g is not None else None,
                                       p.data_ptr() if p is not None else None, self._stream()), "pcgrl_reset")
        _lib.check(self._L.pcgrl_observe(self._h, self._ptrs[0], self._stream()), "pcgrl_observe")
        if self._ctrl_obs is not None:
            _lib.check(self._L.pcgrl_ctrl_observe(self._h, self._ctrl_obs.data_ptr(), self._stream()), "pcgrl_ctrl_observe")
            return self._obs, {"ctrl_obs": self._ctrl_obs}
        return self._obs, {}

    def _check_action_shape(self, actions):
        if actions.numel() != self.num_envs * self.action_entries:
            raise ValueError(f"actions must hold {self.num_envs} x {self.action_entries} entries, got {tuple(actions.shape)}")

    def step(self, actions):
        self._check_action_shape(actions)
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if self._ex:
            rc = self._L.pcgrl_step_ex(
                self._h, actions.data_ptr(), 1 if self.auto_reset else 0, self._ptrs[0], self._ptrs[1],
                self._reward64.data_ptr() if self._reward64 is not None else None, self._ptrs[2], self._ptrs[3],
                self._ctrl_obs.data_ptr() if self._ctrl_obs is not None else None, self._stream())
        else:
            rc = self._L.pcgrl_step(self._h, actions.data_ptr(), 1 if self.auto_reset else 0, self._ptrs[0],
                                    self._ptrs[1], self._ptrs[2], self._ptrs[3], self._stream())
        if rc:
            _lib.check(rc, "pcgrl_step")
        return self._step_out

    # -- controllable generation (control_wrappers.py:27-121) -----------------------------------------------------
    def queue_targets(self, trgs, mask=None):
        """ControlWrapper.set_trgs: `trgs` = {metric: scalar | (lo, hi) | tensor [N]}; the targets take effect at each
        env's next reset (explicit or automatic), like the reference's _ctrl_trg_queue."""
        if not self.controls:
            raise ValueError("this env was built without `controls`")
        N, dev = self.num_envs, self.device
        lo = torch.zeros((N, self.n_stats), dtype=torch.float64, device=dev)
        hi = torch.zeros((N, self.n_stats), dtype=torch.float64, device=dev)
        for k, v in trgs.items():
            if k not in self.controls:
                raise ValueError(f"'{k}' is not a control metric of this env ({self.controls})")
            j = self.stat_keys.index(k)
            if isinstance(v, tuple):
                a, b = target_interval(v)
                lo[:, j], hi[:, j] = a, b
            else:
                t = torch.as_tensor(v, dtype=torch.float64, device=dev)
                lo[:, j] = t
                hi[:, j] = t
        m = None if mask is None else torch.as_tensor(mask, device=dev).to(torch.uint8).contiguous()
        _lib.check(self._L.pcgrl_queue_targets(self._h, m.data_ptr() if m is not None else None, lo.data_ptr(),
                                               hi.data_ptr(), self._stream()), "pcgrl_queue_targets")

    def sample_uniform_targets(self, generator=None, mask=None):
        """UniformNoiseyTargets.set_rand_trgs (control_wrappers.py:453-460): each control target ~ U(cond_bounds)."""
        trgs = {}
        for k in self.controls:
            lb, ub = self.spec.cond_bounds[k]
            u = torch.rand(self.num_envs, generator=generator, device=self.device, dtype=torch.float64)
            trgs[k] = u * (ub - lb) + lb
        self.queue_targets(trgs, mask=mask)
        return trgs

    def set_target_resampling(self, enable=True, seed=0):
        """UniformNoiseyTargets on the device (control_wrappers.py:442-471): from each env's next reset on -- explicit or
        automatic, also inside a captured HIP graph -- every control target is drawn ~ U(cond_bounds) from the env's own
        counter-based stream (pcgrl_set_target_resampling) and replaces whatever was queued.  `resampled_target` below
        restates the draw on the host."""
        if not self.controls:
            raise ValueError("this env was built without `controls`")
        lo = np.array([self.spec.cond_bounds[k][0] for k in self.controls], dtype=np.float64)
        hi = np.array([self.spec.cond_bounds[k][1] for k in self.controls], dtype=np.float64)
        _lib.check(self._L.pcgrl_set_target_resampling(self._h, 1 if enable else 0, int(seed) & (2 ** 64 - 1),
                                                       lo.ctypes.data, hi.ctypes.data), "pcgrl_set_target_resampling")

    def sample_actions(self, seed=0, out=None):
        """action_space.sample() for every env, drawn on the device (pcgrl_sample_actions: the reference's random-action
        loops, profile_env.py:134-139): int32 [N] (or [N, prod(act_window)]), fresh at every call and at every replay of a
        HIP graph that captured the call."""
        if out is None:
            shape = (self.num_envs, self.action_entries) if self.action_entries > 1 else (self.num_envs,)
            out = torch.empty(shape, dtype=torch.int32, device=self.device)
        _lib.check(self._L.pcgrl_sample_actions(self._h, out.data_ptr(), int(seed) & (2 ** 64 - 1), self._stream()),
                   "pcgrl_sample_actions")
        return out

    def reserve_solver_pool(self, n_slots=0, allow_lazy_growth=True):
        """sokoban: size the device solver's workspace pool now (synchronous) instead of inside the first step that sees
        the solver running; n_slots 0 = full size for this batch.  Returns the slots of the pool."""
        _lib.check(self._L.pcgrl_reserve_solver_pool(self._h, int(n_slots), 1 if allow_lazy_growth else 0),
                   "pcgrl_reserve_solver_pool")
        return self.solver_pool_slots()[0]

    def solver_pool_slots(self):
        """(slots now, full size for this batch, last growth failed?)"""
        full, failed = C.c_int32(0), C.c_int32(0)
        n = int(self._L.pcgrl_solver_pool_slots(self._h, C.byref(full), C.byref(failed)))
        return n, int(full.value), bool(failed.value)

    # -- asynchronous stepping (sokoban): resumable device solver behind a per-env status byte -----------------------
    EMITTED, BUSY = 1, 2  # include/pcgrl_amd.h PCGRL_ENV_EMITTED / PCGRL_ENV_BUSY

    def set_solver_budget(self, budget):
        """budget > 0: the device solver works to `budget` iteration units per env and launch and parks what it could not
        finish; step with step_ready() from here on (step() / rollout() / update() are refused).  0: synchronous again.
        In the reference a slow _run_game (sokoban_prob.py:99-148) stalls one env, not the fleet (rl/utils.py:412-415)."""
        _lib.check(self._L.pcgrl_set_solver_budget(self._h, int(budget)), "pcgrl_set_solver_budget")
        if budget > 0 and getattr(self, "_status", None) is None:
            self._status = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            info = dict(self._step_out[4], status=self._status)
            self._ready_out = self._step_out[:4] + (info,)

    def step_ready(self, actions):
        """pcgrl_step_ready: like step(), plus info["status"] uint8 [N] = EMITTED (this env completed a step in this launch:
        its reward / done / stats / obs rows are valid) | BUSY (a search of its level is parked: it ignores the NEXT call's
        action).  An env consumes the action of a call iff it was not busy after the previous one; an emitted transition
        belongs to the last action the env consumed.  Rows of envs that did not emit keep their previous contents."""
        self._check_action_shape(actions)
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        rc = self._L.pcgrl_step_ready(self._h, actions.data_ptr(), 1 if self.auto_reset else 0, self._ptrs[0], self._ptrs[1],
                                      self._ptrs[2], self._ptrs[3], self._status.data_ptr(), self._stream())
        if rc:
            _lib.check(rc, "pcgrl_step_ready")
        return self._ready_out

    def step_ready_raw(self, actions_ptr, status_ptr, stream):
        return self._L.pcgrl_step_ready(self._h, actions_ptr, 1 if self.auto_reset else 0, self._ptrs[0], self._ptrs[1],
                                        self._ptrs[2], self._ptrs[3], status_ptr, stream)

    def env_busy(self):
        """uint8 [N]: 1 = the env waits for a parked search (after reset(): which envs will ignore the first action)"""
        out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        _lib.check(self._L.pcgrl_env_busy(self._h, out.data_ptr(), self._stream()), "pcgrl_env_busy")
        return out

    def step_raw(self, actions_ptr, stream):
        """Lowest-overhead launch: device pointer of int32 actions + raw hipStream_t."""
        return self._L.pcgrl_step(self._h, actions_ptr, 1 if self.auto_reset else 0, self._ptrs[0], self._ptrs[1],
                                  self._ptrs[2], self._ptrs[3], stream)

    def step_seq_raw(self, rows_ptr, row_stride, n_rows, first_row, n_steps, stream):
        """n_steps pcgrl_step launches from ONE foreign call (pcgrl_step_seq): step k uses action row
        (first_row + k) % n_rows of the int32 buffer at rows_ptr (rows row_stride entries apart)."""
        return self._L.pcgrl_step_seq(self._h, rows_ptr, row_stride, n_rows, first_row, n_steps, 1 if self.auto_reset else 0,
                                      self._ptrs[0], self._ptrs[1], self._ptrs[2], self._ptrs[3], stream)

    def rollout(self, actions, want_obs="all"):
        """Open-loop rollout: `actions` int32 [K, N] (or [K, N, prod(act_window)] with an action patch); K steps in one
        launch (pcgrl_rollout / pcgrl_rollout_ex).  Returns (obs, reward [K, N], done [K, N], stats [K, N, n_stats]); obs
        is [K, N, ...] for want_obs="all", [N, ...] (after the last step) for "last", None for "none".  Fresh tensors, not
        the env's step buffers.  Controllable mode: rewards are float64 and `self.ctrl_obs` holds the control observation
        after the last step."""
        K = int(actions.shape[0])
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.numel() != K * self.num_envs * self.action_entries:
            raise ValueError(f"actions must be [K, {self.num_envs}" + (f", {self.action_entries}]" if self.action_entries > 1 else "]"))
        N, dev = self.num_envs, self.device
        obs = None
        if want_obs == "all":
            obs = torch.empty((K, N) + self.obs_shape, dtype=torch.uint8, device=dev)
        elif want_obs == "last":
            obs = torch.empty((N,) + self.obs_shape, dtype=torch.uint8, device=dev)
        rew = torch.empty((K, N), dtype=torch.float32, device=dev)
        rew64 = torch.empty((K, N), dtype=torch.float64, device=dev) if self._reward64 is not None else None
        done = torch.empty((K, N), dtype=torch.uint8, device=dev)
        stats = torch.empty((K, N, self.n_stats), dtype=torch.int32, device=dev)
        _lib.check(self._L.pcgrl_rollout_ex(self._h, actions.data_ptr(), K, 1 if self.auto_reset else 0,
                                            obs.data_ptr() if obs is not None else None, 1 if want_obs == "last" else 0,
                                            rew.data_ptr(), rew64.data_ptr() if rew64 is not None else None, done.data_ptr(),
                                            stats.data_ptr(), self._ctrl_obs.data_ptr() if self._ctrl_obs is not None else None,
                                            self._stream()), "pcgrl_rollout_ex")
        return obs, (rew64 if rew64 is not None else rew), done.view(torch.bool), stats

    @property
    def ctrl_obs(self):
        """float32 [N, 2 * len(controls)]: (target / range, metric / range) per control metric, as of the last step"""
        return self._ctrl_obs

    # -- evolution-driver pattern (evo/evolve.py:1083-1120): rep.update() many times, get_stats() once ---------------
    def update(self, actions, want_obs=True):
        """rep.update(action) for every env (+ observation); counters / stats / reward are untouched."""
        self._check_action_shape(actions)
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        _lib.check(self._L.pcgrl_update(self._h, actions.data_ptr(), self._ptrs[0] if want_obs else None, self._stream()),
                   "pcgrl_update")
        return self._obs if want_obs else None

    def refresh_stats(self):
        """Problem.get_stats() of the current maps; returns int32 [N, n_stats]."""
        _lib.check(self._L.pcgrl_refresh_stats(self._h, self._ptrs[3], self._stream()), "pcgrl_refresh_stats")
        return self._stats

    def observe(self):
        _lib.check(self._L.pcgrl_observe(self._h, self._ptrs[0], self._stream()), "pcgrl_observe")
        return self._obs

    # -- static tiles (envs/reps/wrappers.py:234-376) ---------------------------------------------------------------
    def get_static(self):
        """StaticTileRepresentation.static_tiles, uint8 [N, H+2, W+2] (bordered layout, border ring = 1)."""
        if not self.static_tiles:
            raise ValueError("this env was built without static tiles")
        h, w = self.map_shape
        out = torch.empty((self.num_envs, h + 2, w + 2), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.pcgrl_get_static(self._h, out.data_ptr(), self._stream()), "pcgrl_get_static")
        return out

    def set_static(self, static_prob=None, n_static_walls=None, eval_mode=None):
        """set_static_prob / set_n_static_walls / set_eval_mode (:256-263; rl/evaluate.py:128-129); next reset on.  None
        leaves a value as the engine holds it (also after load_state_dict of a checkpoint taken in another mode)."""
        _lib.check(self._L.pcgrl_set_static(self._h, -1.0 if static_prob is None else float(static_prob),
                                            -1 if n_static_walls is None else int(n_static_walls),
                                            -1 if eval_mode is None else int(bool(eval_mode))), "pcgrl_set_static")

    def check_errors(self):
        """Synchronises; raises ValueError if a kernel saw an action outside the action space."""
        _lib.check(self._L.pcgrl_poll_error(self._h), "pcgrl_poll_error")

    # -- state access ------------------------------------------------------------------------------
    def get_state(self):
        N, dev = self.num_envs, self.device
        out = SimpleNamespace(
            grids=torch.empty((N,) + self.map_shape, dtype=torch.uint8, device=dev),
            pos=torch.empty((N, 3), dtype=torch.int32, device=dev),
            counters=torch.empty((N, 4), dtype=torch.int32, device=dev),
            stats=torch.empty((N, self.n_stats), dtype=torch.int32, device=dev),
            last_loss=torch.empty(N, dtype=torch.float64, device=dev),
            ep_return=torch.empty(N, dtype=torch.float64, device=dev))
        _lib.check(self._L.pcgrl_get_state(self._h, out.grids.data_ptr(), out.pos.data_ptr(), out.counters.data_ptr(),
                                           out.stats.data_ptr(), out.last_loss.data_ptr(), out.ep_return.data_ptr(),
                                           self._stream()), "pcgrl_get_state")
        out.iteration, out.changes, out.n_step, out.ep_len = (out.counters[:, i] for i in range(4))
        return out

    def last_episode(self):
        N, dev = self.num_envs, self.device
        out = SimpleNamespace(ep_return=torch.empty(N, dtype=torch.float64, device=dev),
                              ep_len=torch.empty(N, dtype=torch.int32, device=dev),
                              final_stats=torch.empty((N, self.n_stats), dtype=torch.int32, device=dev),
                              n_episodes=torch.empty(N, dtype=torch.int64, device=dev))
        _lib.check(self._L.pcgrl_get_last_episode(self._h, out.ep_return.data_ptr(), out.ep_len.data_ptr(),
                                                  out.final_stats.data_ptr(), out.n_episodes.data_ptr(),
                                                  self._stream()), "pcgrl_get_last_episode")
        return out

    def stats_for_grids(self, grids):
        """Problem.get_stats on caller maps (the evolution driver's entry, evo/evolve.py:1083-1120).  Any number of
        maps; asynchronous (uses this engine's scratch; device-side errors surface in check_errors())."""
        g = torch.as_tensor(grids, device=self.device).to(torch.uint8).contiguous()
        n = g.numel() // self.n_cells
        out = torch.empty((n, self.n_stats), dtype=torch.int32, device=self.device)
        _lib.check(self._L.pcgrl_stats_for_grids_h(self._h, n, g.data_ptr(), out.data_ptr(), self._stream()),
                   "pcgrl_stats_for_grids_h")
        return out

    # -- episodic-return reduction (rl/callbacks.py:91-117 on_episode_end, summed over the batch) -------------------
    def reduce_episodes(self, clear=True, out=None):
        """float64 [3 + n_stats] on the env's GPU: sum of returns, sum of lengths, number of episodes, sum of final
        stats over the episodes that ended by auto-reset since the last clearing call.  One launch, no sync."""
        if out is None:
            out = torch.empty(3 + self.n_stats, dtype=torch.float64, device=self.device)
        _lib.check(self._L.pcgrl_reduce_episodes(self._h, out.data_ptr(), 1 if clear else 0, self._stream()),
                   "pcgrl_reduce_episodes")
        return out

    # -- checkpoint / restore (envs/pcgrl_env.py:102-112 get_task / set_task pickle the env) ------------------------
    def get_rng_state(self):
        """uint64-as-int64 [N, 10]: both numpy-compatible PCG64 streams of every env (see include/pcgrl_amd.h)."""
        out = torch.empty((self.num_envs, 10), dtype=torch.int64, device=self.device)
        _lib.check(self._L.pcgrl_get_rng_state(self._h, out.data_ptr(), self._stream()), "pcgrl_get_rng_state")
        return out

    def set_rng_state(self, rng, mask=None):
        r = torch.as_tensor(rng, device=self.device).to(torch.int64).contiguous()
        assert r.shape == (self.num_envs, 10)
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        _lib.check(self._L.pcgrl_set_rng_state(self._h, m.data_ptr() if m is not None else None, r.data_ptr(),
                                               self._stream()), "pcgrl_set_rng_state")

    def state_dict(self):
        """Everything needed to continue bit-exactly later.  `blob` is the engine's complete per-env state
        (pcgrl_export_state: maps, incremental-statistics masks, counters, statistics, returns and episode totals, RNG
        streams, static-tile masks / lagging bordered planes / spare RNG half, active and queued control targets, the 3-D
        move table and cached searches) -- what pickling the reference's env keeps (pcgrl_env.py:102-112,
        reps/wrappers.py:80-87); the portable fields (maps, positions, counters, returns, RNG streams) ride along."""
        st = self.get_state()
        blob = torch.empty(int(self._L.pcgrl_state_bytes(self._h)), dtype=torch.uint8, device=self.device)
        stale = C.c_int32(0)
        _lib.check(self._L.pcgrl_export_state(self._h, blob.data_ptr(), C.byref(stale), self._stream()), "pcgrl_export_state")
        return {"grids": st.grids.clone(), "pos": st.pos.clone(), "counters": st.counters.clone(),
                "ep_return": st.ep_return.clone(), "rng": self.get_rng_state(), "blob": blob, "maybe_stale": int(stale.value)}

    def load_state_dict(self, sd, mask=None):
        def dev(t, dtype):
            return torch.as_tensor(t, device=self.device).to(dtype).contiguous()

        m = None if mask is None else dev(mask, torch.uint8)
        if "blob" in sd:
            b = dev(sd["blob"], torch.uint8)
            if b.numel() != int(self._L.pcgrl_state_bytes(self._h)):
                raise ValueError("state_dict from an engine with another config or batch size")
            _lib.check(self._L.pcgrl_import_state(self._h, m.data_ptr() if m is not None else None, b.data_ptr(),
                                                  int(sd.get("maybe_stale", 1)), self._stream()), "pcgrl_import_state")
            return
        g, p, c = dev(sd["grids"], torch.uint8), dev(sd["pos"], torch.int32), dev(sd["counters"], torch.int32)
        r = dev(sd["ep_return"], torch.float64)
        assert g.numel() == self.num_envs * self.n_cells and p.shape == (self.num_envs, 3) and c.shape == (self.num_envs, 4)
        _lib.check(self._L.pcgrl_set_state(self._h, m.data_ptr() if m is not None else None, g.data_ptr(), p.data_ptr(),
                                           c.data_ptr(), r.data_ptr(), self._stream()), "pcgrl_set_state")
        self.set_rng_state(sd["rng"], mask=mask)


class SubBatchedVecEnv:
    """The batch cut into k independent sub-batches: k engines of N / k envs, each launched on a HIP stream of its own.

    A step launch lasts as long as its slowest env (one long path search); with k chains a launch only waits for the
    slowest env of ITS sub-batch and the chains overlap on the device (`async_sub_batches`: up to
    1.3 x on 64 x 64 maps and the 15^3 maze, a loss on the 16 x 16 maps whose launches are already short).  Envs are
    independent objects in the reference too (rl/utils.py:412-415 workers x envs); results are those of one batch of N.

      step(actions)              all k launches forked from / joined back into the current stream: drop-in for
                                 VecPcgrlEnv.step (outputs are [N, ...] tensors, sub-batch i owns rows [i*n, (i+1)*n)),
                                 also under HIP-graph capture (k parallel branches)
      step_async(i, actions_i)   launch sub-batch i alone on its stream -> its rows of the outputs; wait(i) makes the
                                 current stream wait for it (Sample-Factory-style double buffering: the policy runs on
                                 sub-batch i's observation while sub-batch j steps)
    """

    def __init__(self, problem, representation, map_shape, num_envs, sub_batches, device="cuda:0", seeds=None, **kw):
        k = int(sub_batches)
        if k < 1 or num_envs % k:
            raise ValueError("num_envs must be a multiple of sub_batches")
        self.k, self.num_envs, self.n_sub = k, int(num_envs), int(num_envs) // k
        self.device = torch.device(device)
        seeds = np.arange(num_envs) if seeds is None else np.broadcast_to(np.asarray(seeds), (num_envs,))
        n = self.n_sub
        probe = build_config(problem, representation, map_shape, kw.get("obs_window"), kw.get("weights"),
                             kw.get("max_board_scans", 3), kw.get("change_percentage"), kw.get("solver_power", 10000),
                             kw.get("static_trgs"), kw.get("controls"), kw.get("act_window"), kw.get("static_prob"),
                             kw.get("n_static_walls"), kw.get("static_eval", False))
        cfg, spec, obs_window = probe
        nt = spec.n_tiles
        if representation == "wide":
            obs_shape = tuple(map_shape) + (nt,)
        elif len(map_shape) == 3:
            obs_shape = tuple(obs_window) + (4,)
        else:
            obs_shape = tuple(obs_window) + (nt + 1 + (1 if cfg.static_tiles else 0),)
        N, dev = self.num_envs, self.device
        self._obs = torch.empty((N,) + obs_shape, dtype=torch.uint8, device=dev)
        self._reward = torch.empty(N, dtype=torch.float32, device=dev)
        self._done = torch.empty(N, dtype=torch.uint8, device=dev)
        self._stats = torch.empty((N, len(spec.stat_keys)), dtype=torch.int32, device=dev)
        self.streams = [torch.cuda.Stream(dev) for _ in range(k)]
        self.envs = []
        for i in range(k):
            sl = slice(i * n, (i + 1) * n)
            self.envs.append(VecPcgrlEnv(problem, representation, map_shape, n, device=dev, seeds=seeds[sl],
                                         _out=(self._obs[sl], self._reward[sl], self._done[sl], self._stats[sl]), **kw))
        e0 = self.envs[0]
        assert e0.obs_shape == obs_shape, (e0.obs_shape, obs_shape)
        for a in ("obs_shape", "num_actions", "action_entries", "stat_keys", "n_stats", "spec", "cfg", "map_shape", "auto_reset",
                  "problem", "representation"):
            setattr(self, a, getattr(e0, a))
        done = self._done.view(torch.bool)
        self._step_out = (self._obs, self._reward, done, done, {"stats": self._stats})

    def close(self):
        for e in self.envs:
            e.close()

    def _fork(self, i):
        self.streams[i].wait_stream(torch.cuda.current_stream(self.device))

    def wait(self, i=None):
        """the current stream waits for sub-batch i's stream (None: for all of them)"""
        cur = torch.cuda.current_stream(self.device)
        for j in (range(self.k) if i is None else (i,)):
            cur.wait_stream(self.streams[j])

    def reset(self, **kw):
        n = self.n_sub
        for i, e in enumerate(self.envs):
            self._fork(i)
            with torch.cuda.stream(self.streams[i]):
                e.reset(**{key: (None if v is None else torch.as_tensor(v)[i * n:(i + 1) * n]) for key, v in kw.items()})
        self.wait()
        return self._obs, {}

    def step_async(self, i, actions):
        """sub-batch i alone, on its own stream (ordered after the work already queued on the current stream); returns
        the sub-batch's step tuple -- valid once wait(i) has been called (or its stream synchronised)"""
        self._fork(i)
        # `actions` was allocated on the caller's stream and is read on streams[i]: a temporary (policy(obs).argmax().int())
        # freed right after this call must not be handed out again before that read has happened (not while capturing:
        # a captured graph keeps its buffers alive itself)
        if isinstance(actions, torch.Tensor) and actions.is_cuda and not torch.cuda.is_current_stream_capturing():
            actions.record_stream(self.streams[i])
        with torch.cuda.stream(self.streams[i]):
            return self.envs[i].step(actions)

    def step(self, actions):
        if self.envs[0]._ex:
            raise NotImplementedError("controllable mode / float64 rewards: use step_async(i, ...) (per-sub-batch outputs)")
        n = self.n_sub
        a = actions.reshape(self.num_envs, -1)
        for i in range(self.k):
            self.step_async(i, a[i * n:(i + 1) * n])
        self.wait()
        return self._step_out

    def get_state(self):
        self.wait()
        parts = [e.get_state() for e in self.envs]
        return SimpleNamespace(**{key: torch.cat([getattr(p, key) for p in parts]) for key in
                                  ("grids", "pos", "counters", "stats", "last_loss", "ep_return", "iteration", "changes", "n_step", "ep_len")})

    def reduce_episodes(self, clear=True):
        self.wait()
        return torch.stack([e.reduce_episodes(clear=clear) for e in self.envs]).sum(0)

    def last_episode(self):
        self.wait()
        parts = [e.last_episode() for e in self.envs]
        return SimpleNamespace(**{key: torch.cat([getattr(p, key) for p in parts]) for key in ("ep_return", "ep_len", "final_stats", "n_episodes")})

    def sample_actions(self, seed=0):
        """one device-side draw per sub-batch (sub-batch i uses seed + i: its engines keep their own draw counters)"""
        self.wait()
        return torch.cat([e.sample_actions(seed + i) for i, e in enumerate(self.envs)])

    def observe(self):
        for i, e in enumerate(self.envs):
            self._fork(i)
            with torch.cuda.stream(self.streams[i]):
                e.observe()
        self.wait()
        return self._obs

    def state_dict(self):
        """a list of the sub-batches' checkpoints (VecPcgrlEnv.state_dict); loads into an env with the same split"""
        self.wait()
        return {"sub_batches": [e.state_dict() for e in self.envs]}

    def load_state_dict(self, sd, mask=None):
        if len(sd.get("sub_batches", ())) != self.k:
            raise ValueError(f"state_dict of {len(sd.get('sub_batches', ()))} sub-batches, this env has {self.k}")
        self.wait()
        for i, (e, part) in enumerate(zip(self.envs, sd["sub_batches"])):
            e.load_state_dict(part, mask=self._rows(mask, i))

    def check_errors(self):
        for e in self.envs:
            e.check_errors()

    # -- the rest of VecPcgrlEnv's surface, fanned out over the sub-batches ---------------------------------------------
    def _rows(self, v, i):
        n = self.n_sub
        return None if v is None else torch.as_tensor(v)[i * n:(i + 1) * n]

    def seed(self, seeds):
        s = np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.num_envs,))
        for i, e in enumerate(self.envs):
            e.seed(s[i * self.n_sub:(i + 1) * self.n_sub])

    def set_static(self, static_prob=None, n_static_walls=None, eval_mode=None):
        for e in self.envs:
            e.set_static(static_prob, n_static_walls, eval_mode)

    def get_static(self):
        self.wait()
        return torch.cat([e.get_static() for e in self.envs])

    def queue_targets(self, trgs, mask=None):
        for i, e in enumerate(self.envs):
            e.queue_targets({k: (v if isinstance(v, tuple) or not hasattr(v, "__len__") else self._rows(v, i)) for k, v in trgs.items()},
                            mask=self._rows(mask, i))

    def set_target_resampling(self, enable=True, seed=0):
        raise NotImplementedError("target resampling draws from (seed, env index): per sub-batch the env indices restart at 0 -- use one "
                                  "VecPcgrlEnv, or call envs[i].set_target_resampling with distinct seeds")

    @property
    def ctrl_obs(self):
        self.wait()
        return None if self.envs[0].ctrl_obs is None else torch.cat([e.ctrl_obs for e in self.envs])

    def get_rng_state(self):
        self.wait()
        return torch.cat([e.get_rng_state() for e in self.envs])

    def set_rng_state(self, rng, mask=None):
        for i, e in enumerate(self.envs):
            e.set_rng_state(self._rows(rng, i), mask=self._rows(mask, i))

    def solver_pool_slots(self):
        """(slots, full size, failed) summed over the sub-batches' engines (each keeps a pool of its own)"""
        parts = [e.solver_pool_slots() for e in self.envs]
        return sum(p[0] for p in parts), sum(p[1] for p in parts), any(p[2] for p in parts)

    def rollout(self, actions, want_obs="all"):
        raise NotImplementedError("open-loop rollouts: use VecPcgrlEnv (a rollout's waves advance independently already)")


def make_vec_env(cfg, num_envs, device="cuda:0", seeds=None, auto_reset=True, sub_batches=1):
    """Batched counterpart of control_pcgrl/rl/envs.py:make_env(cfg).  `cfg` is the reference's Config-like
    object (attributes or dict keys): task.problem, task.map_shape, task.obs_window, task.weights,
    representation, max_board_scans, change_percentage."""
    unsupported = {
        "n_aux_tiles": _cfg_get(cfg, "n_aux_tiles", 0) or None,
        "show_agents": _cfg_get(cfg, "show_agents", False) or None,
        "multiagent.n_agents": _cfg_get(cfg, "multiagent.n_agents", 0) or None,
    }
    bad = {k: v for k, v in unsupported.items() if v not in (None, 0, False)}
    if bad:
        raise NotImplementedError(f"outside the accelerated hot path (SURVEY.md section 8f 'next'): {bad}")
    if int(sub_batches) > 1 and _cfg_get(cfg, "controls"):
        raise NotImplementedError("sub_batches > 1 with cfg.controls: SubBatchedVecEnv.step() has no float64 rewards / control "
                                  "observation of the whole batch (use sub_batches=1, or step_async per sub-batch)")
    ctor = VecPcgrlEnv if int(sub_batches) <= 1 else (lambda **kw: SubBatchedVecEnv(sub_batches=sub_batches, **kw))
    return ctor(
        problem=_cfg_get(cfg, "task.problem"), representation=_cfg_get(cfg, "representation"),
        map_shape=tuple(_cfg_get(cfg, "task.map_shape")), num_envs=num_envs, device=device,
        obs_window=_cfg_get(cfg, "task.obs_window"), weights=_cfg_get(cfg, "task.weights"),
        max_board_scans=_cfg_get(cfg, "max_board_scans", 3), change_percentage=_cfg_get(cfg, "change_percentage"),
        seeds=seeds, auto_reset=auto_reset, controls=_cfg_get(cfg, "controls"),
        reward_dtype=torch.float64 if _cfg_get(cfg, "controls") else torch.float32,
        act_window=_cfg_get(cfg, "act_window"), static_prob=_cfg_get(cfg, "static_prob"),
        n_static_walls=_cfg_get(cfg, "n_static_walls"))
