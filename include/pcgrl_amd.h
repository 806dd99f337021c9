/*
 * pcgrl_amd.h -- C ABI of the MI355X-native batched PCGRL environment engine (libpcgrl_amd.so).
 *
 * The reference (smearle/control-pcgrl) is pure Python and has no FFI; the boundary it exposes for this
 * path is the gym.Env interface of one environment object.  Each entry point below names the reference
 * interface it replaces for a *batch* of N environments resident on one GPU
 * (paths relative to the reference's control_pcgrl/ directory):
 *
 *   pcgrl_create / pcgrl_destroy   rl/envs.py:28-81 make_env(cfg)  ->  envs/pcgrl_env.py:39-94 PcgrlEnv.__init__
 *                                  + wrappers.py:443-476 / :502-526 wrapper stacks + control_wrappers.py:27-121
 *   pcgrl_seed                     envs/pcgrl_env.py:142-146 PcgrlEnv.seed (rep + problem RNG, PCG64(SeedSequence))
 *   pcgrl_reset                    control_wrappers.py:174-187 ControlWrapper.reset -> envs/pcgrl_env.py:158-188;
 *                                  init_grids != NULL additionally covers envs/pcgrl_ctrl_env.py:12-14 set_map
 *   pcgrl_step                     control_wrappers.py:216-244 ControlWrapper.step -> wrappers.py:126-132, :219-224,
 *                                  :394-399, :304-323 -> envs/pcgrl_env.py:267-342 PcgrlEnv.step
 *   pcgrl_observe                  wrappers.py:407-437 Cropped._transform, :232-257 OneHotEncoding._transform,
 *                                  :140-150 ToImage._transform
 *   pcgrl_get_state                env.unwrapped._rep._map / _rep._pos / _iteration / _changes / _rep_stats
 *                                  (read by rl/callbacks.py:91-117)
 *   pcgrl_get_last_episode         rl/callbacks.py:91-117 StatsCallbacks.on_episode_end (final stats, return)
 *   pcgrl_rollout                  the random-action rollout loop of profile_env.py:124-142 / train_reward_model.py:43-45
 *                                  (K x PcgrlEnv.step with actions that do not depend on the observations)
 *   pcgrl_get_static / _set_static envs/reps/wrappers.py:234-376 StaticTileRepresentation.static_tiles, set_static_prob ...
 *   pcgrl_stats_for_grids[_h]      envs/probs/problem.py:128 Problem.get_stats(map) as called directly by
 *                                  evo/evolve.py:1083-1120
 *   pcgrl_reduce_episodes          rl/callbacks.py:91-117 on_episode_end metrics, summed over the batch
 *   pcgrl_set_state / _rng_state   envs/pcgrl_env.py:102-112 get_task / set_task (env pickling = checkpoint / restore)
 *   pcgrl_export_state / _import   the same, complete: wrapped representation (reps/wrappers.py:80-87), control targets
 *
 * Conventions
 *   - every `d_` pointer is a DEVICE pointer on the engine's GPU; the caller owns all I/O buffers,
 *     the engine owns persistent env state.  `seeds` in pcgrl_seed is a HOST pointer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All calls except create /
 *     destroy / seed / poll_error are asynchronous on that stream.
 *   - return value 0 = success, otherwise a PCGRL_E* code; pcgrl_last_error() gives the message.
 *   - one handle per (process, GPU); a handle is not re-entrant; several handles may coexist.
 *
 * HIP graphs.  Every asynchronous entry point except pcgrl_import_state (it reads the image header on the host) may be
 * captured into a HIP graph (it only enqueues kernels and copies between engine-owned or caller-owned buffers on `stream`)
 * and replayed with new contents in the same buffers.  Three host-side decisions are taken when a launch is ISSUED and are
 * therefore frozen into a captured launch:
 *   - which step / rollout kernel runs: the compile-time 16x16 kernels carry no code for statistics left stale by
 *     pcgrl_update (the reference recomputes them from scratch at the next changing step, pcgrl_env.py:314-323), so after
 *     pcgrl_update the engine issues the general kernels until pcgrl_refresh_stats or a full reset.  A graph captured
 *     BEFORE pcgrl_update and replayed AFTER it would update stale statistics incrementally: such a launch raises a
 *     device error bit instead (pcgrl_poll_error returns PCGRL_ESTALE; the statistics of the affected envs are
 *     undefined until pcgrl_refresh_stats or a reset).  Re-capture after pcgrl_update, or call pcgrl_refresh_stats
 *     before replaying.
 *   - sokoban: whether a step launch gives every env a workgroup of its own and carries the solver's helper wavefronts
 *     (chosen from how recently the device solver ran).  Performance only; results do not depend on it.
 *   - pcgrl_set_static: the static-tile parameters in force when the launch was issued.
 */
#ifndef PCGRL_AMD_H
#define PCGRL_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCGRL_MAX_STATS 8

enum { PCGRL_PROB_BINARY = 0, PCGRL_PROB_ZELDA = 1, PCGRL_PROB_SOKOBAN = 2, PCGRL_PROB_MC3DMAZE = 3 };
enum { PCGRL_REP_NARROW = 0, PCGRL_REP_TURTLE = 1, PCGRL_REP_WIDE = 2 };
enum {
  PCGRL_OK = 0,
  PCGRL_EINVAL = 1,       /* bad argument / config */
  PCGRL_EUNSUPPORTED = 2, /* problem x representation x shape not supported by the kernels */
  PCGRL_EHIP = 3,         /* HIP runtime error */
  PCGRL_EACTION = 4,      /* an out-of-range action was seen on the device (pcgrl_poll_error) */
  PCGRL_ESTALE = 5        /* a captured launch met stale statistics (see "HIP graphs" below; pcgrl_poll_error) */
};

/* Stat order per problem (columns of every `stats` array):
 *   binary   : regions, path-length
 *   zelda    : player, key, door, enemies, regions, nearest-enemy, path-length
 *   sokoban  : player, crate, target, regions, dist-win, sol-length, ratio
 *   mc3dmaze : regions, path-length, n_jump                                                       */
typedef struct {
  int32_t problem;        /* PCGRL_PROB_* */
  int32_t representation; /* PCGRL_REP_*  */
  int32_t ndim;           /* 2 or 3 */
  int32_t dims[3];        /* cfg.task.map_shape: 2-D {H, W, 1}, H, W <= 64 (sokoban: H, W <= 62); 3-D {Z, Y, X}, each <= 16
                           * (BASELINE's 7 x 7 x 7, the reference's stock 15 x 15 x 15: configs/config.py:153-157) */
  int32_t obs_window[3];  /* cfg.task.obs_window (wide: must equal map_shape) */
  int32_t max_iterations; /* prod(map_shape) * cfg.max_board_scans + 1     (envs/pcgrl_env.py:241) */
  int32_t max_changes;    /* max(int(cfg.change_percentage * prod), 1) or -1 (envs/pcgrl_env.py:235-239) */
  int32_t n_stats;
  int32_t has_trg[PCGRL_MAX_STATS]; /* stat enters the loss (key of Problem.static_trgs) */
  double weights[PCGRL_MAX_STATS];  /* ControlWrapper.metric_weights (control_wrappers.py:41-45) */
  double trg_lo[PCGRL_MAX_STATS];   /* inclusive target interval; scalar target t: lo = hi = t;        */
  double trg_hi[PCGRL_MAX_STATS];   /* tuple (a, b): [a, last of arange(a, b)] (control_wrappers.py:337) */
  int32_t solver_power;             /* sokoban solver iterations per stage (sokoban_prob.py:40) */
  /* controllable generation (control_wrappers.py:27-121): ctrl_metrics observed and re-targeted per episode */
  int32_t n_ctrl;                       /* len(cfg.controls), 0 = plain mode */
  int32_t ctrl_idx[PCGRL_MAX_STATS];    /* stat column of each control metric, in cfg.controls order */
  double ctrl_range[PCGRL_MAX_STATS];   /* param_ranges[k] = |cond_bounds[k][1] - cond_bounds[k][0]| (:70-73) */
  /* representation wrappers (envs/reps/wrappers.py:720-727 wrap_rep), 2-D problems only */
  int32_t act_window[3];  /* cfg.act_window: MultiActionRepresentation (:397-545), narrow only; {0,0,0} = None */
  int32_t static_tiles;   /* cfg.static_tile_wrapper: StaticTileRepresentation (:234-376), narrow / turtle */
  int32_t n_static_walls; /* cfg.n_static_walls or 0 */
  int32_t static_eval;    /* StaticTileRepresentation._eval_mode (:262-263): static_prob is used as is */
  double static_prob;     /* cfg.static_prob or 0: upper bound of the per-episode static-tile probability (:269-278) */
} pcgrl_config;

typedef struct pcgrl_engine *pcgrl_handle;

int pcgrl_create(const pcgrl_config *cfg, int32_t n_envs, int32_t device, pcgrl_handle *out);
void pcgrl_destroy(pcgrl_handle h);

/* seeds: HOST pointer to n_envs uint64.  Synchronous. */
int pcgrl_seed(pcgrl_handle h, const uint64_t *seeds);

/* d_mask NULL = every env.  d_init_grids (uint8 [N][cells]) / d_init_pos (int32 [N][3]) non-NULL: start
 * from the given maps / agent positions instead of drawing them (no RNG consumption).  Injected maps carry no static
 * tiles; d_init_pos is ignored by the wide representation and with cfg.act_window (the patch position is a function
 * of the step counter alone). */
int pcgrl_reset(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_init_grids, const int32_t *d_init_pos,
                void *stream);

/* One env step for all N envs.  Any output pointer may be NULL.
 *   d_actions int32 [N], or int32 [N][prod(act_window)] when cfg.act_window is set (the reference's MultiDiscrete
 *             action: a row-major patch of tile ids, reps/wrappers.py:475-478) -- also for pcgrl_step_ex / pcgrl_update
 *   d_obs uint8 [N][obs_bytes]   d_reward float [N]
 *   d_done uint8 [N]         d_stats int32 [N][n_stats]
 * auto_reset != 0: an env whose episode ended is reset inside the same launch; reward/done/stats are
 * those of the finished step, the observation is the first one of the new episode (RLlib convention),
 * and the finished episode's return / length / final stats are latched for pcgrl_get_last_episode. */
int pcgrl_step(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward,
               uint8_t *d_done, int32_t *d_stats, void *stream);

/* n_steps pcgrl_step launches issued by one call: step k takes the action row (first_row + k) % n_rows of
 * d_action_rows (rows row_stride int32 apart) and writes the same output buffers.  Exactly what a C host's loop over
 * pcgrl_step does (one launch per step, each with its own actions); it exists so that a Python host pays the
 * foreign-call cost once per sequence instead of once per launch (the reference's random-action loop,
 * profile_env.py:124-142, and bench.py's eager launches). */
int pcgrl_step_seq(pcgrl_handle h, const int32_t *d_action_rows, int64_t row_stride, int32_t n_rows, int32_t first_row,
                   int32_t n_steps, int32_t auto_reset, uint8_t *d_obs, float *d_reward, uint8_t *d_done, int32_t *d_stats,
                   void *stream);

/* pcgrl_step with the extra outputs of the controllable mode (any pointer may be NULL):
 *   d_reward64 double [N]         the reward in float64 (float targets make the loss non-integral)
 *   d_ctrl_obs float [N][2*n_ctrl] observe_metric_trgs (control_wrappers.py:189-214): for control k, column 2k =
 *                                 target / range, column 2k+1 = metric / range -- the values of the 2*n_ctrl constant
 *                                 planes the reference prepends to the observation. */
int pcgrl_step_ex(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward,
                  double *d_reward64, uint8_t *d_done, int32_t *d_stats, float *d_ctrl_obs, void *stream);

/* env.action_space.sample() for every env of the batch, on the device: what the reference's own throughput loops feed
 * step() with (profile_env.py:134-139; random exploration in rl/train.py:144-184).  d_actions int32 [N] (or
 * [N][prod(act_window)]), each entry uniform in [0, pcgrl_num_actions(h)).  Counter-based: entry i of the engine's c-th draw
 * under `seed` is a pure function of (seed, c, i); the draw counter lives on the device and advances with every launch, so
 * a [pcgrl_sample_actions -> pcgrl_step] pair captured in a HIP graph is a closed random-action loop with fresh actions at
 * every replay.  Synthetic input, not env state: the counter is not part of pcgrl_export_state. */
int pcgrl_sample_actions(pcgrl_handle h, int32_t *d_actions, uint64_t seed, void *stream);
int32_t pcgrl_num_actions(pcgrl_handle h); /* narrow: n_tiles; turtle: n_tiles + 4; wide: cells * n_tiles; act_window: n_tiles */

/* Open-loop rollout: n_steps consecutive pcgrl_step()s of every env in ONE launch, for action sequences that do not
 * depend on the observations (random-action rollouts as in the reference's own env tests, replays, evaluation of stored
 * action sequences).  Results are identical to n_steps calls of pcgrl_step; there is no kernel boundary between steps.
 * (One launch where that is the faster form: 2-D maps of up to 16 rows and 32 columns and the 3-D mazes -- 1.7 x / 2 x the
 * stepping rate for binary-narrow / the 7^3 maze.  On 2-D maps of more than 16 rows or more than 32 columns the observation
 * wants a wavefront of its own next to the statistics and the call issues its n_steps as step launches instead: same results.)
 *   d_actions int32 [n_steps][N]          d_reward float [n_steps][N]      d_done uint8 [n_steps][N]
 *   d_stats   int32 [n_steps][N][n_stats] d_obs uint8 [n_steps][N][obs_bytes], or [N][obs_bytes] (the observation
 *             after the last step only) when obs_last_only != 0.  Any output pointer may be NULL.
 * With cfg.act_window the actions are int32 [n_steps][N][prod(act_window)].
 * pcgrl_rollout_ex adds the outputs of the controllable mode: d_reward64 double [n_steps][N], d_ctrl_obs float
 * [N][2*n_ctrl] (the control observation after the LAST step); queued targets take effect at the resets inside the
 * launch exactly as with pcgrl_step_ex. */
int pcgrl_rollout(pcgrl_handle h, const int32_t *d_actions, int32_t n_steps, int32_t auto_reset, uint8_t *d_obs,
                  int32_t obs_last_only, float *d_reward, uint8_t *d_done, int32_t *d_stats, void *stream);
int32_t pcgrl_rollout_is_one_launch(pcgrl_handle h); /* 1: one launch per call; 0: n_steps step launches (2-D maps of more than 16 rows or 32 columns) */
/* which form pcgrl_rollout[_ex] takes on this engine: -1 = chosen by shape (default), 0 = always n_steps step launches, 1 = always
 * the one-launch kernel (a simulate and an observe wavefront per workgroup), 2 = two kernels -- the simulate role on `stream`, the
 * observe role on a stream of the engine's own, forked from and joined back into `stream` with events (capturable like any fork /
 * join), reading a snapshot of the pre-call state: the roles never exchange anything, and apart each has its own register budget
 * (simulate: 103 / 114 / 157 VGPRs for binary / zelda / sokoban against 224 / 323 / 335 together).  The role kernels exist for
 * 16 x 16 maps with the default window in plain mode (no wrappers, no control metrics; form 2 is PCGRL_EUNSUPPORTED otherwise).
 * What -1 does on those configurations: a call with d_obs == NULL runs the simulate kernel alone; when the batch has more
 * workgroups than the two-role kernel keeps resident at once (asked of the runtime at pcgrl_create: more than 4096 binary or 2048
 * zelda / sokoban envs on an MI355X) a call for the last observation runs the simulate kernel and then pcgrl_observe's, and a call
 * of >= 16 steps for every observation takes form 2 (its snapshot + fork + join cost 15-25 us per call).  Same results in every
 * form (tests and the fuzzer exercise all of them).  PCGRL_ROLLOUT_KERNEL=0|1 sets the engine's initial value at pcgrl_create. */
int pcgrl_set_rollout_form(pcgrl_handle h, int32_t form);
int pcgrl_rollout_ex(pcgrl_handle h, const int32_t *d_actions, int32_t n_steps, int32_t auto_reset, uint8_t *d_obs,
                     int32_t obs_last_only, float *d_reward, double *d_reward64, uint8_t *d_done, int32_t *d_stats,
                     float *d_ctrl_obs, void *stream);

/* ControlWrapper.set_trgs (control_wrappers.py:168-172): queue per-env targets; they replace the env's targets at its
 * next reset (explicit or automatic), exactly like the reference's _ctrl_trg_queue (:174-178).
 * d_trg_lo / d_trg_hi: double [N][n_stats], inclusive interval per stat (lo == hi for a scalar target); only the
 * control metrics' columns are used.  d_mask NULL = every env.  Requires cfg.n_ctrl > 0. */
int pcgrl_queue_targets(pcgrl_handle h, const uint8_t *d_mask, const double *d_trg_lo, const double *d_trg_hi,
                        void *stream);
/* UniformNoiseyTargets (control_wrappers.py:442-471): at every reset each control target is drawn ~ U(cond_bounds[k]) and
 * replaces whatever was queued (its reset() overwrites the queue with the draw, :453-471).  The reference draws from numpy's
 * GLOBAL generator (no seed of its own); the engine draws on the device, inside the reset path of every kernel that resets an env
 * (pcgrl_reset, the auto-resets of pcgrl_step[_ex] / pcgrl_rollout[_ex]), from a counter-based stream per env: the c-th draw
 * of env i's control j under `seed` is  u * (hi[j] - lo[j]) + lo[j]  with u = (mix(seed, c, i, j) >> 11) * 2^-53  (trg_resampled,
 * csrc/pcgrl_kernels2d.h; the draw counter c is part of the env's checkpointed state).  A closed loop captured in a HIP graph
 * therefore re-targets at every episode end with no host call.  lo / hi: HOST arrays [n_ctrl], cfg.controls order
 * (Problem.cond_bounds).  Engine-wide run-time parameters like pcgrl_set_static's: synchronous, in force from each env's next
 * reset on, not part of pcgrl_export_state.  enable = 0 switches back to queued targets. */
int pcgrl_set_target_resampling(pcgrl_handle h, int32_t enable, uint64_t seed, const double *lo, const double *hi);
/* the control observation of the current state (after reset): float [N][2*n_ctrl] */
int pcgrl_ctrl_observe(pcgrl_handle h, float *d_ctrl_obs, void *stream);

/* Evolution-driver pattern (evo/evolve.py:1083-1120): the representation is updated directly, PcgrlEnv.step() is not
 * involved and the statistics are only computed at the end.
 *   pcgrl_update         rep.update(action) for every env + the observation (d_obs may be NULL); iteration / changes
 *                        counters, stats, reward and done are not touched.  An env whose map changed is marked "stats
 *                        stale": pcgrl_get_state keeps returning the old statistics (the reference's _rep_stats), and the
 *                        next pcgrl_step that changes its map recomputes them from scratch (pcgrl_env.py:314-323)
 *   pcgrl_refresh_stats  Problem.get_stats() of the current maps -> engine state (and d_stats int32 [N][n_stats] if
 *                        non-NULL); also re-bases the loss so that later pcgrl_step rewards are consistent */
int pcgrl_update(pcgrl_handle h, const int32_t *d_actions, uint8_t *d_obs, void *stream);
int pcgrl_refresh_stats(pcgrl_handle h, int32_t *d_stats, void *stream);

int pcgrl_observe(pcgrl_handle h, uint8_t *d_obs, void *stream);
int64_t pcgrl_obs_bytes(pcgrl_handle h); /* bytes per env: prod(obs_shape) */
int pcgrl_obs_shape(pcgrl_handle h, int32_t shape_out[4], int32_t *ndim_out);

/* d_counters int32 [N][4] = iteration, changes, n_step, episode length so far (== iteration).  Any pointer may be NULL. */
int pcgrl_get_state(pcgrl_handle h, uint8_t *d_grids, int32_t *d_pos, int32_t *d_counters, int32_t *d_stats,
                    double *d_last_loss, double *d_ep_return, void *stream);
int pcgrl_get_last_episode(pcgrl_handle h, double *d_ep_return, int32_t *d_ep_len, int32_t *d_final_stats,
                           int64_t *d_n_episodes, void *stream);

/* StaticTileRepresentation.static_tiles (reps/wrappers.py:267-303): d_static uint8 [N][(H+2)*(W+2)], the reference's
 * bordered layout (entry (r+1, c+1) protects map cell (r, c); the border ring is always 1).  Requires cfg.static_tiles. */
int pcgrl_get_static(pcgrl_handle h, uint8_t *d_static, void *stream);
/* set_static_prob / set_n_static_walls / set_eval_mode (:256-263; used by rl/evaluate.py:128-129): take effect at each
 * env's next reset.  n_static_walls < 0, static_prob < 0 or eval_mode < 0 leave that value unchanged.  Host-side and
 * immediate: a pcgrl_export_state issued earlier on a stream that has not run yet (or a captured one replayed later) carries
 * the parameters in force when its copy EXECUTES -- order the call after exports that are to keep the old ones. */
int pcgrl_set_static(pcgrl_handle h, double static_prob, int32_t n_static_walls, int32_t eval_mode);

/* Problem.get_stats on n caller-provided maps (evo/evolve.py:1083-1120 calls it once per individual):
 * d_grids uint8 [n][cells] -> d_stats int32 [n][n_stats].  Asynchronous on `stream`.
 *   pcgrl_stats_for_grids_h  uses the scratch (error word, sokoban solver workspace) of an existing engine of the same
 *                            problem and map shape; n is independent of the engine's batch size; device-side errors
 *                            (a level beyond the solver's limits: more than 512 crate / target pairs) are reported by
 *                            pcgrl_poll_error(h).  sokoban: the solver's workspace pool is sized for the engine's own
 *                            envs (one 46 MB slot per four; at most 64 slots until the solver has been seen running); the
 *                            first call whose n is much larger grows it once, synchronously (one slot per four maps,
 *                            at most 256 = 11.8 GB at solver_power 10000) -- not to be captured in a HIP graph
 *   pcgrl_stats_for_grids    handle-less: keeps one hidden scratch engine per (problem, map shape, solver_power, device),
 *                            created on first use; its device-side errors are read with pcgrl_stats_poll_error (which
 *                            synchronises); pcgrl_stats_cache_clear frees the hidden engines */
int pcgrl_stats_for_grids_h(pcgrl_handle h, int32_t n, const uint8_t *d_grids, int32_t *d_stats, void *stream);
int pcgrl_stats_for_grids(const pcgrl_config *cfg, int32_t n, const uint8_t *d_grids, int32_t *d_stats,
                          int32_t device, void *stream);
int pcgrl_stats_poll_error(const pcgrl_config *cfg, int32_t device);
void pcgrl_stats_cache_clear(void);

/* Episodic-return reduction (the only quantity the data-parallel path ever exchanges; rl/callbacks.py:91-117
 * on_episode_end reads the same values per env).  Sums, over every env of the engine, the episodes that ended by
 * auto-reset since the last call with clear != 0:
 *   d_out double [3 + n_stats] = sum of returns, sum of lengths, number of episodes, sum of final stats (stat order)
 * One launch, fixed summation order (bit-reproducible), asynchronous on `stream`.  A multi-GPU caller all-reduces
 * d_out (RCCL) and divides by d_out[2]. */
int pcgrl_reduce_episodes(pcgrl_handle h, double *d_out, int32_t clear, void *stream);

/* Checkpoint / restore (envs/pcgrl_env.py:102-112 get_task / set_task pickle the whole env object).
 *   pcgrl_set_state      the inverse of pcgrl_get_state for the envs selected by d_mask (NULL = all): maps d_grids uint8
 *                        [N][cells], positions d_pos int32 [N][3] (NULL = origin), d_counters int32 [N][4] = iteration,
 *                        changes, n_step, (ignored) and the running return d_ep_return double [N] (either may be NULL =
 *                        zero).  Statistics and the loss base are recomputed from the map (they are a function of it).
 *                        Not available with static tiles / action patches.
 *   pcgrl_get/set_rng_state  both numpy-compatible PCG64 streams of every env (+ the buffered 32-bit half used by the
 *                        representation wrappers): uint64 [N][10] = rep {state hi, lo, inc hi, lo}, prob {...},
 *                        flags | has32 << 32, val32. */
int pcgrl_set_state(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_grids, const int32_t *d_pos,
                    const int32_t *d_counters, const double *d_ep_return, void *stream);
int pcgrl_get_rng_state(pcgrl_handle h, uint64_t *d_out, void *stream);
int pcgrl_set_rng_state(pcgrl_handle h, const uint8_t *d_mask, const uint64_t *d_in, void *stream);

/* Synchronises the device and returns PCGRL_EACTION if any kernel saw an out-of-range action since the
 * last poll (the reference raises IndexError there), PCGRL_EUNSUPPORTED if a level / search exceeded the device
 * solver's or the 3-D path search's limits, PCGRL_ESTALE if a captured launch met stale statistics (see "HIP graphs"),
 * PCGRL_EHIP on a pending HIP error, else 0. */
int pcgrl_poll_error(pcgrl_handle h);

/* Checkpoint / restore of EVERYTHING the engine keeps per env (envs/pcgrl_env.py:102-112 get_task / set_task: RLlib
 * pickles the whole env, wrapped representation included -- reps/wrappers.py:80-87): tile planes and incremental-
 * statistics masks, counters, statistics, losses, running / last-episode returns and totals, both RNG streams, and where
 * configured the static-tile mask with its lagging bordered-map planes and the spare half of the representation RNG's last
 * draw, the active and queued control targets, the 3-D maze's move table and cached searches.
 *   pcgrl_state_bytes   size of the buffer
 *   pcgrl_export_state  d_buf uint8 [pcgrl_state_bytes]; *maybe_stale_out (host, may be NULL) = 1 when statistics left
 *                       stale by pcgrl_update may be among them -- hand it back to pcgrl_import_state
 *   pcgrl_import_state  into an engine created with the same config and batch size; d_mask uint8 [N] (NULL = all envs).
 *                       The image starts with a 256-byte header (magic, fingerprint of every create-time config field +
 *                       batch size + library version + per-env layout, and the three static-tile parameters
 *                       pcgrl_set_static moves at run time); an image from any other engine is refused with PCGRL_EINVAL
 *                       before anything is overwritten.  A full import (d_mask NULL) also takes over the exporter's
 *                       static_prob / n_static_walls / eval mode (a curriculum's set_static call survives a checkpoint);
 *                       a masked import leaves the engine-wide parameters alone.  pcgrl_export_state copies the header
 *                       from pinned memory owned by the engine: capturable, replays carry the current parameters.  Reading the header waits for `stream` once (the only
 *                       synchronising call among the state entry points; not to be captured in a HIP graph).
 * The buffer is an opaque image for this library version and config; pcgrl_get_state / pcgrl_set_state remain the
 * portable (maps, positions, counters) form. */
int64_t pcgrl_state_bytes(pcgrl_handle h);
int pcgrl_export_state(pcgrl_handle h, uint8_t *d_buf, int32_t *maybe_stale_out, void *stream);
int pcgrl_import_state(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_buf, int32_t maybe_stale, void *stream);
/* Asynchronous stepping (sokoban).  In the reference a slow SokobanProblem._run_game (envs/probs/sokoban/sokoban_prob.py:99-148:
 * BFS, then A* with balance 1, 0.5, 0, up to solver_power iterations each) stalls ONE env object -- a Ray worker's
 * num_envs_per_worker at most -- not the fleet (rl/utils.py:412-415: num_rollout_workers x num_envs_per_worker).  pcgrl_step waits
 * for the slowest search of the whole batch; with a solver budget the searches are RESUMABLE instead:
 *   pcgrl_set_solver_budget(h, budget)   budget > 0: every env gets a solver workspace of its own (11.5 MB at the default
 *       solver_power; synchronous allocation on the first call) and a launch gives every search `budget` iteration units (a BFS
 *       iteration costs 1, an A* iteration 2: roughly the same time).  budget = 0: back to synchronous stepping (allowed when no
 *       env is busy).  Sokoban without control metrics, static tiles or action patches; PCGRL_EUNSUPPORTED otherwise.
 *   pcgrl_step_ready(...)                pcgrl_step + d_status uint8 [N], per env a combination of
 *       PCGRL_ENV_EMITTED  the env completed a step in this launch: its rows of d_reward / d_done / d_stats (and d_obs) are valid
 *       PCGRL_ENV_BUSY     the env is busy after this launch (a search of its level is parked): it IGNORES the next launch's
 *                          action, and unless EMITTED is set as well its output rows were not written
 *     An env that is not busy when a launch starts takes that launch's action; the step completes -- EMITTED -- in the launch in
 *     which its search ends (usually the same one).  An env whose auto-reset (or pcgrl_reset) drew a playable map is EMITTED |
 *     BUSY (or just busy) until a launch has finished the new episode's statistics; it then reports 0 once and takes the next
 *     action.  So: env i consumes the action of launch t  iff  (status[i] of launch t-1 & PCGRL_ENV_BUSY) == 0  (after
 *     pcgrl_reset: pcgrl_env_busy), and every emitted transition belongs to the last action the env consumed.  Per-env
 *     trajectories are exactly the reference's; which launch an env advances in depends on the budget alone (deterministic).
 *     d_obs rows of busy envs hold the observation of the step in flight (it does not depend on the search).
 *     With a budget set, pcgrl_step / pcgrl_step_ex / pcgrl_rollout / pcgrl_update are refused (PCGRL_EINVAL): they have no way
 *     to say "busy".  pcgrl_reset / pcgrl_set_state may leave envs busy (their level's search is parked the same way).
 *   pcgrl_env_busy(h, d_busy, stream)    uint8 [N]: 1 = the env is busy now (what the last launch's PCGRL_ENV_BUSY said, or what
 *       a pcgrl_reset left behind)
 * A search is resumed only for exactly the level it was started on (the park record carries the map); anything else -- a reset,
 * an imported checkpoint, a step abandoned by pcgrl_reset -- restarts it: results never depend on parked state. */
enum { PCGRL_ENV_EMITTED = 1, PCGRL_ENV_BUSY = 2 };
int pcgrl_set_solver_budget(pcgrl_handle h, int32_t budget);
int32_t pcgrl_get_solver_budget(pcgrl_handle h);
int pcgrl_step_ready(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward, uint8_t *d_done,
                     int32_t *d_stats, uint8_t *d_status, void *stream);
int pcgrl_env_busy(pcgrl_handle h, uint8_t *d_busy, void *stream);

/* sokoban: the device solver's workspace pool (one 46 MB slot per search in flight at the default solver_power).
 * pcgrl_create allocates min(full, 64) slots (full = clamp(N / 4, 4, 512)); by default the engine grows the pool to full
 * size the first time it has SEEN the solver running -- a synchronous hipMalloc + device synchronise inside whichever
 * pcgrl_step / pcgrl_reset / pcgrl_rollout call notices it (never while `stream` itself is being captured; a capture in
 * global mode on ANOTHER thread would be invalidated by it).  A caller that wants to choose the moment:
 *   pcgrl_reserve_solver_pool(h, n_slots, allow_lazy_growth)   synchronous; n_slots > 0: at least that many (<= 512),
 *       0: full size for the engine's batch, < 0: keep the pool as it is; allow_lazy_growth = 0 switches the implicit growth
 *       off for this engine.  PCGRL_EHIP when the allocation fails (nothing stays allocated from the failed attempt; the
 *       engine keeps working with the pool it has: searches beyond it wait for a slot -- speed, not results).
 *   pcgrl_solver_pool_slots(h, &full, &failed)   slots of the current pool; `failed` = 1 when the last (implicit or explicit)
 *       growth failed -- pcgrl_last_error() then holds the reason.  The implicit growth is not retried after a failure. */
int pcgrl_reserve_solver_pool(pcgrl_handle h, int32_t n_slots, int32_t allow_lazy_growth);
int32_t pcgrl_solver_pool_slots(pcgrl_handle h, int32_t *full_size_out, int32_t *grow_failed_out);

/* Host utilities (no reference counterpart: a ctypes / cgo host needs them to stay on the HIP runtime this library is
 * linked against, which owns the stream handles it is given): asynchronous device -> host copy on `stream`, wait for
 * `stream`, hipGraphUpload of an instantiated graph (hipGraphExec_t as void*). */
int pcgrl_copy_to_host(void *dst_host, const void *d_src, int64_t bytes, void *stream);
int pcgrl_stream_synchronize(void *stream);
int pcgrl_graph_upload(void *graph_exec, void *stream);

/* Development aid: copies n 64-bit device counters to `out` (HOST pointer) and zeroes them.  They are only written by
 * a library built with -DPCGRL_PHASE_TIMING (tools/phase_timing.py); otherwise all zero. */
int pcgrl_debug_counters(pcgrl_handle h, uint64_t *out, int32_t n);
const char *pcgrl_last_error(void);
const char *pcgrl_version(void);

#ifdef __cplusplus
}
#endif
#endif
