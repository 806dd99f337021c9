/*
 * pcgrl_oracle.h -- CPU restatement of control-pcgrl's env hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP engine and the timed
 * `cpu_baseline` leg of bench.py.  Nothing in the product path (control_pcgrl_amd/) may call it.
 * It is pinned against golden vectors captured from the reference itself (tests/golden/, produced by
 * oracle/gen_golden.py importing /root/reference) -- see tests/test_oracle_golden.py.
 *
 * Every function in pcgrl_oracle.c cites the reference file:line it restates
 * (paths relative to /root/reference/control_pcgrl/).
 */
#ifndef PCGRL_ORACLE_H
#define PCGRL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_STATS 8
#define ORC_MAX_TILES 8

enum { ORC_PROB_BINARY = 0, ORC_PROB_ZELDA = 1, ORC_PROB_SOKOBAN = 2, ORC_PROB_MC3DMAZE = 3 };
enum { ORC_REP_NARROW = 0, ORC_REP_TURTLE = 1, ORC_REP_WIDE = 2 };

/* Canonical stat order per problem:
 *   binary   : regions, path-length
 *   zelda    : player, key, door, enemies, regions, nearest-enemy, path-length
 *   sokoban  : player, crate, target, regions, dist-win, sol-length, ratio
 *   mc3dmaze : regions, path-length, n_jump                                                    */
typedef struct {
  int32_t problem;          /* ORC_PROB_* */
  int32_t representation;   /* ORC_REP_*  */
  int32_t ndim;             /* 2 or 3 */
  int32_t dims[3];          /* map_shape: 2-D {H, W, 1}; 3-D {Z(height), Y(width), X(length)} */
  int32_t obs_window[3];    /* cfg.task.obs_window (narrow/turtle crop; wide: == map_shape) */
  int32_t max_iterations;   /* prod(map_shape) * max_board_scans + 1  (envs/pcgrl_env.py:241) */
  int32_t max_changes;      /* max(int(change_percentage * prod(map_shape)), 1), or -1 for None (:235-239) */
  int32_t n_stats;
  int32_t has_trg[ORC_MAX_STATS];  /* 1 if the stat is in static_trgs (enters the loss) */
  double weights[ORC_MAX_STATS];   /* ControlWrapper.metric_weights (control_wrappers.py:41-45) */
  double trg_lo[ORC_MAX_STATS];    /* inclusive target interval: scalar t -> [t,t]; tuple (lo,hi) -> */
  double trg_hi[ORC_MAX_STATS];    /*   [lo, last element of arange(lo,hi)]  (control_wrappers.py:337-341) */
  int32_t solver_power;     /* sokoban: iterations per solver stage (sokoban_prob.py:40) */
  /* controllable generation (control_wrappers.py:27-121): stats observed + re-targeted per episode */
  int32_t n_ctrl;
  int32_t ctrl_idx[ORC_MAX_STATS];   /* stat index of each control metric, in cfg.controls order */
  double ctrl_range[ORC_MAX_STATS];  /* param_ranges[k] = |cond_bounds[k][1] - cond_bounds[k][0]|  (:70-73) */
  /* representation wrappers (envs/reps/wrappers.py:725-727 wrap_rep) */
  int32_t act_window[3];    /* cfg.act_window (MultiActionRepresentation :397-545, narrow only); {0,0,0} = None */
  int32_t static_tiles;     /* cfg.static_tile_wrapper (StaticTileRepresentation :234-376, narrow / turtle) */
  int32_t n_static_walls;   /* cfg.n_static_walls or 0 */
  int32_t static_eval;      /* StaticTileRepresentation._eval_mode (:262-263) */
  double static_prob;       /* cfg.static_prob or 0 */
} orc_config;

typedef struct orc_engine orc_engine;

orc_engine *orc_create(const orc_config *cfg, int32_t n_envs);
void orc_destroy(orc_engine *e);
void orc_set_threads(orc_engine *e, int32_t n_threads);

/* Seed both RNG streams of env i with PCG64(SeedSequence(seeds[i]))  (envs/pcgrl_env.py:142-146). */
void orc_seed(orc_engine *e, const uint64_t *seeds);

/* reset(): mask NULL = all envs.  init_grids/init_pos non-NULL = inject (no RNG draw at all). */
void orc_reset(orc_engine *e, const uint8_t *mask, const uint8_t *init_grids, const int32_t *init_pos);

/* With cfg.act_window set, every `actions` argument below is int32 [N][prod(act_window)] (MultiDiscrete, row-major
 * patch); otherwise int32 [N].
 * step(): outputs may be NULL.  obs is the uint8 one-hot observation AFTER an auto-reset, reward/done/
 * stats are those of the step itself (RLlib auto-reset convention). */
void orc_step(orc_engine *e, const int32_t *actions, int32_t auto_reset, uint8_t *obs, double *reward,
              uint8_t *done, int32_t *stats);
/* ... for the envs selected by mask (NULL = all); the other envs are not stepped, their output rows are untouched */
void orc_step_masked(orc_engine *e, const uint8_t *mask, const int32_t *actions, int32_t auto_reset, uint8_t *obs,
                     double *reward, uint8_t *done, int32_t *stats);

/* evolution-driver pattern (evo/evolve.py:1083-1120): rep.update only, then get_stats once */
void orc_update(orc_engine *e, const int32_t *actions, uint8_t *obs);
void orc_refresh_stats(orc_engine *e, int32_t *stats);

void orc_observe(orc_engine *e, uint8_t *obs);
int64_t orc_obs_size(const orc_engine *e); /* bytes per env */

/* counters per env: iteration, changes, n_step, episode_len */
void orc_get_state(orc_engine *e, uint8_t *grids, int32_t *pos, int32_t *counters, int32_t *stats,
                   double *last_loss, double *ep_return);
/* results of the last finished episode per env (valid after an auto-reset): return, length, final stats */
void orc_get_last_episode(orc_engine *e, double *ep_return, int32_t *ep_len, int32_t *final_stats,
                          int64_t *n_episodes);

/* set_trgs() (control_wrappers.py:168-172): queue per-env targets [N][n_stats] (inclusive interval lo..hi; only the
 * control metrics' columns are read); they replace the env's targets at its next reset (:174-178). mask NULL = all. */
void orc_queue_targets(orc_engine *e, const uint8_t *mask, const double *trg_lo, const double *trg_hi);
/* observe_metric_trgs (:189-214): per env 2*n_ctrl values (target / range, metric / range), float64 */
void orc_get_ctrl_obs(orc_engine *e, double *out);

/* Stateless Problem.get_stats() on n grids. */
void orc_stats_for_grids(const orc_config *cfg, int32_t n, const uint8_t *grids, int32_t *stats_out);
/* the same, OpenMP over maps (bench.py's cpu_baseline leg of the evolution-pattern workloads) */
void orc_stats_for_grids_mt(const orc_config *cfg, int32_t n, const uint8_t *grids, int32_t *stats_out, int32_t threads);

/* StaticTileRepresentation.static_tiles per env: uint8 [N][(H+2)*(W+2)] (bordered shape, wrappers.py:267) */
void orc_get_static(orc_engine *e, uint8_t *out);
void orc_set_static(orc_engine *e, double static_prob, int32_t n_static_walls, int32_t eval_mode);

/* RNG known-answer hooks: state after seeding, and a stream of doubles. */
void orc_rng_probe(uint64_t seed, int32_t n, uint64_t state_out[4], double *doubles_out);

#ifdef __cplusplus
}
#endif
#endif
