// pcgrl_common.h -- state layout shared by the host API (pcgrl_engine.hip) and the gfx950 kernels.
//
// HBM layout (all arrays owned by the engine, contiguous over envs):
//   planes  M[N][ROW_WORDS = 4][H]  planes 0..NB-1: tile grid as NB = ceil(log2(n_tiles)) bit-planes; word (e,k,r)
//                         holds bit k of the tile ids of row r (bit x = column x).  One wavefront lane owns one row, so
//                         a 16x16 binary map is 16 consecutive 32-bit words and a group of 16 lanes loads its env with
//                         one coalesced 64-byte access per plane.  binary: planes 1, 2 = fars / best (incremental
//                         path-length state), plane 3 = pre-flooded component of the next edit cell (PREFLOOD).
//   st      EnvState[N]   256-byte record: one hot 128-byte line of per-env scalars + the finished-episode totals.
//   rng     RngState[N]   two PCG64 streams (representation, problem), numpy-compatible.
//   xplanes M[N][1+NB][H] only with static tiles / action patches: static mask + lagging bordered-map planes
//   xstate  u32[N][4]     flags and the spare half of the representation RNG's last 64-bit draw (Generator.integers)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/pcgrl_amd.h"

namespace pcgrl {

// Per-env totals over the episodes finished since the last pcgrl_reduce_episodes (written only at an env's reset, summed
// in a fixed order by the reduction kernel: no atomics, deterministic).  rl/callbacks.py:91-117 reads the same values.
struct alignas(8) EpAcc {
  double sum_return;
  int64_t sum_len;
  int64_t n;
  int64_t sum_stats[PCGRL_MAX_STATS];
};

struct alignas(16) EnvState {
  int32_t pos[3];
  int32_t n_step;
  int32_t iteration;
  int32_t changes;
  int32_t flags;  // ENV_STATS_DIRTY (the episode length is `iteration`: both count PcgrlEnv.step calls since the reset)
  int32_t last_ep_len;
  double last_loss;
  double ep_return;
  double last_ep_return;
  int64_t n_episodes;
  int32_t stats[PCGRL_MAX_STATS];
  int32_t final_stats[PCGRL_MAX_STATS];
  // second 128-byte line: only touched when an episode ends (same base address as the hot line: no extra address registers
  // in the step kernel, whose occupancy at large batches hangs on a handful of VGPRs)
  EpAcc acc;
  int32_t pend_action;  // asynchronous stepping (ENV_PENDING_STEP): the action of the step that waits for its search
  uint8_t pad_[36];
};
static_assert(sizeof(EnvState) == 256, "EnvState must be two 128-byte lines");
static_assert(offsetof(EnvState, acc) == 128, "the episode totals start the second line");

// EnvState::flags
//   ENV_STATS_DIRTY  pcgrl_update changed the map and the statistics (and the binary fars / best masks) have not been
//                    refreshed: the next CHANGING step recomputes them from scratch, exactly like the reference's
//                    get_stats (pcgrl_env.py:314-323); cleared by that step, pcgrl_refresh_stats and every reset.
constexpr int32_t ENV_STATS_DIRTY = 1;
// Asynchronous stepping (pcgrl_step_ready, sokoban: the device solver works to a per-launch iteration budget and parks an
// unfinished search in the env's own workspace, pcgrl_sokoban.h "resumable"):
//   ENV_PENDING_STEP   the env has taken an action (pend_action) whose statistics wait for a parked search.  The record still
//                      holds the state BEFORE that step; every launch re-plays the step up to the solver call and resumes the
//                      search there, and the launch in which it ends completes the step and emits its outputs.
//   ENV_PENDING_STATS  the map is final (a reset, explicit or automatic) but its statistics -- and with them last_loss -- wait
//                      for a parked search: the env takes no action until a launch has finished them.
// Either way the env is BUSY: it ignores the actions it is handed and its output rows are not written.
constexpr int32_t ENV_PENDING_STEP = 2;
constexpr int32_t ENV_PENDING_STATS = 4;



struct alignas(16) RngState {
  uint64_t rep[4];   // state_hi, state_lo, inc_hi, inc_lo
  uint64_t prob[4];
};

// LCG skip-ahead by k draws:  state' = A*state + G*inc  (mod 2^128), G = 1 + a + ... + a^(k-1)
struct alignas(16) JumpEntry {
  uint64_t a_hi, a_lo, g_hi, g_lo;
};

// Controllable mode, device-side target resampling (pcgrl_set_target_resampling; the reference's UniformNoiseyTargets.reset,
// control_wrappers.py:453-471: every control target ~ U(cond_bounds) at every reset).  The record sits in device memory
// IMMEDIATELY BEFORE Params::trg (one 256-byte section: the kernel-argument block has no room for another pointer), is read
// by a kernel when an env resets, and is engine-wide run-time state like the static-tile parameters.
struct alignas(256) TrgResample {
  int32_t enable, pad_;
  uint64_t seed;
  double lo[PCGRL_MAX_STATS], hi[PCGRL_MAX_STATS];  // cond_bounds of control j (cfg.controls order)
};
static_assert(sizeof(TrgResample) == 256, "TrgResample: one 256-byte section in front of the targets");

struct Params {
  pcgrl_config cfg;
  int32_t n_envs;
  int32_t n_tiles;
  int32_t n_cells;
  int32_t obs_chunks;  // 16-byte chunks per observation row
  int32_t obs16;       // bit 0: 16x16 map, 32x32 window: the general kernels use the compile-time observation encoder, too;
                       // bit 1: that encoder's stores are non-temporal (observations per launch far beyond the last-level cache)
  int32_t obs_codes;   // general kernels, cropped window: > 0 = observation chunks are generated from per-cell tile codes in
                       // LDS (encode_obs_codes); the value is the byte stride of a code row.  0 = one-hot rows in LDS
  void *planes;
  EnvState *st;
  RngState *rng;
  const JumpEntry *jump;  // [H+1]: skip by row*W draws; entry H = H*W draws
  int32_t *err;           // device error word
  void *soko;             // SokoPool* (sokoban solver workspace), else null
  int32_t *solver_seen;   // host-mapped counter of device solver runs (sokoban), else null
  // the static targets as integers when all of them are integral or infinite (see get_loss); int_targets = 0 otherwise
  int32_t trg_lo_i[PCGRL_MAX_STATS], trg_hi_i[PCGRL_MAX_STATS];
  int32_t int_targets;
  int32_t spread;         // sokoban step: one env per wave pair (see step_kernel)
  int32_t sk_helpers;     // sokoban: A* stages run by helper wavefronts (0 or 3; two waves per stage, see pcgrl_sokoban.h)
  // asynchronous stepping (pcgrl_step_ready): solver iteration units per env and launch (0 = synchronous) and the per-env
  // status byte of the launch (PCGRL_ENV_EMITTED | PCGRL_ENV_BUSY); the per-env workspaces hang off the SokoPool
  int32_t sk_budget;
  uint8_t *ready;
  // per-call I/O
  const int32_t *actions;
  uint8_t *obs;
  float *reward;
  uint8_t *done;
  int32_t *stats_out;
  int32_t auto_reset;
  int32_t update_only;   // pcgrl_update: representation update + observation only (no counters, stats, reward)
  int32_t no_fast;       // stale statistics may exist (after pcgrl_update): step / rollout use the general kernels
  int32_t refresh_only;  // pcgrl_refresh_stats: recompute the stats of the current maps (reset kernel without a new map)
  const uint8_t *mask;
  const uint8_t *init_grids;
  const int32_t *init_pos;
  // pcgrl_set_state (reset kernel with init_grids): counters / loss / return to restore, any may be null
  const int32_t *in_counters;   // [N][4] iteration, changes, n_step, (ignored)
  const double *in_ep_return;   // [N]
  // get_state outputs
  uint8_t *out_grids;
  int32_t *out_pos;
  int32_t *out_counters;
  double *out_last_loss;
  double *out_ep_return;
  int32_t *out_ep_len;
  int64_t *out_n_episodes;
  // controllable mode (cfg.n_ctrl > 0): per-env target intervals, [N][PCGRL_MAX_STATS][2] = (lo, hi)
  double *trg;            // active targets (null in plain mode: cfg.trg_lo/hi apply to every env)
  double *trg_pending;    // queued by pcgrl_queue_targets, applied at the env's next reset
  int32_t *trg_flag;      // [N] bit 0 = pending targets waiting; bits 1.. = resets that resampled this env's targets so far (draw counter)
  double *reward64;       // per-call outputs of pcgrl_step_ex
  float *ctrl_obs;
  // representation wrappers (cfg.static_tiles / cfg.act_window), see "ext" in pcgrl_kernels2d.h
  int32_t ext;              // 1 when static tiles or an action patch are configured (selects the general kernels)
  int32_t n_act;            // action entries per env: prod(act_window) or 1
  int32_t lds_pair_bytes;   // LDS bytes of one (simulate, observe) wave pair of the step kernel
  int32_t set_state;        // pcgrl_set_state (see in_counters; here: it fills the padding, the block stays within 14 lines)
  void *xplanes;            // M[N][1+NB][H]: static mask in map coordinates, then the lagging bordered-map tile planes
  uint32_t *xstate;         // [N][4]: flags (bit 0: bordered map lags behind the map), rep-RNG spare 32 bits: has, value
  const JumpEntry *jump_b;  // [H+3]: skip r*(W+2) draws (rows of the bordered static mask)
  uint8_t *out_static;      // pcgrl_get_static output
  // pcgrl_rollout: several steps per launch
  int32_t n_steps;          // steps per launch; actions / reward / done / stats are [n_steps][N]...
  int32_t obs_last_only;    // 0: obs is [n_steps][N][obs_env_bytes]; else only the last step's observation [N][...]
  int64_t obs_env_bytes;    // observation bytes per env
  // 3-D: plane bits (y * X + x, up to 256 of them) whose x is not 0 / not X - 1: constants of the map shape, filled in by
  // pcgrl_create (every wave used to rebuild them with Y multi-word range fills: 5 us of a 15^3 step)
  uint64_t m3_notx0[4], m3_notxl[4];
};

}  // namespace pcgrl
