/*
 * pcgrl_oracle.c -- CPU restatement of control-pcgrl's env hot path (see pcgrl_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: parity checker for the HIP engine + bench.py's cpu_baseline leg.
 * Pinned against golden vectors captured from the reference (tests/test_oracle_golden.py).
 *
 * The algorithms deliberately follow the reference's *sequential* formulation (FIFO queues, one
 * flood fill / BFS per component) so that the oracle is an independent statement of the semantics
 * from the bit-parallel HIP kernels.  Reference paths are relative to /root/reference/control_pcgrl/.
 */
#include "pcgrl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* implemented in sokoban_solver.c / mc3d.c */
void orc_sokoban_solve(const uint8_t *grid, int H, int W, int power, int *dist_win, int *sol_len);
void orc_mc3d_stats(const uint8_t *grid, int Z, int Y, int X, int32_t *stats, int16_t *path_xyz,
                    int32_t *path_len);

#define MAX_CELLS 4096 /* up to 64x64 */

/* ------------------------------------------------------------------------------------------------
 * RNG: numpy SeedSequence + PCG64 (third-party arithmetic, not under /root/reference; numpy is pinned
 * to 1.23.5 by the reference's requirements.txt:16, algorithm unchanged through 2.2).  Used through
 * gymnasium.utils.seeding.np_random (envs/reps/representation.py:50-53, envs/probs/problem.py:79-81).
 * Known-answer tested against numpy in tests/test_oracle_golden.py::test_rng_matches_numpy.
 * ---------------------------------------------------------------------------------------------- */
typedef unsigned __int128 u128;
typedef struct {
  u128 state, inc;
  int has32;      /* spare high half of the last 64-bit draw (pcg64_next32) */
  uint32_t val32;
} pcg64;

#define PCG_MULT ((((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL)

static void seedseq_state(uint64_t seed, uint64_t out[4]) {
  const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
  const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
  uint32_t ent[2];
  int n_ent = 1;
  ent[0] = (uint32_t)seed;
  ent[1] = (uint32_t)(seed >> 32);
  if (ent[1]) n_ent = 2;
  uint32_t pool[4], hc = INIT_A;
#define HASHMIX(v, res)      \
  do {                       \
    uint32_t _v = (v);       \
    _v ^= hc;                \
    hc *= MULT_A;            \
    _v *= hc;                \
    _v ^= _v >> 16;          \
    (res) = _v;              \
  } while (0)
  for (int i = 0; i < 4; i++) HASHMIX(i < n_ent ? ent[i] : 0u, pool[i]);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) {
        uint32_t h;
        HASHMIX(pool[s], h);
        uint32_t r = MIX_L * pool[d] - MIX_R * h;
        r ^= r >> 16;
        pool[d] = r;
      }
#undef HASHMIX
  uint32_t w[8], hb = INIT_B;
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  for (int i = 0; i < 4; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

static void pcg64_seed(pcg64 *r, uint64_t seed) {
  uint64_t w[4];
  seedseq_state(seed, w);
  u128 initstate = ((u128)w[0] << 64) | w[1];
  u128 initseq = ((u128)w[2] << 64) | w[3];
  r->state = 0;
  r->has32 = 0;
  r->val32 = 0;
  r->inc = (initseq << 1) | 1;
  r->state = r->state * PCG_MULT + r->inc;
  r->state += initstate;
  r->state = r->state * PCG_MULT + r->inc;
}

static inline uint64_t pcg64_next(pcg64 *r) {
  r->state = r->state * PCG_MULT + r->inc;
  uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}

static inline double pcg64_double(pcg64 *r) { return (double)(pcg64_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* numpy/random/src/pcg64/pcg64.h pcg64_next32: a 64-bit draw serves two 32-bit requests, low half first; the
 * spare half survives any number of next64/next_double calls in between. */
static inline uint32_t pcg64_next32(pcg64 *r) {
  if (r->has32) {
    r->has32 = 0;
    return r->val32;
  }
  uint64_t n = pcg64_next(r);
  r->has32 = 1;
  r->val32 = (uint32_t)(n >> 32);
  return (uint32_t)n;
}

/* Generator.integers(low, high) for a scalar int64 request whose range fits 32 bits: numpy/random/src/distributions/
 * distributions.c random_bounded_uint64 -> buffered_bounded_lemire_uint32 (Lemire's nearly-divisionless rejection). */
static int64_t pcg64_integers(pcg64 *r, int64_t low, int64_t high) {
  uint32_t rng = (uint32_t)(high - 1 - low);
  if (rng == 0) return low;
  const uint32_t rng_excl = rng + 1;
  uint64_t m = (uint64_t)pcg64_next32(r) * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (leftover < rng_excl) {
    const uint32_t threshold = (uint32_t)(0u - rng_excl) % rng_excl;
    while (leftover < threshold) {
      m = (uint64_t)pcg64_next32(r) * rng_excl;
      leftover = (uint32_t)m;
    }
  }
  return low + (int64_t)(m >> 32);
}

void orc_rng_probe(uint64_t seed, int32_t n, uint64_t state_out[4], double *doubles_out) {
  pcg64 r;
  pcg64_seed(&r, seed);
  state_out[0] = (uint64_t)(r.state >> 64);
  state_out[1] = (uint64_t)r.state;
  state_out[2] = (uint64_t)(r.inc >> 64);
  state_out[3] = (uint64_t)r.inc;
  for (int i = 0; i < n; i++) doubles_out[i] = pcg64_double(&r);
}

/* ------------------------------------------------------------------------------------------------
 * Engine state
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  uint8_t *grid;
  int32_t pos[3];
  int32_t n_step, iteration, changes;
  int32_t stats[ORC_MAX_STATS];
  double last_loss, ep_return;
  int32_t ep_len;
  pcg64 rng_rep, rng_prob;
  /* last finished episode */
  double last_ep_return;
  int32_t last_ep_len;
  int32_t final_stats[ORC_MAX_STATS];
  int64_t n_episodes;
  /* controllable mode: active targets and targets queued for the next reset */
  double trg_lo[ORC_MAX_STATS], trg_hi[ORC_MAX_STATS], pend_lo[ORC_MAX_STATS], pend_hi[ORC_MAX_STATS];
  int32_t has_pending;
  /* 3-D maze: path overlay shown in the NEXT observation (minecraft_3D_maze_prob.py:84-93) */
  int16_t *path_xyz;
  int32_t path_len;
  /* StaticTileRepresentation: static_tiles in the bordered shape, and the interior of rep._bordered_map, which lags
   * behind rep._map between reset() (static walls are written into _map only, wrappers.py:309) and the first update */
  uint8_t *static_b, *bord;
  int32_t bord_stale;
} env_t;

struct orc_engine {
  orc_config cfg;
  int32_t n_envs, n_cells, n_tiles, n_threads;
  env_t *envs;
  uint8_t *grid_pool;
  int16_t *path_pool;
  uint8_t *static_pool;
  int32_t n_act; /* action entries per env: prod(act_window) or 1 */
};

static int n_tiles_of(int problem) {
  switch (problem) {
    case ORC_PROB_BINARY: return 2;   /* probs/binary/binary_prob.py:17 */
    case ORC_PROB_ZELDA: return 8;    /* probs/zelda/zelda_prob.py:20 */
    case ORC_PROB_SOKOBAN: return 5;  /* probs/sokoban/sokoban_prob.py:26 */
    case ORC_PROB_MC3DMAZE: return 2; /* probs/minecraft/minecraft_3D_maze_prob.py:26 */
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * 2-D grid algorithms (envs/helper.py)
 * ---------------------------------------------------------------------------------------------- */
static const int DX4[4] = {-1, 1, 0, 0}; /* helper.py:183, :235: (-1,0),(1,0),(0,-1),(0,1) as (dx,dy) */
static const int DY4[4] = {0, 0, -1, 1};

#define PASS(t, mask) (((mask) >> (t)) & 1u)

/* helper.py:173-187 _flood_fill + :200-210 calc_num_regions.  One FIFO flood fill per passable cell
 * that is still uncoloured; the count is order independent. */
static int calc_num_regions2d(const uint8_t *g, int H, int W, uint32_t passmask) {
  int16_t color[MAX_CELLS];
  int16_t q[MAX_CELLS * 4 + 4];
  int n = H * W, regions = 0;
  for (int i = 0; i < n; i++) color[i] = -1;
  for (int s = 0; s < n; s++) {
    if (!PASS(g[s], passmask) || color[s] != -1) continue;
    int head = 0, tail = 0, num = 0;
    q[tail++] = (int16_t)s;
    color[s] = -2; /* queued marker: keeps the queue bounded; the fill result is identical */
    while (head < tail) {
      int c = q[head++];
      int cx = c % W, cy = c / W;
      num++;
      color[c] = (int16_t)(regions + 1);
      for (int d = 0; d < 4; d++) {
        int nx = cx + DX4[d], ny = cy + DY4[d];
        if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
        int ni = ny * W + nx;
        if (color[ni] != -1 || !PASS(g[ni], passmask)) continue;
        color[ni] = -2;
        q[tail++] = (int16_t)ni;
      }
    }
    if (num > 0) regions++;
  }
  return regions;
}

/* helper.py:225-240 run_dijkstra: FIFO BFS; dist = -1 for impassable / unreachable. */
static void run_dijkstra2d(const uint8_t *g, int H, int W, uint32_t passmask, int sx, int sy, int16_t *dist) {
  int16_t q[MAX_CELLS + 4];
  int n = H * W;
  for (int i = 0; i < n; i++) dist[i] = -1;
  int s = sy * W + sx;
  if (!PASS(g[s], passmask)) return;
  int head = 0, tail = 0;
  dist[s] = 0;
  q[tail++] = (int16_t)s;
  while (head < tail) {
    int c = q[head++];
    int cx = c % W, cy = c / W;
    for (int d = 0; d < 4; d++) {
      int nx = cx + DX4[d], ny = cy + DY4[d];
      if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
      int ni = ny * W + nx;
      if (!PASS(g[ni], passmask) || dist[ni] >= 0) continue;
      dist[ni] = (int16_t)(dist[c] + 1);
      q[tail++] = (int16_t)ni;
    }
  }
}

/* helper.py:255-276 calc_longest_path (value only; the get_path=True back-trace is render-only).
 * Components are entered at their first cell in `_get_certain_tiles` order (tile order of
 * passable_values, then row-major); far = np.argmax = first maximum in row-major order (:265);
 * strict '>' keeps the earliest best component (:268). */
static int calc_longest_path2d(const uint8_t *g, int H, int W, const int *pass_tiles, int n_pass) {
  uint8_t visited[MAX_CELLS];
  int16_t dist[MAX_CELLS];
  int n = H * W, final_value = 0;
  uint32_t passmask = 0;
  for (int k = 0; k < n_pass; k++) passmask |= 1u << pass_tiles[k];
  memset(visited, 0, (size_t)n);
  for (int k = 0; k < n_pass; k++) {
    for (int s = 0; s < n; s++) {
      if (g[s] != pass_tiles[k] || visited[s]) continue;
      run_dijkstra2d(g, H, W, passmask, s % W, s / W, dist);
      int far = 0, best = dist[0];
      for (int i = 0; i < n; i++) {
        if (dist[i] >= 0) visited[i] = 1;
        if (dist[i] > best) {
          best = dist[i];
          far = i;
        }
      }
      run_dijkstra2d(g, H, W, passmask, far % W, far / W, dist);
      int mx = dist[0];
      for (int i = 1; i < n; i++)
        if (dist[i] > mx) mx = dist[i];
      if (mx > final_value) final_value = mx;
    }
  }
  return final_value;
}

static int count_tile(const uint8_t *g, int n, uint32_t tilemask) {
  int c = 0;
  for (int i = 0; i < n; i++) c += PASS(g[i], tilemask);
  return c;
}

static int first_tile(const uint8_t *g, int n, int tile) { /* helper.py:19-26: row-major location lists */
  for (int i = 0; i < n; i++)
    if (g[i] == tile) return i;
  return -1;
}

/* probs/binary/binary_prob.py:152-158 */
static void stats_binary(const uint8_t *g, int H, int W, int32_t *st) {
  const int pass[1] = {0}; /* "empty" */
  st[0] = calc_num_regions2d(g, H, W, 1u << 0);
  st[1] = calc_longest_path2d(g, H, W, pass, 1);
}

/* probs/zelda/zelda_ctrl_prob.py:90-168.  Tiles (zelda_prob.py:20):
 * 0 empty 1 solid 2 player 3 key 4 door 5 bat 6 scorpion 7 spider */
static void stats_zelda(const uint8_t *g, int H, int W, int32_t *st) {
  int n = H * W;
  const uint32_t ENEMY = (1u << 5) | (1u << 6) | (1u << 7);
  const uint32_t WALK = (1u << 0) | (1u << 2) | (1u << 3) | ENEMY; /* :98-102, :117-124, :138-144 */
  const uint32_t WALK_DOOR = WALK | (1u << 4);                     /* :145-151 */
  int16_t dist[MAX_CELLS];
  st[0] = count_tile(g, n, 1u << 2);
  st[1] = count_tile(g, n, 1u << 3);
  st[2] = count_tile(g, n, 1u << 4);
  st[3] = count_tile(g, n, ENEMY);
  st[4] = calc_num_regions2d(g, H, W, WALK);
  st[5] = 0;
  st[6] = 0;
  if (st[0] == 1) {
    int p = first_tile(g, n, 2);
    if (st[3] > 0) {
      const int UPPER = W * H * 100; /* :114 */
      int min_dist = UPPER;
      run_dijkstra2d(g, H, W, WALK, p % W, p / W, dist);
      for (int i = 0; i < n; i++)
        if (PASS(g[i], ENEMY) && dist[i] > 0 && dist[i] < min_dist) min_dist = dist[i];
      if (min_dist == UPPER) min_dist = 0;
      st[5] = min_dist;
    }
    if (st[1] == 1 && st[2] == 1) {
      int k = first_tile(g, n, 3), d = first_tile(g, n, 4);
      run_dijkstra2d(g, H, W, WALK, p % W, p / W, dist);
      st[6] += dist[k]; /* may be -1 (Q7) */
      run_dijkstra2d(g, H, W, WALK_DOOR, k % W, k / W, dist);
      st[6] += dist[d];
    }
  }
}

/* probs/sokoban/sokoban_prob.py:160-180 + sokoban_ctrl_prob.py:58-65.  Tiles (sokoban_prob.py:26):
 * 0 empty 1 solid 2 player 3 crate 4 target */
static void stats_sokoban(const uint8_t *g, int H, int W, int power, int32_t *st) {
  int n = H * W;
  st[0] = count_tile(g, n, 1u << 2);
  st[1] = count_tile(g, n, 1u << 3);
  st[2] = count_tile(g, n, 1u << 4);
  st[3] = calc_num_regions2d(g, H, W, (1u << 0) | (1u << 2) | (1u << 3) | (1u << 4));
  st[4] = W * H * (W + H); /* :169, real map size (Problem.adjust_param restores it, problem.py:113-115) */
  st[5] = 0;
  if (st[0] == 1 && st[1] == st[2] && st[1] > 0 && st[3] == 1) {
    int dist_win, sol_len;
    orc_sokoban_solve(g, H, W, power, &dist_win, &sol_len);
    st[4] = dist_win;
    st[5] = sol_len;
  }
  st[6] = abs(st[1] - st[2]);
}

static void get_stats(const orc_config *cfg, const uint8_t *g, int32_t *st, int16_t *path_xyz, int32_t *path_len) {
  switch (cfg->problem) {
    case ORC_PROB_BINARY: stats_binary(g, cfg->dims[0], cfg->dims[1], st); break;
    case ORC_PROB_ZELDA: stats_zelda(g, cfg->dims[0], cfg->dims[1], st); break;
    case ORC_PROB_SOKOBAN: stats_sokoban(g, cfg->dims[0], cfg->dims[1], cfg->solver_power, st); break;
    case ORC_PROB_MC3DMAZE: orc_mc3d_stats(g, cfg->dims[0], cfg->dims[1], cfg->dims[2], st, path_xyz, path_len); break;
  }
}

void orc_stats_for_grids_mt(const orc_config *cfg, int32_t n, const uint8_t *grids, int32_t *stats_out, int32_t threads) {
  int cells = cfg->dims[0] * cfg->dims[1] * (cfg->ndim == 3 ? cfg->dims[2] : 1);
#pragma omp parallel num_threads(threads < 1 ? 1 : threads)
  {
    int16_t *path = (int16_t *)malloc(sizeof(int16_t) * 3 * (size_t)(cells * 4 + 8));
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
      int32_t st[ORC_MAX_STATS] = {0}, pl = 0;
      get_stats(cfg, grids + (size_t)i * cells, st, path, &pl);
      for (int k = 0; k < cfg->n_stats; k++) stats_out[(size_t)i * cfg->n_stats + k] = st[k];
    }
    free(path);
  }
}

void orc_stats_for_grids(const orc_config *cfg, int32_t n, const uint8_t *grids, int32_t *stats_out) {
  orc_stats_for_grids_mt(cfg, n, grids, stats_out, 1);
}

/* control_wrappers.py:318-345 get_loss: sum over static targets of -w * distance(value, target). */
static double get_loss(const orc_config *cfg, const env_t *ev, const int32_t *st) {
  double loss = 0.0;
  for (int k = 0; k < cfg->n_stats; k++) {
    if (!cfg->has_trg[k]) continue;
    double v = (double)st[k], d = 0.0;
    if (v < ev->trg_lo[k]) d = ev->trg_lo[k] - v;
    else if (v > ev->trg_hi[k]) d = v - ev->trg_hi[k];
    loss += (-d) * cfg->weights[k];
  }
  return loss;
}

/* ------------------------------------------------------------------------------------------------
 * Representations (envs/reps/)
 * ---------------------------------------------------------------------------------------------- */
static void unravel(const orc_config *cfg, int flat, int32_t *pos) {
  if (cfg->ndim == 2) {
    pos[0] = flat / cfg->dims[1];
    pos[1] = flat % cfg->dims[1];
    pos[2] = 0;
  } else {
    int yx = cfg->dims[1] * cfg->dims[2];
    pos[0] = flat / yx;
    pos[1] = (flat % yx) / cfg->dims[2];
    pos[2] = flat % cfg->dims[2];
  }
}

static int ravel(const orc_config *cfg, const int32_t *pos) {
  if (cfg->ndim == 2) return pos[0] * cfg->dims[1] + pos[1];
  return (pos[0] * cfg->dims[1] + pos[1]) * cfg->dims[2] + pos[2];
}

/* MultiActionRepresentation.update (reps/wrappers.py:466-524) around a narrow representation: the action is a
 * row-major act_window patch of tile ids centred on _pos (inner pads floor/ceil((k-1)/2), :404-410); positions walk
 * the row-major list of patch centres that keep the patch inside the map (get_act_coords :441-463). */
static int multi_action_update(const orc_engine *e, env_t *v, const int32_t *act) {
  const orc_config *cfg = &e->cfg;
  const int ah = cfg->act_window[0], aw = cfg->act_window[1], W = cfg->dims[1];
  const int l0 = (ah - 1) / 2, l1 = (aw - 1) / 2;
  const int ny = cfg->dims[0] - ah + 1, nx = W - aw + 1;
  int change = 0;
  for (int a = 0; a < ah; a++)
    for (int b = 0; b < aw; b++) {
      int idx = (v->pos[0] - l0 + a) * W + (v->pos[1] - l1 + b);
      uint8_t t = (uint8_t)act[a * aw + b];
      change |= v->grid[idx] != t;
      v->grid[idx] = t;
    }
  int k = v->n_step % (ny * nx); /* narrow_rep.py:137-138 get_pos_at_step with the pre-increment n_step */
  v->pos[0] = l0 + k / nx;
  v->pos[1] = l1 + k % nx;
  v->n_step++;
  return change;
}

static int base_update(const orc_engine *e, env_t *v, int action);

/* rep.update() through the wrapper stack of wrap_rep (reps/wrappers.py:720-727): StaticTile(MultiAction(rep)). */
static int rep_update(const orc_engine *e, env_t *v, const int32_t *act) {
  const orc_config *cfg = &e->cfg;
  int change = cfg->act_window[0] > 0 ? multi_action_update(e, v, act) : base_update(e, v, act[0]);
  if (cfg->static_tiles) { /* StaticTileRepresentation.update :349-366 */
    /* old_state = _bordered_map before the inner update; the inner update always re-syncs _bordered_map with _map */
    const uint8_t *old = v->bord;
    if (change > 0) {
      const int H = cfg->dims[0], W = cfg->dims[1];
      for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
          if (v->static_b[(r + 1) * (W + 2) + c + 1]) v->grid[r * W + c] = old[r * W + c];
      /* :362 compares old_state with the pre-revert array, so `change` stays True even if everything was undone */
      change = 1;
    }
    memcpy(v->bord, v->grid, (size_t)e->n_cells);
    v->bord_stale = 0;
  }
  return change;
}

/* returns change (0/1) */
static int base_update(const orc_engine *e, env_t *v, int action) {
  const orc_config *cfg = &e->cfg;
  int change = 0;
  switch (cfg->representation) {
    case ORC_REP_NARROW: { /* reps/narrow_rep.py:89-102 (Q1: coords[0] is edited twice) */
      int idx = ravel(cfg, v->pos);
      change = v->grid[idx] != (uint8_t)action;
      v->grid[idx] = (uint8_t)action;
      unravel(cfg, v->n_step % e->n_cells, v->pos);
      v->n_step++;
      break;
    }
    case ORC_REP_TURTLE: { /* reps/turtle_rep.py:87-107, _dirs :14 on (row, col), no wrap (:20) */
      static const int DR[4] = {-1, 1, 0, 0}, DC[4] = {0, 0, -1, 1};
      if (action < 4) {
        int r = v->pos[0] + DR[action], c = v->pos[1] + DC[action];
        if (r < 0) r = 0;
        if (r >= cfg->dims[0]) r = cfg->dims[0] - 1;
        if (c < 0) c = 0;
        if (c >= cfg->dims[1]) c = cfg->dims[1] - 1;
        v->pos[0] = r;
        v->pos[1] = c;
      } else {
        int idx = ravel(cfg, v->pos);
        uint8_t t = (uint8_t)(action - 4);
        change = v->grid[idx] != t;
        v->grid[idx] = t;
      }
      break;
    }
    case ORC_REP_WIDE: { /* wrappers.py:304-323 ActionMap.step + reps/wide_rep.py:40-45 (Q4: transposed) */
      int h = cfg->dims[0], w = cfg->dims[1], dim = e->n_tiles;
      int y = action / (w * dim), x = (action / dim) % w, t = action % dim;
      (void)h;
      v->pos[0] = x; /* rep receives [x, y, v] and indexes map[x, y] */
      v->pos[1] = y;
      int idx = x * cfg->dims[1] + y;
      change = v->grid[idx] != (uint8_t)t;
      v->grid[idx] = (uint8_t)t;
      break;
    }
  }
  return change;
}

/* StaticTileRepresentation.reset (reps/wrappers.py:265-319), after the wrapped representation's reset.  All draws come
 * from the representation RNG (unwrapped._random). */
static void static_reset(const orc_engine *e, env_t *v, int injected) {
  const orc_config *cfg = &e->cfg;
  const int H = cfg->dims[0], W = cfg->dims[1], BW = W + 2, BH = H + 2;
  memset(v->static_b, 0, (size_t)BH * BW);
  memcpy(v->bord, v->grid, (size_t)e->n_cells); /* representation.py:65-76: _update_bordered_map at the end of reset */
  v->bord_stale = 0;
  if (injected) { /* engine extension (set_map): no draws, border only */
  } else {
    if (cfg->static_prob > 0) {
      double ps = cfg->static_eval ? cfg->static_prob : pcg64_double(&v->rng_rep) * cfg->static_prob; /* :269-274 */
      for (int i = 0; i < BH * BW; i++) v->static_b[i] = pcg64_double(&v->rng_rep) < ps;              /* :278 */
    }
    for (int n = 0; n < cfg->n_static_walls; n++) { /* :280-299 */
      const int shape[2] = {H, W};
      int wshape[2] = {1, 1}, wpos[2] = {0, 0};
      int dim = (int)pcg64_integers(&v->rng_rep, 0, 2);
      int wall_len = (int)pcg64_integers(&v->rng_rep, 1, shape[dim] - 1);
      wshape[dim] = wall_len;
      for (int d = 0; d < 2; d++) /* :292 draws from shape[dim] for the OTHER axis as well */
        if (d != dim) wpos[d] = (int)pcg64_integers(&v->rng_rep, 0, shape[dim]);
      wpos[dim] = (int)pcg64_integers(&v->rng_rep, 0, shape[dim] - wall_len);
      wpos[0] += 1; /* :295 "shift to account for border" -- then used on the UNbordered _map too (:298) */
      wpos[1] += 1;
      for (int r = wpos[0]; r < wpos[0] + wshape[0]; r++)
        for (int c = wpos[1]; c < wpos[1] + wshape[1]; c++) {
          if (r < H && c < W) v->grid[r * W + c] = 1; /* _wall_tile = tiles[1] (probs/problem.py:41); slices clip */
          if (r < BH && c < BW) v->static_b[r * BW + c] = 1;
        }
      v->bord_stale = 1;
    }
  }
  for (int c = 0; c < BW; c++) v->static_b[c] = v->static_b[(BH - 1) * BW + c] = 1; /* :302-303 */
  for (int r = 0; r < BH; r++) v->static_b[r * BW] = v->static_b[r * BW + BW - 1] = 1;
}

/* envs/pcgrl_env.py:158-188 reset + reps/representation.py:65-76 + helper.py:491-494, :527-536 +
 * reps/narrow_rep.py:41-51 / turtle_rep.py:31-44 + control_wrappers.py:174-187. */
static void env_reset(const orc_engine *e, env_t *v, const uint8_t *init_grid, const int32_t *init_pos) {
  const orc_config *cfg = &e->cfg;
  int nt = e->n_tiles;
  if (v->has_pending) { /* control_wrappers.py:174-178: queued targets take effect at reset */
    for (int k = 0; k < cfg->n_ctrl; k++) {
      int s = cfg->ctrl_idx[k];
      v->trg_lo[s] = v->pend_lo[s];
      v->trg_hi[s] = v->pend_hi[s];
    }
    v->has_pending = 0;
  }
  v->changes = 0;
  v->iteration = 0;
  v->n_step = 0;
  v->path_len = 0;
  if (init_grid) {
    memcpy(v->grid, init_grid, (size_t)e->n_cells);
    v->pos[0] = v->pos[1] = v->pos[2] = 0;
    if (init_pos && cfg->representation != ORC_REP_WIDE)
      for (int d = 0; d < cfg->ndim; d++) v->pos[d] = init_pos[d];
  } else {
    double probs[ORC_MAX_TILES], cdf[ORC_MAX_TILES], total = 0.0;
    for (int t = 0; t < nt; t++) probs[t] = pcg64_double(&v->rng_prob); /* pcgrl_env.py:162-164 */
    for (int t = 0; t < nt; t++) total += probs[t];                      /* helper.py:527-536 */
    for (int t = 0; t < nt; t++) probs[t] /= total;
    v->pos[0] = v->pos[1] = v->pos[2] = 0;
    if (cfg->representation == ORC_REP_TURTLE) /* turtle_rep.py:31-44: position drawn BEFORE the map */
      for (int d = 0; d < cfg->ndim; d++) v->pos[d] = (int)(pcg64_double(&v->rng_rep) * cfg->dims[d]);
    /* Generator.choice(n, size=dims, p): cdf = cumsum(p); cdf /= cdf[-1];
     * idx = searchsorted(cdf, random(dims), side='right')  (numpy/random/_generator.pyx) */
    double acc = 0.0;
    for (int t = 0; t < nt; t++) {
      acc += probs[t];
      cdf[t] = acc;
    }
    for (int t = 0; t < nt; t++) cdf[t] /= acc;
    for (int i = 0; i < e->n_cells; i++) {
      double u = pcg64_double(&v->rng_rep);
      int idx = 0;
      while (idx < nt && cdf[idx] <= u) idx++; /* side='right': first cdf[idx] > u */
      v->grid[i] = (uint8_t)idx;
    }
  }
  if (cfg->act_window[0] > 0) { /* narrow_rep.py:41-51 with the wrapper's get_act_coords: first patch centre */
    v->pos[0] = (cfg->act_window[0] - 1) / 2;
    v->pos[1] = (cfg->act_window[1] - 1) / 2;
  }
  if (cfg->static_tiles) static_reset(e, v, init_grid != NULL);
  get_stats(cfg, v->grid, v->stats, v->path_xyz, &v->path_len);
  if (cfg->problem == ORC_PROB_MC3DMAZE) {
    /* the reset observation carries no path overlay: PcgrlEnv.reset() does not call
     * process_observation (pcgrl_env.py:180-188); the path found here shows from the first step on. */
  }
  if (!init_grid && cfg->problem == ORC_PROB_BINARY) (void)pcg64_double(&v->rng_prob); /* binary_prob.py:139-143 */
  v->last_loss = get_loss(cfg, v, v->stats);
  v->ep_return = 0.0;
  v->ep_len = 0;
}

/* ------------------------------------------------------------------------------------------------
 * Observation encoder (wrappers.py)
 * ---------------------------------------------------------------------------------------------- */
int64_t orc_obs_size(const orc_engine *e) {
  const orc_config *c = &e->cfg;
  if (c->representation == ORC_REP_WIDE) return (int64_t)e->n_cells * e->n_tiles;
  int64_t n = 1;
  for (int d = 0; d < c->ndim; d++) n *= c->obs_window[d];
  int chans = e->n_tiles + 1 + (c->problem == ORC_PROB_MC3DMAZE ? 1 : 0) + (c->static_tiles ? 1 : 0);
  return n * chans;
}

/* Cropped._transform wrappers.py:407-437 (map+1, zero pad floor(obs_window/2), window starts at pos)
 * -> OneHotEncoding._transform :232-257 (dim = n_tiles+1, channel 0 = out of bounds)
 * -> ToImage :140-150 (channel last).  Wide: ActionMap -> OneHot (dim = n_tiles) -> ToImage. */
static void encode_obs(const orc_engine *e, const env_t *v, uint8_t *out, int show_path) {
  const orc_config *c = &e->cfg;
  memset(out, 0, (size_t)orc_obs_size(e));
  if (c->representation == ORC_REP_WIDE) {
    for (int i = 0; i < e->n_cells; i++) out[(size_t)i * e->n_tiles + v->grid[i]] = 1;
    return;
  }
  if (c->ndim == 2) {
    int oh = c->obs_window[0], ow = c->obs_window[1], C = e->n_tiles + 1 + (c->static_tiles ? 1 : 0);
    for (int i = 0; i < oh; i++)
      for (int j = 0; j < ow; j++) {
        int r = v->pos[0] - oh / 2 + i, q = v->pos[1] - ow / 2 + j;
        int val = 0;
        if (r >= 0 && r < c->dims[0] && q >= 0 && q < c->dims[1]) val = v->grid[r * c->dims[1] + q] + 1;
        out[((size_t)i * ow + j) * C + val] = 1;
        /* wrappers.py:452-461: the static_builds plane goes through the same Cropped wrapper, but it has the
         * BORDERED shape while pos and pad are those of the map: the plane is read at bordered index (r, q), i.e. it
         * is shifted by one cell against the map channels; zero outside the bordered array */
        if (c->static_tiles && r >= 0 && r < c->dims[0] + 2 && q >= 0 && q < c->dims[1] + 2)
          out[((size_t)i * ow + j) * C + e->n_tiles + 1] = v->static_b[r * (c->dims[1] + 2) + q];
      }
  } else {
    /* 3-D: the reference's image wrappers raise on this problem; this follows their
     * intent: cropped window, channel 0 = OOB, 1..n = tiles, n+1 = path overlay written at the
     * reference's transposed [x][y][z] indices (minecraft_3D_maze_prob.py:84-93). */
    int o0 = c->obs_window[0], o1 = c->obs_window[1], o2 = c->obs_window[2], C = e->n_tiles + 2;
    uint8_t *m = (uint8_t *)malloc((size_t)e->n_cells);
    memcpy(m, v->grid, (size_t)e->n_cells);
    if (show_path)
      for (int k = 0; k < v->path_len; k++) {
        int x = v->path_xyz[3 * k], y = v->path_xyz[3 * k + 1], z = v->path_xyz[3 * k + 2];
        /* non-cubic maps: the reference would raise IndexError here; out-of-range tiles are skipped */
        if (x < c->dims[0] && y < c->dims[1] && z < c->dims[2]) m[(x * c->dims[1] + y) * c->dims[2] + z] = (uint8_t)e->n_tiles;
      }
    for (int i = 0; i < o0; i++)
      for (int j = 0; j < o1; j++)
        for (int k = 0; k < o2; k++) {
          int a = v->pos[0] - o0 / 2 + i, b = v->pos[1] - o1 / 2 + j, d = v->pos[2] - o2 / 2 + k;
          int val = 0;
          if (a >= 0 && a < c->dims[0] && b >= 0 && b < c->dims[1] && d >= 0 && d < c->dims[2])
            val = m[(a * c->dims[1] + b) * c->dims[2] + d] + 1;
          out[(((size_t)i * o1 + j) * o2 + k) * C + val] = 1;
        }
    free(m);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Public API
 * ---------------------------------------------------------------------------------------------- */
orc_engine *orc_create(const orc_config *cfg, int32_t n_envs) {
  orc_engine *e = (orc_engine *)calloc(1, sizeof(*e));
  e->cfg = *cfg;
  e->n_envs = n_envs;
  e->n_cells = cfg->dims[0] * cfg->dims[1] * (cfg->ndim == 3 ? cfg->dims[2] : 1);
  e->n_tiles = n_tiles_of(cfg->problem);
  e->n_threads = 1;
  e->n_act = cfg->act_window[0] > 0 ? cfg->act_window[0] * cfg->act_window[1] : 1;
  size_t static_sz = (size_t)(cfg->dims[0] + 2) * (cfg->dims[1] + 2) + (size_t)e->n_cells;
  e->static_pool = (uint8_t *)calloc((size_t)n_envs, static_sz);
  e->envs = (env_t *)calloc((size_t)n_envs, sizeof(env_t));
  e->grid_pool = (uint8_t *)calloc((size_t)n_envs, (size_t)e->n_cells);
  int path_cap = e->n_cells * 4 + 8;
  e->path_pool = (int16_t *)calloc((size_t)n_envs * path_cap * 3, sizeof(int16_t));
  for (int i = 0; i < n_envs; i++) {
    e->envs[i].grid = e->grid_pool + (size_t)i * e->n_cells;
    e->envs[i].path_xyz = e->path_pool + (size_t)i * path_cap * 3;
    e->envs[i].static_b = e->static_pool + (size_t)i * static_sz;
    e->envs[i].bord = e->envs[i].static_b + (size_t)(cfg->dims[0] + 2) * (cfg->dims[1] + 2);
    pcg64_seed(&e->envs[i].rng_rep, (uint64_t)i);
    pcg64_seed(&e->envs[i].rng_prob, (uint64_t)i);
    memcpy(e->envs[i].trg_lo, cfg->trg_lo, sizeof(cfg->trg_lo));
    memcpy(e->envs[i].trg_hi, cfg->trg_hi, sizeof(cfg->trg_hi));
  }
  return e;
}

void orc_destroy(orc_engine *e) {
  if (!e) return;
  free(e->envs);
  free(e->grid_pool);
  free(e->path_pool);
  free(e->static_pool);
  free(e);
}

void orc_set_threads(orc_engine *e, int32_t n) { e->n_threads = n < 1 ? 1 : n; }

void orc_seed(orc_engine *e, const uint64_t *seeds) {
  for (int i = 0; i < e->n_envs; i++) {
    pcg64_seed(&e->envs[i].rng_rep, seeds[i]);
    pcg64_seed(&e->envs[i].rng_prob, seeds[i]);
  }
}

void orc_reset(orc_engine *e, const uint8_t *mask, const uint8_t *init_grids, const int32_t *init_pos) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(e->n_threads)
  for (int i = 0; i < e->n_envs; i++) {
    if (mask && !mask[i]) continue;
    env_reset(e, &e->envs[i], init_grids ? init_grids + (size_t)i * e->n_cells : NULL,
              init_pos ? init_pos + (size_t)i * 3 : NULL);
  }
}

void orc_step(orc_engine *e, const int32_t *actions, int32_t auto_reset, uint8_t *obs, double *reward,
              uint8_t *done, int32_t *stats) {
  orc_step_masked(e, NULL, actions, auto_reset, obs, reward, done, stats);
}

/* The same for the envs selected by `mask` (NULL = all): the others are not stepped and their rows of the outputs are left
 * alone.  The reference's envs are independent objects, each stepped when its own worker steps it (rl/utils.py:412-415);
 * the tests of the engine's asynchronous stepping (ready mask) step an env here exactly when the engine says it stepped. */
void orc_step_masked(orc_engine *e, const uint8_t *mask, const int32_t *actions, int32_t auto_reset, uint8_t *obs,
                     double *reward, uint8_t *done, int32_t *stats) {
  const orc_config *cfg = &e->cfg;
  int64_t osz = orc_obs_size(e);
#pragma omp parallel for schedule(dynamic, 16) num_threads(e->n_threads)
  for (int i = 0; i < e->n_envs; i++) {
    if (mask != NULL && mask[i] == 0) continue;
    env_t *v = &e->envs[i];
    /* envs/pcgrl_env.py:267-342 */
    v->iteration++;
    int change = rep_update(e, v, actions + (size_t)i * e->n_act);
    int show_path = 1;
    if (obs && cfg->problem == ORC_PROB_MC3DMAZE) {
      /* the 3-D observation is assembled BEFORE the stats refresh (pcgrl_env.py:298-299 then :314-323),
       * so it carries the path of the previous stats update */
      encode_obs(e, v, obs + (size_t)i * osz, 1);
    }
    if (change > 0) {
      v->changes += change;
      get_stats(cfg, v->grid, v->stats, v->path_xyz, &v->path_len);
    }
    int d = v->iteration > cfg->max_iterations; /* :307 */
    if (cfg->max_changes >= 0) d = d || (v->changes > cfg->max_changes); /* :308-309 */
    /* control_wrappers.py:216-244 */
    double loss = get_loss(cfg, v, v->stats);
    double r = loss - v->last_loss;
    v->last_loss = loss;
    v->ep_return += r;
    v->ep_len++;
    if (reward) reward[i] = r;
    if (done) done[i] = (uint8_t)d;
    if (stats)
      for (int k = 0; k < cfg->n_stats; k++) stats[(size_t)i * cfg->n_stats + k] = v->stats[k];
    if (d && auto_reset) {
      v->last_ep_return = v->ep_return;
      v->last_ep_len = v->ep_len;
      memcpy(v->final_stats, v->stats, sizeof(v->stats));
      v->n_episodes++;
      env_reset(e, v, NULL, NULL);
      show_path = 0;
      if (obs && cfg->problem == ORC_PROB_MC3DMAZE) encode_obs(e, v, obs + (size_t)i * osz, 0);
    }
    if (obs && cfg->problem != ORC_PROB_MC3DMAZE) encode_obs(e, v, obs + (size_t)i * osz, show_path);
  }
}

void orc_update(orc_engine *e, const int32_t *actions, uint8_t *obs) {
  int64_t osz = orc_obs_size(e);
#pragma omp parallel for schedule(dynamic, 16) num_threads(e->n_threads)
  for (int i = 0; i < e->n_envs; i++) {
    env_t *v = &e->envs[i];
    (void)rep_update(e, v, actions + (size_t)i * e->n_act);
    if (obs) encode_obs(e, v, obs + (size_t)i * osz, 1);
  }
}

void orc_refresh_stats(orc_engine *e, int32_t *stats) {
  const orc_config *cfg = &e->cfg;
#pragma omp parallel for schedule(dynamic, 16) num_threads(e->n_threads)
  for (int i = 0; i < e->n_envs; i++) {
    env_t *v = &e->envs[i];
    get_stats(cfg, v->grid, v->stats, v->path_xyz, &v->path_len);
    v->last_loss = get_loss(cfg, v, v->stats);
    if (stats)
      for (int k = 0; k < cfg->n_stats; k++) stats[(size_t)i * cfg->n_stats + k] = v->stats[k];
  }
}

void orc_observe(orc_engine *e, uint8_t *obs) {
  int64_t osz = orc_obs_size(e);
  for (int i = 0; i < e->n_envs; i++) encode_obs(e, &e->envs[i], obs + (size_t)i * osz, 0);
}

/* set_static_prob / set_n_static_walls / set_eval_mode (reps/wrappers.py:256-263); negative = unchanged */
void orc_set_static(orc_engine *e, double static_prob, int32_t n_static_walls, int32_t eval_mode) {
  if (static_prob >= 0) e->cfg.static_prob = static_prob;
  if (n_static_walls >= 0) e->cfg.n_static_walls = n_static_walls;
  e->cfg.static_eval = eval_mode ? 1 : 0;
}

void orc_get_static(orc_engine *e, uint8_t *out) {
  size_t n = (size_t)(e->cfg.dims[0] + 2) * (e->cfg.dims[1] + 2);
  for (int i = 0; i < e->n_envs; i++) memcpy(out + (size_t)i * n, e->envs[i].static_b, n);
}

void orc_get_state(orc_engine *e, uint8_t *grids, int32_t *pos, int32_t *counters, int32_t *stats,
                   double *last_loss, double *ep_return) {
  for (int i = 0; i < e->n_envs; i++) {
    env_t *v = &e->envs[i];
    if (grids) memcpy(grids + (size_t)i * e->n_cells, v->grid, (size_t)e->n_cells);
    if (pos)
      for (int d = 0; d < 3; d++) pos[i * 3 + d] = v->pos[d];
    if (counters) {
      counters[i * 4 + 0] = v->iteration;
      counters[i * 4 + 1] = v->changes;
      counters[i * 4 + 2] = v->n_step;
      counters[i * 4 + 3] = v->ep_len;
    }
    if (stats)
      for (int k = 0; k < e->cfg.n_stats; k++) stats[(size_t)i * e->cfg.n_stats + k] = v->stats[k];
    if (last_loss) last_loss[i] = v->last_loss;
    if (ep_return) ep_return[i] = v->ep_return;
  }
}

void orc_queue_targets(orc_engine *e, const uint8_t *mask, const double *trg_lo, const double *trg_hi) {
  int S = e->cfg.n_stats;
  for (int i = 0; i < e->n_envs; i++) {
    if (mask && !mask[i]) continue;
    env_t *v = &e->envs[i];
    for (int k = 0; k < S; k++) {
      v->pend_lo[k] = trg_lo[(size_t)i * S + k];
      v->pend_hi[k] = trg_hi[(size_t)i * S + k];
    }
    v->has_pending = 1;
  }
}

void orc_get_ctrl_obs(orc_engine *e, double *out) {
  const orc_config *c = &e->cfg;
  for (int i = 0; i < e->n_envs; i++) {
    const env_t *v = &e->envs[i];
    for (int k = 0; k < c->n_ctrl; k++) {
      int s = c->ctrl_idx[k];
      double trg = (v->trg_lo[s] + v->trg_hi[s]) / 2; /* tuple target -> its midpoint (:203-204) */
      out[(size_t)i * 2 * c->n_ctrl + 2 * k] = trg / c->ctrl_range[k];
      out[(size_t)i * 2 * c->n_ctrl + 2 * k + 1] = (double)v->stats[s] / c->ctrl_range[k];
    }
  }
}

void orc_get_last_episode(orc_engine *e, double *ep_return, int32_t *ep_len, int32_t *final_stats,
                          int64_t *n_episodes) {
  for (int i = 0; i < e->n_envs; i++) {
    env_t *v = &e->envs[i];
    if (ep_return) ep_return[i] = v->last_ep_return;
    if (ep_len) ep_len[i] = v->last_ep_len;
    if (final_stats)
      for (int k = 0; k < e->cfg.n_stats; k++) final_stats[(size_t)i * e->cfg.n_stats + k] = v->final_stats[k];
    if (n_episodes) n_episodes[i] = v->n_episodes;
  }
}
