// pcgrl_sokoban.h -- device-side Sokoban solver cascade (rare path of sokoban's get_stats).
//
// Reference: envs/probs/sokoban/sokoban_prob.py:99-148 (_run_game: BFS, then A* with balance 1, 0.5, 0, each
// limited to `solver_power` iterations) and envs/probs/sokoban/sokoban/engine.py (Node :4-50, BFSAgent :56-74,
// AStarAgent :96-119, State :121-363).  It only runs when a map has exactly one player, as many crates as targets
// (> 0) and a single region (sokoban_prob.py:172-177) -- never under random actions, often under a trained policy.
//
// The search is inherently sequential and order dependent (FIFO queue; CPython heapq sift order with
// Node.__lt__ = h + balance*depth; visited keyed on player + ordered crate list), so one lane per env walks it,
// with its node pool / visited table / queue in an HBM workspace slot taken from a small lock-protected pool.
// The other lanes of the wavefront idle meanwhile; other wavefronts are unaffected.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "pcgrl_common.h"
#include "pcgrl_kernels2d.h"

namespace pcgrl {

constexpr int SK_MAXC = 128;     // crates (= targets) the device solver supports (a 16x16 map holds at most 127 pairs + player)
constexpr int SK_MAXDIM = 34;    // bordered level side (W+2, H+2 <= 34)
constexpr int SK_VCAP = 1 << 15; // visited table slots (>= 2 x iterations per stage)
constexpr int SK_MAX_POWER = SK_VCAP / 2;  // cfg.solver_power accepted by pcgrl_create

struct SokoLevel {
  int32_t w, h, ncr, ntg;
  int32_t use_order, n_dead_cur;  // heuristic via the per-cell target order table; crates of the expanded node on dead cells
  uint64_t solid[SK_MAXDIM], dead[SK_MAXDIM], tgt[SK_MAXDIM];
  uint64_t occ[SK_MAXDIM];        // crate occupancy of the node being expanded
  uint8_t tx[SK_MAXC], ty[SK_MAXC];
};
struct SokoNode {
  int32_t parent;
  int16_t depth, h;
  uint8_t px, py, pad[6];
};
static_assert(sizeof(SokoNode) == 16, "node header");

struct SokoPool {  // lives in Params-reachable global memory
  int32_t n_slots, max_nodes;
  size_t slot_bytes;
  uint8_t *base;
  int32_t *locks;   // [n_slots] 0 = free
  uint32_t *epochs; // [n_slots]
};

struct SokoCtx {
  SokoLevel *lv;
  SokoNode *nodes;
  uint8_t *crates;  // [max_nodes][2*SK_MAXC]
  uint32_t *vis;    // [SK_VCAP]  (epoch << 17) | (node + 1)
  int32_t *q;       // [max_nodes] BFS queue / A* heap / scratch
  uint8_t *order;   // [SK_MAXDIM * SK_MAXDIM][SK_MAXC] targets sorted by (Manhattan distance, index) per cell
  int32_t n_nodes, max_nodes, ncr;
  uint32_t epoch;
};

__device__ inline uint8_t *sk_crates(const SokoCtx &c, int n) { return c.crates + (size_t)n * (2 * SK_MAXC); }
__device__ inline bool sk_bit(const uint64_t *rows, int x, int y) { return (rows[y] >> x) & 1ull; }

__device__ inline int sk_crate_at(const SokoCtx &c, const uint8_t *cr, int x, int y) {  // engine.py:263-267
  for (int i = 0; i < c.ncr; i++)
    if (cr[2 * i] == x && cr[2 * i + 1] == y) return i;
  return -1;
}
__device__ inline bool sk_movable(const SokoCtx &c, const uint8_t *cr, int x, int y) {  // engine.py:254-255, :269-270
  if (x < 0 || y < 0 || x > c.lv->w - 1 || y > c.lv->h - 1) return false;
  if (sk_bit(c.lv->solid, x, y)) return false;
  return sk_crate_at(c, cr, x, y) < 0;
}
__device__ inline bool sk_win(const SokoCtx &c, const uint8_t *cr) {  // engine.py:272-280
  if (c.lv->ntg != c.ncr || c.ncr == 0) return false;
  for (int t = 0; t < c.lv->ntg; t++)
    if (sk_crate_at(c, cr, c.lv->tx[t], c.lv->ty[t]) < 0) return false;
  return true;
}
// engine.py:282-296 getHeuristic: crates in list order greedily take the nearest remaining target (first minimum in
// list order).  With many targets the scan over all targets per crate dominates the whole search, so levels with more
// than 8 targets use a per-cell table of the targets sorted by (distance, index): the nearest REMAINING target is the
// first unused entry of the crate cell's row -- the same choice, found without computing all distances.
__device__ inline void sk_build_order(SokoCtx &c) {
  SokoLevel *lv = c.lv;
  const int nt = lv->ntg, maxd = lv->w + lv->h;
  int32_t *cnt = c.q;  // scratch: maxd + 2 counters
  for (int y = 1; y < lv->h - 1; y++)
    for (int x = 1; x < lv->w - 1; x++) {
      uint8_t *row = c.order + (size_t)(y * SK_MAXDIM + x) * SK_MAXC;
      for (int d = 0; d <= maxd + 1; d++) cnt[d] = 0;
      for (int t = 0; t < nt; t++) cnt[abs(x - (int)lv->tx[t]) + abs(y - (int)lv->ty[t]) + 1]++;
      for (int d = 1; d <= maxd + 1; d++) cnt[d] += cnt[d - 1];
      for (int t = 0; t < nt; t++) {  // stable: equal distances keep index order
        int d = abs(x - (int)lv->tx[t]) + abs(y - (int)lv->ty[t]);
        row[cnt[d]++] = (uint8_t)t;
      }
    }
}

__device__ inline int sk_heuristic(const SokoCtx &c, const uint8_t *cr) {
  uint64_t used0 = 0, used1 = 0;  // targets already matched (the reference deletes them from a list: order is preserved)
  int distance = 0;
  const int nt = c.lv->ntg;
  if (c.lv->use_order) {
    for (int k = 0; k < c.ncr; k++) {
      const int cx = cr[2 * k], cy = cr[2 * k + 1];
      const uint8_t *row = c.order + (size_t)(cy * SK_MAXDIM + cx) * SK_MAXC;
      int r = 0, t = row[0];
      while (((t < 64 ? used0 : used1) >> (t & 63)) & 1ull) t = row[++r];  // ncr == ntg: an unused target always exists
      distance += abs((int)c.lv->tx[t] - cx) + abs((int)c.lv->ty[t] - cy);
      if (t < 64) used0 |= 1ull << t; else used1 |= 1ull << (t & 63);
    }
    return distance;
  }
  for (int k = 0; k < c.ncr; k++) {
    int best = c.lv->w + c.lv->h, match = -1, first_free = -1;
    for (int i = 0; i < nt; i++) {
      if (((i < 64 ? used0 : used1) >> (i & 63)) & 1ull) continue;
      if (first_free < 0) first_free = i;
      int d = abs((int)cr[2 * k] - (int)c.lv->tx[i]) + abs((int)cr[2 * k + 1] - (int)c.lv->ty[i]);
      if (best > d) {
        match = i;
        best = d;
      }
    }
    if (match < 0) match = first_free;  // bestMatch = 0 default: first remaining target
    distance += abs((int)c.lv->tx[match] - (int)cr[2 * k]) + abs((int)c.lv->ty[match] - (int)cr[2 * k + 1]);
    if (match < 64) used0 |= 1ull << match; else used1 |= 1ull << (match & 63);
  }
  return distance;
}
__device__ inline bool sk_deadlock(const SokoCtx &c, const uint8_t *cr) {  // engine.py:248-252
  for (int k = 0; k < c.ncr; k++)
    if (sk_bit(c.lv->dead, cr[2 * k], cr[2 * k + 1])) return true;
  return false;
}

// engine.py:203-246 intializeDeadlocks (corner list kept in the queue scratch)
__device__ inline void sk_init_deadlocks(SokoCtx &c) {
  SokoLevel *lv = c.lv;
  const int w = lv->w, h = lv->h;
  int nc = 0;
  for (int y = 0; y < h; y++) lv->dead[y] = 0;
#define SOL(x, y) sk_bit(lv->solid, (x), (y))
#define TGT(x, y) sk_bit(lv->tgt, (x), (y))
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1 || SOL(x, y)) continue;
      if ((SOL(x, y - 1) && SOL(x - 1, y)) || (SOL(x, y - 1) && SOL(x + 1, y)) || (SOL(x, y + 1) && SOL(x - 1, y)) ||
          (SOL(x, y + 1) && SOL(x + 1, y))) {
        if (!TGT(x, y)) {
          c.q[nc++] = x | (y << 8);
          lv->dead[y] |= 1ull << x;
        }
      }
    }
  for (int a = 0; a < nc; a++)
    for (int b = 0; b < nc; b++) {
      int ax = c.q[a] & 255, ay = c.q[a] >> 8, bx = c.q[b] & 255, by = c.q[b] >> 8;
      int dx = (ax > bx) - (ax < bx), dy = (ay > by) - (ay < by);
      if ((dx == 0 && dy == 0) || (dx != 0 && dy != 0)) continue;
      int x = bx, y = by;
      bool ok = true;
      if (dx != 0) {
        for (x += dx; x != ax; x += dx)
          if (TGT(x, y) || SOL(x, y) || (!SOL(x, y - 1) && !SOL(x, y + 1))) {
            ok = false;
            break;
          }
        if (ok)
          for (x = bx + dx; x != ax; x += dx) lv->dead[y] |= 1ull << x;
      } else {
        for (y += dy; y != ay; y += dy)
          if (TGT(x, y) || SOL(x, y) || (!SOL(x - 1, y) && !SOL(x + 1, y))) {
            ok = false;
            break;
          }
        if (ok)
          for (y = by + dy; y != ay; y += dy) lv->dead[y] |= 1ull << x;
      }
    }
#undef SOL
#undef TGT
}

__device__ inline uint32_t sk_hash(const SokoCtx &c, int n) {
  const uint8_t *cr = sk_crates(c, n);
  uint32_t h = 2166136261u;
  h = (h ^ c.nodes[n].px) * 16777619u;
  h = (h ^ c.nodes[n].py) * 16777619u;
  // 8 bytes at a time: the crate lists are 256-byte aligned and zero-padded to a multiple of 8 bytes (root: see
  // sokoban_solve; children copy whole words), so equal keys hash and compare equal word by word
  const uint64_t *w = (const uint64_t *)cr;
  for (int i = 0; i < (2 * c.ncr + 7) / 8; i++) {
    const uint64_t v = w[i];
    h = (h ^ (uint32_t)v) * 16777619u;
    h = (h ^ (uint32_t)(v >> 32)) * 16777619u;
  }
  return h;
}
__device__ inline bool sk_same_key(const SokoCtx &c, int a, int b) {  // State.getKey engine.py:330-336
  if (c.nodes[a].px != c.nodes[b].px || c.nodes[a].py != c.nodes[b].py) return false;
  const uint64_t *ca = (const uint64_t *)sk_crates(c, a), *cb = (const uint64_t *)sk_crates(c, b);
  for (int i = 0; i < (2 * c.ncr + 7) / 8; i++)
    if (ca[i] != cb[i]) return false;
  return true;
}
// returns true if node n's key was already in the visited set; inserts it otherwise
__device__ inline bool sk_visited_test_and_set(SokoCtx &c, int n) {
  uint32_t i = sk_hash(c, n) & (SK_VCAP - 1);
  while (true) {
    uint32_t e = c.vis[i];
    if ((e >> 17) != c.epoch || (e & 0x1FFFFu) == 0) {
      c.vis[i] = (c.epoch << 17) | (uint32_t)(n + 1);
      return false;
    }
    if (sk_same_key(c, (int)(e & 0x1FFFFu) - 1, n)) return true;
    i = (i + 1) & (SK_VCAP - 1);
  }
}

// crate occupancy of node n as row bit masks, and how many of its crates stand on dead cells
__device__ inline void sk_load_occupancy(SokoCtx &c, int n) {
  SokoLevel *lv = c.lv;
  for (int y = 0; y < lv->h; y++) lv->occ[y] = 0;
  const uint8_t *cr = sk_crates(c, n);
  int nd = 0;
  for (int k = 0; k < c.ncr; k++) {
    lv->occ[cr[2 * k + 1]] |= 1ull << cr[2 * k];
    nd += sk_bit(lv->dead, cr[2 * k], cr[2 * k + 1]) ? 1 : 0;
  }
  lv->n_dead_cur = nd;
}
__device__ inline bool sk_win_occ(const SokoCtx &c) {  // engine.py:272-280 on the occupancy masks
  if (c.lv->ntg != c.ncr || c.ncr == 0) return false;
  for (int y = 0; y < c.lv->h; y++)
    if (c.lv->tgt[y] & ~c.lv->occ[y]) return false;
  return true;
}

// Node.getChildren engine.py:14-25 + State.update :298-328 (the expanded node's occupancy is loaded)
__device__ inline int sk_children(SokoCtx &c, int n, int *out) {
  const int DX[4] = {-1, 1, 0, 0}, DY[4] = {0, 0, -1, 1};  // engine.py:3
  const SokoLevel *lv = c.lv;
  int cnt = 0;
  const uint8_t *cr = sk_crates(c, n);
  const int px = c.nodes[n].px, py = c.nodes[n].py;
  auto free_cell = [&](int x, int y) {  // checkMovableLocation :269-270
    if (x < 0 || y < 0 || x > lv->w - 1 || y > lv->h - 1) return false;
    return !sk_bit(lv->solid, x, y) && !sk_bit(lv->occ, x, y);
  };
  for (int d = 0; d < 4; d++) {
    int nx = px + DX[d], ny = py + DY[d];
    int moved = -1;
    if (!free_cell(nx, ny)) {
      if (nx < 0 || ny < 0 || nx > lv->w - 1 || ny > lv->h - 1 || !sk_bit(lv->occ, nx, ny)) continue;
      if (!free_cell(nx + DX[d], ny + DY[d])) continue;
      moved = sk_crate_at(c, cr, nx, ny);
    }
    if (c.n_nodes >= c.max_nodes) continue;  // cannot happen: <= 1 + 4 * iterations nodes per stage
    if (moved >= 0) {  // engine.py:22-23 checkDeadlock over all crates of the child
      int nd = lv->n_dead_cur - (sk_bit(lv->dead, nx, ny) ? 1 : 0) + (sk_bit(lv->dead, nx + DX[d], ny + DY[d]) ? 1 : 0);
      if (nd > 0) continue;
    }
    int k = c.n_nodes++;
    uint8_t *ncr = sk_crates(c, k);
    for (int i = 0; i < (2 * c.ncr + 7) / 8; i++) ((uint64_t *)ncr)[i] = ((const uint64_t *)cr)[i];
    if (moved >= 0) {
      ncr[2 * moved] = (uint8_t)(nx + DX[d]);
      ncr[2 * moved + 1] = (uint8_t)(ny + DY[d]);
    }
    c.nodes[k].parent = n;
    c.nodes[k].depth = (int16_t)(c.nodes[n].depth + 1);
    c.nodes[k].px = (uint8_t)nx;
    c.nodes[k].py = (uint8_t)ny;
    c.nodes[k].h = moved >= 0 ? (int16_t)sk_heuristic(c, ncr) : c.nodes[n].h;  // h depends on the crates only
    out[cnt++] = k;
  }
  return cnt;
}

__device__ inline bool sk_better(const SokoCtx &c, int cur, int best) {  // engine.py:66-69
  if (best < 0) return true;
  if (c.nodes[cur].h < c.nodes[best].h) return true;
  return c.nodes[cur].h == c.nodes[best].h && c.nodes[cur].depth < c.nodes[best].depth;
}

// stage: balance < 0 -> BFSAgent (engine.py:56-74); else AStarAgent with that balance (engine.py:96-119)
__device__ inline bool sk_stage(SokoCtx &c, const SokoPool &pool, int slot, double balance, int max_iter, int &res_h, int &res_depth,
                                bool *exhausted = nullptr) {
  c.epoch = (atomicAdd(&pool.epochs[slot], 1u) + 1u) & 0x7FFFu;
  if (c.epoch == 0) {  // wrapped: start over with a clean table
    for (int i = 0; i < SK_VCAP; i++) c.vis[i] = 0;
    c.epoch = (atomicAdd(&pool.epochs[slot], 1u) + 1u) & 0x7FFFu;
  }
  c.n_nodes = 1;  // node 0 = root, already filled by the caller
  int head = 0, tail = 0, best = -1, iters = 0;
  auto lt = [&](int a, int b) {  // Node.__lt__ engine.py:49-50
    return (double)c.nodes[a].h + balance * (double)c.nodes[a].depth < (double)c.nodes[b].h + balance * (double)c.nodes[b].depth;
  };
  auto siftdown = [&](int startpos, int pos) {  // heapq._siftdown
    int item = c.q[pos];
    while (pos > startpos) {
      int pp = (pos - 1) >> 1, parent = c.q[pp];
      if (lt(item, parent)) {
        c.q[pos] = parent;
        pos = pp;
        continue;
      }
      break;
    }
    c.q[pos] = item;
  };
  c.q[tail++] = 0;
  while (iters < max_iter && head < tail) {
    iters++;
    int cur;
    if (balance < 0) {
      cur = c.q[head++];  // queue.pop(0)
    } else {              // heapq.heappop
      int last = c.q[--tail];
      if (tail > 0) {
        cur = c.q[0];
        c.q[0] = last;
        int pos = 0, child = 1;
        while (child < tail) {
          int right = child + 1;
          if (right < tail && !lt(c.q[child], c.q[right])) child = right;
          c.q[pos] = c.q[child];
          pos = child;
          child = 2 * pos + 1;
        }
        c.q[pos] = last;
        siftdown(0, pos);
      } else {
        cur = last;
      }
    }
    sk_load_occupancy(c, cur);
    if (sk_win_occ(c)) {
      res_h = c.nodes[cur].h;
      res_depth = c.nodes[cur].depth;
      return true;
    }
    if (!sk_visited_test_and_set(c, cur)) {
      if (sk_better(c, cur, best)) best = cur;
      int ch[4];
      int nc = sk_children(c, cur, ch);
      for (int i = 0; i < nc; i++) {
        c.q[tail++] = ch[i];
        if (balance >= 0) siftdown(0, tail - 1);  // heapq.heappush
      }
    }
  }
  res_h = c.nodes[best].h;
  res_depth = c.nodes[best].depth;
  if (exhausted) *exhausted = head >= tail;  // the open list ran dry: every reachable state was expanded
  return false;
}

// Called by every lane of the wave in uniform control flow; `need` is uniform per group.  Groups that need the
// solver are served one after the other, so a wavefront holds at most one workspace slot at a time and never waits
// for a slot while holding one (no lock cycles however many envs need solving at once).
template <int LPE>
__device__ void sokoban_solve(const Grp<LPE> &g, const Params &p, int env, bool need, uint32_t solid, uint32_t player,
                              uint32_t crate, uint32_t target, int &dist_win, int &sol_len) {
  (void)env;
  const SokoPool &pool = *(const SokoPool *)p.soko;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  constexpr int EPW = 64 / LPE;
  // one simulate wave per workgroup calls the solver, and it serves its groups one after the other
  __shared__ SokoLevel s_level;
  for (int gi = 0; gi < EPW; gi++) {
    const bool mine = need && (g.lane / LPE) == gi;
    if (__ballot(mine) == 0) continue;
    const bool leader = mine && g.row == 0;
    int slot = -1;
    SokoCtx c;
    c.lv = nullptr;
    if (leader) {
      int s = (int)((blockIdx.x * 7u + gi) % (unsigned)pool.n_slots);
      while (atomicCAS(&pool.locks[s], 0, 1) != 0) {
        s = (s + 1) % pool.n_slots;
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      slot = s;
      uint8_t *b = pool.base + (size_t)slot * pool.slot_bytes;
      c.max_nodes = pool.max_nodes;
      c.lv = &s_level;  // the level description (wall / dead / target / occupancy rows) is the hottest data: LDS
      b += (sizeof(SokoLevel) + 15) & ~(size_t)15;
      c.nodes = (SokoNode *)b;
      b += sizeof(SokoNode) * (size_t)c.max_nodes;
      c.crates = b;
      b += (size_t)c.max_nodes * 2 * SK_MAXC;
      c.vis = (uint32_t *)b;
      b += sizeof(uint32_t) * SK_VCAP;
      c.q = (int32_t *)b;
      b += sizeof(int32_t) * (size_t)c.max_nodes;
      c.order = b;
      c.lv->w = W + 2;
      c.lv->h = H + 2;
      const uint64_t full = (1ull << (W + 2)) - 1ull;
      c.lv->solid[0] = full;  // sokoban_prob.py:107-124: one-tile solid border around the map
      c.lv->solid[H + 1] = full;
      c.lv->tgt[0] = 0;
      c.lv->tgt[H + 1] = 0;
    }
    // gather the rows from their lanes (uniform control flow); level coords = map coords + 1
    int px = 0, py = 0, ncr = 0, ntg = 0;
    bool too_big = false;
    for (int r = 0; r < H; r++) {
      uint32_t sm = g.gbcast(solid, r), pl = g.gbcast(player, r), cr = g.gbcast(crate, r), tg = g.gbcast(target, r);
      if (leader) {
        c.lv->solid[r + 1] = ((uint64_t)sm << 1) | 1ull | (1ull << (W + 1));
        c.lv->tgt[r + 1] = (uint64_t)tg << 1;
        if (pl) {
          px = __builtin_ctz(pl) + 1;
          py = r + 1;
        }
        while (cr) {  // crates / targets are listed in row-major order (engine.py:170-188)
          int x = __builtin_ctz(cr);
          cr &= cr - 1;
          if (ncr < SK_MAXC) {
            c.crates[2 * ncr] = (uint8_t)(x + 1);  // node 0 = root
            c.crates[2 * ncr + 1] = (uint8_t)(r + 1);
          } else {
            too_big = true;
          }
          ncr++;
        }
        while (tg) {
          int x = __builtin_ctz(tg);
          tg &= tg - 1;
          if (ntg < SK_MAXC) {
            c.lv->tx[ntg] = (uint8_t)(x + 1);
            c.lv->ty[ntg] = (uint8_t)(r + 1);
          }
          ntg++;
        }
      }
    }
    int dw = dist_win, sl = sol_len;
    if (leader) {
      if (too_big || W + 2 > SK_MAXDIM || H + 2 > SK_MAXDIM) {
        atomicOr(p.err, 2);  // beyond the device solver's limits: reported by pcgrl_poll_error
      } else {
        c.ncr = ncr;
        for (int i = 2 * ncr; i < ((2 * ncr + 7) / 8) * 8; i++) c.crates[i] = 0;  // zero padding of the root's crate list
        c.lv->ncr = ncr;
        c.lv->ntg = ntg;
        sk_init_deadlocks(c);
        c.lv->use_order = ntg > 8 ? 1 : 0;
        if (c.lv->use_order) sk_build_order(c);
        c.nodes[0].parent = -1;
        c.nodes[0].depth = 0;
        c.nodes[0].px = (uint8_t)px;
        c.nodes[0].py = (uint8_t)py;
        c.nodes[0].h = (int16_t)sk_heuristic(c, sk_crates(c, 0));
        int h = 0, depth = 0;
        const int power = p.cfg.solver_power;
        // If the BFS stage expands the whole reachable state space without finding a win, no stage can win, each A*
        // stage would expand exactly the same set of states (pushes = 1 + sum of children over unique states, whatever
        // the order) and end with bestNode.h = min h over that set -- which the BFS stage already holds.  Skipping the
        // three A* stages is therefore exact (pinned by tests/golden/stats_sokoban_solver.npz against the reference).
        bool exhausted = false;
        bool won = sk_stage(c, pool, slot, -1.0, power, h, depth, &exhausted);
        if (!won && !exhausted)
          won = sk_stage(c, pool, slot, 1.0, power, h, depth) || sk_stage(c, pool, slot, 0.5, power, h, depth) ||
                sk_stage(c, pool, slot, 0.0, power, h, depth);
        if (won) {
          dw = 0;
          sl = depth;
        } else {
          dw = h;  // heuristic of the last stage's best node (sokoban_prob.py:147)
          sl = 0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      atomicExch(&pool.locks[slot], 0);
    }
    // hand the leader's result to its group
    int bdw = (int)g.gbcast((uint32_t)dw, 0), bsl = (int)g.gbcast((uint32_t)sl, 0);
    if (mine) {
      dist_win = bdw;
      sol_len = bsl;
    }
  }
}

// ---------------------------------------------------------------------------------------------- host side
static inline hipError_t sokoban_alloc(Params &p, std::vector<void *> &allocs) {
  SokoPool pool;
  pool.n_slots = p.n_envs < 64 ? p.n_envs : 64;
  pool.max_nodes = 4 * (p.cfg.solver_power > 0 ? p.cfg.solver_power : 1) + 8;
  if (pool.max_nodes > 0x1FFFE) pool.max_nodes = 0x1FFFE;  // 17-bit node ids in the visited table
  size_t sz = (sizeof(SokoLevel) + 15) & ~(size_t)15;
  sz += sizeof(SokoNode) * (size_t)pool.max_nodes + (size_t)pool.max_nodes * 2 * SK_MAXC;
  sz += sizeof(uint32_t) * SK_VCAP + sizeof(int32_t) * (size_t)pool.max_nodes;
  sz += (size_t)SK_MAXDIM * SK_MAXDIM * SK_MAXC;  // per-cell target order table
  pool.slot_bytes = (sz + 255) & ~(size_t)255;
  hipError_t e;
  void *base = nullptr, *locks = nullptr, *epochs = nullptr, *dpool = nullptr;
  if ((e = hipMalloc(&base, pool.slot_bytes * pool.n_slots)) != hipSuccess) return e;
  allocs.push_back(base);
  if ((e = hipMemset(base, 0, pool.slot_bytes * pool.n_slots)) != hipSuccess) return e;
  if ((e = hipMalloc(&locks, sizeof(int32_t) * pool.n_slots)) != hipSuccess) return e;
  allocs.push_back(locks);
  if ((e = hipMemset(locks, 0, sizeof(int32_t) * pool.n_slots)) != hipSuccess) return e;
  if ((e = hipMalloc(&epochs, sizeof(uint32_t) * pool.n_slots)) != hipSuccess) return e;
  allocs.push_back(epochs);
  if ((e = hipMemset(epochs, 0, sizeof(uint32_t) * pool.n_slots)) != hipSuccess) return e;
  pool.base = (uint8_t *)base;
  pool.locks = (int32_t *)locks;
  pool.epochs = (uint32_t *)epochs;
  if ((e = hipMalloc(&dpool, sizeof(SokoPool))) != hipSuccess) return e;
  allocs.push_back(dpool);
  if ((e = hipMemcpy(dpool, &pool, sizeof(pool), hipMemcpyHostToDevice)) != hipSuccess) return e;
  p.soko = dpool;
  return hipSuccess;
}

}  // namespace pcgrl
