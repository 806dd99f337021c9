/*
 * sokoban_solver.c -- restatement of the reference's Sokoban solver cascade on integer state.
 *
 * TEST INFRASTRUCTURE ONLY (part of the oracle, see pcgrl_oracle.h).
 *
 * Follows envs/probs/sokoban/sokoban_prob.py:99-148 (_run_game: BFS, then A* with balance 1, 0.5, 0,
 * each limited to `power` iterations) and envs/probs/sokoban/sokoban/engine.py
 * (Node :4-50, BFSAgent :56-74, AStarAgent :96-119, State :121-363).  The A* open list reproduces
 * CPython's heapq sift order because the result depends on it (queue.PriorityQueue -> heapq,
 * Node.__lt__ engine.py:49-50).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SK_MAXW 66

typedef struct {
  int w, h;                   /* bordered level size (W+2, H+2) */
  uint8_t solid[SK_MAXW * SK_MAXW];
  uint8_t dead[SK_MAXW * SK_MAXW];
  uint8_t is_target[SK_MAXW * SK_MAXW];
  int ntg, ncr;
  uint8_t tx[4096], ty[4096];
} level_t;

typedef struct {
  int32_t parent;
  int32_t depth;
  int32_t h;
  uint8_t px, py;
} node_t;

typedef struct {
  const level_t *lv;
  node_t *nodes;
  uint8_t *crates; /* nodes x ncr x 2 */
  int32_t n_nodes, cap;
} pool_t;

static void pool_init(pool_t *p, const level_t *lv) {
  p->lv = lv;
  p->cap = 1024;
  p->n_nodes = 0;
  p->nodes = (node_t *)malloc(sizeof(node_t) * (size_t)p->cap);
  p->crates = (uint8_t *)malloc((size_t)p->cap * (size_t)lv->ncr * 2);
}
static void pool_free(pool_t *p) {
  free(p->nodes);
  free(p->crates);
}
static int pool_new(pool_t *p) {
  if (p->n_nodes == p->cap) {
    p->cap *= 2;
    p->nodes = (node_t *)realloc(p->nodes, sizeof(node_t) * (size_t)p->cap);
    p->crates = (uint8_t *)realloc(p->crates, (size_t)p->cap * (size_t)p->lv->ncr * 2);
  }
  return p->n_nodes++;
}
static inline uint8_t *crates_of(pool_t *p, int n) { return p->crates + (size_t)n * p->lv->ncr * 2; }

static int crate_at(const level_t *lv, const uint8_t *cr, int x, int y) { /* engine.py:263-267 */
  for (int i = 0; i < lv->ncr; i++)
    if (cr[2 * i] == x && cr[2 * i + 1] == y) return i;
  return -1;
}
static int movable(const level_t *lv, const uint8_t *cr, int x, int y) { /* engine.py:254-255, :269-270 */
  if (x < 0 || y < 0 || x > lv->w - 1 || y > lv->h - 1) return 0;
  if (lv->solid[y * lv->w + x]) return 0;
  return crate_at(lv, cr, x, y) < 0;
}
static int check_win(const level_t *lv, const uint8_t *cr) { /* engine.py:272-280 */
  if (lv->ntg != lv->ncr || lv->ntg == 0) return 0;
  for (int t = 0; t < lv->ntg; t++)
    if (crate_at(lv, cr, lv->tx[t], lv->ty[t]) < 0) return 0;
  return 1;
}
static int heuristic(const level_t *lv, const uint8_t *cr) { /* engine.py:282-296 */
  uint8_t tx[4096], ty[4096];
  int nt = lv->ntg, distance = 0;
  memcpy(tx, lv->tx, (size_t)nt);
  memcpy(ty, lv->ty, (size_t)nt);
  for (int c = 0; c < lv->ncr; c++) {
    int best = lv->w + lv->h, match = 0;
    for (int i = 0; i < nt; i++) {
      int d = abs((int)cr[2 * c] - tx[i]) + abs((int)cr[2 * c + 1] - ty[i]);
      if (best > d) {
        match = i;
        best = d;
      }
    }
    distance += abs((int)tx[match] - cr[2 * c]) + abs((int)ty[match] - cr[2 * c + 1]);
    for (int i = match; i + 1 < nt; i++) {
      tx[i] = tx[i + 1];
      ty[i] = ty[i + 1];
    }
    nt--;
  }
  return distance;
}
static int check_deadlock(const level_t *lv, const uint8_t *cr) { /* engine.py:248-252 */
  for (int c = 0; c < lv->ncr; c++)
    if (lv->dead[cr[2 * c + 1] * lv->w + cr[2 * c]]) return 1;
  return 0;
}

static int sgn(int v) { return (v > 0) - (v < 0); }

/* engine.py:203-246 intializeDeadlocks */
static void init_deadlocks(level_t *lv) {
  int w = lv->w, h = lv->h, nc = 0;
  static __thread int16_t cx[SK_MAXW * SK_MAXW], cy[SK_MAXW * SK_MAXW];
  memset(lv->dead, 0, sizeof(lv->dead));
#define SOL(x, y) (lv->solid[(y)*w + (x)])
#define TGT(x, y) (lv->is_target[(y)*w + (x)])
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1 || SOL(x, y)) continue;
      if ((SOL(x, y - 1) && SOL(x - 1, y)) || (SOL(x, y - 1) && SOL(x + 1, y)) || (SOL(x, y + 1) && SOL(x - 1, y)) ||
          (SOL(x, y + 1) && SOL(x + 1, y))) {
        if (!TGT(x, y)) {
          cx[nc] = (int16_t)x;
          cy[nc] = (int16_t)y;
          nc++;
          lv->dead[y * w + x] = 1;
        }
      }
    }
  for (int a = 0; a < nc; a++)
    for (int b = 0; b < nc; b++) {
      int dx = sgn(cx[a] - cx[b]), dy = sgn(cy[a] - cy[b]);
      if ((dx == 0 && dy == 0) || (dx != 0 && dy != 0)) continue;
      int x = cx[b], y = cy[b], ok = 1;
      if (dx != 0) {
        for (x += dx; x != cx[a]; x += dx)
          if (TGT(x, y) || SOL(x, y) || (!SOL(x, y - 1) && !SOL(x, y + 1))) {
            ok = 0;
            break;
          }
        if (ok)
          for (x = cx[b] + dx; x != cx[a]; x += dx) lv->dead[y * w + x] = 1;
      } else {
        for (y += dy; y != cy[a]; y += dy)
          if (TGT(x, y) || SOL(x, y) || (!SOL(x - 1, y) && !SOL(x + 1, y))) {
            ok = 0;
            break;
          }
        if (ok)
          for (y = cy[b] + dy; y != cy[a]; y += dy) lv->dead[y * w + x] = 1;
      }
    }
#undef SOL
#undef TGT
}

/* visited set keyed like State.getKey (engine.py:330-336): player, then crates in list order */
typedef struct {
  uint8_t *keys;
  uint8_t *used;
  int cap, klen, count;
} vset_t;
static void vset_init(vset_t *s, int klen) {
  s->cap = 1 << 12;
  s->klen = klen;
  s->count = 0;
  s->keys = (uint8_t *)malloc((size_t)s->cap * klen);
  s->used = (uint8_t *)calloc((size_t)s->cap, 1);
}
static void vset_free(vset_t *s) {
  free(s->keys);
  free(s->used);
}
static uint32_t khash(const uint8_t *k, int n) {
  uint32_t h = 2166136261u;
  for (int i = 0; i < n; i++) h = (h ^ k[i]) * 16777619u;
  return h;
}
static int vset_insert(vset_t *s, const uint8_t *key); /* returns 1 if newly inserted */
static void vset_grow(vset_t *s) {
  vset_t n = *s;
  n.cap = s->cap * 2;
  n.count = 0;
  n.keys = (uint8_t *)malloc((size_t)n.cap * s->klen);
  n.used = (uint8_t *)calloc((size_t)n.cap, 1);
  for (int i = 0; i < s->cap; i++)
    if (s->used[i]) vset_insert(&n, s->keys + (size_t)i * s->klen);
  vset_free(s);
  *s = n;
}
static int vset_insert(vset_t *s, const uint8_t *key) {
  if (s->count * 2 >= s->cap) vset_grow(s);
  uint32_t i = khash(key, s->klen) & (uint32_t)(s->cap - 1);
  while (s->used[i]) {
    if (!memcmp(s->keys + (size_t)i * s->klen, key, (size_t)s->klen)) return 0;
    i = (i + 1) & (uint32_t)(s->cap - 1);
  }
  s->used[i] = 1;
  memcpy(s->keys + (size_t)i * s->klen, key, (size_t)s->klen);
  s->count++;
  return 1;
}
static int vset_contains(const vset_t *s, const uint8_t *key) {
  uint32_t i = khash(key, s->klen) & (uint32_t)(s->cap - 1);
  while (s->used[i]) {
    if (!memcmp(s->keys + (size_t)i * s->klen, key, (size_t)s->klen)) return 1;
    i = (i + 1) & (uint32_t)(s->cap - 1);
  }
  return 0;
}

static void make_key(pool_t *p, int n, uint8_t *key) {
  key[0] = p->nodes[n].px;
  key[1] = p->nodes[n].py;
  memcpy(key + 2, crates_of(p, n), (size_t)p->lv->ncr * 2);
}

/* Node.getChildren engine.py:14-25 + State.update :298-328. Appends children ids to out[], returns count. */
static int get_children(pool_t *p, int n, int *out) {
  static const int DX[4] = {-1, 1, 0, 0}, DY[4] = {0, 0, -1, 1}; /* engine.py:3 */
  const level_t *lv = p->lv;
  int cnt = 0;
  if (check_win(lv, crates_of(p, n))) return 0; /* update() returns early -> player never moves (:301-302) */
  for (int d = 0; d < 4; d++) {
    const uint8_t *cr = crates_of(p, n);
    int px = p->nodes[n].px, py = p->nodes[n].py;
    int nx = px + DX[d], ny = py + DY[d];
    int moved_crate = -1;
    if (movable(lv, cr, nx, ny)) {
      /* plain move */
    } else {
      int c = crate_at(lv, cr, nx, ny);
      if (c < 0) continue;
      int cx = nx + DX[d], cy = ny + DY[d];
      if (!movable(lv, cr, cx, cy)) continue;
      moved_crate = c;
    }
    int k = pool_new(p);
    uint8_t *ncr = crates_of(p, k);
    memcpy(ncr, crates_of(p, n), (size_t)lv->ncr * 2);
    if (moved_crate >= 0) {
      ncr[2 * moved_crate] = (uint8_t)(nx + DX[d]);
      ncr[2 * moved_crate + 1] = (uint8_t)(ny + DY[d]);
      if (check_deadlock(lv, ncr)) { /* engine.py:22-23 */
        p->n_nodes--;
        continue;
      }
    }
    p->nodes[k].parent = n;
    p->nodes[k].depth = p->nodes[n].depth + 1;
    p->nodes[k].px = (uint8_t)nx;
    p->nodes[k].py = (uint8_t)ny;
    p->nodes[k].h = heuristic(lv, ncr);
    out[cnt++] = k;
  }
  return cnt;
}

static int make_root(pool_t *p, int px, int py, const uint8_t *crates) {
  int r = pool_new(p);
  p->nodes[r].parent = -1;
  p->nodes[r].depth = 0;
  p->nodes[r].px = (uint8_t)px;
  p->nodes[r].py = (uint8_t)py;
  memcpy(crates_of(p, r), crates, (size_t)p->lv->ncr * 2);
  p->nodes[r].h = heuristic(p->lv, crates);
  return r;
}

static __thread int dbg_iters, dbg_exhausted;

#define BETTER(p, cur, best) \
  ((best) < 0 || (p)->nodes[cur].h < (p)->nodes[best].h || ((p)->nodes[cur].h == (p)->nodes[best].h && (p)->nodes[cur].depth < (p)->nodes[best].depth))

/* BFSAgent.getSolution engine.py:56-74. returns 1 on win; res_h / res_depth describe the returned node */
static int solve_bfs(const level_t *lv, int px, int py, const uint8_t *crates, int max_iter, int *res_h, int *res_depth) {
  pool_t p;
  vset_t vs;
  pool_init(&p, lv);
  vset_init(&vs, 2 + 2 * lv->ncr);
  int qcap = 4096, head = 0, tail = 0, best = -1, iters = 0, won = 0;
  int *q = (int *)malloc(sizeof(int) * (size_t)qcap);
  uint8_t *key = (uint8_t *)malloc((size_t)vs.klen);
  q[tail++] = make_root(&p, px, py, crates);
  while (iters < max_iter && head < tail) {
    iters++;
    int cur = q[head++];
    if (check_win(lv, crates_of(&p, cur))) {
      *res_h = p.nodes[cur].h;
      *res_depth = p.nodes[cur].depth;
      won = 1;
      goto done;
    }
    make_key(&p, cur, key);
    if (!vset_contains(&vs, key)) {
      if (BETTER(&p, cur, best)) best = cur;
      vset_insert(&vs, key);
      int ch[4];
      int nc = get_children(&p, cur, ch);
      if (tail + nc > qcap) {
        qcap *= 2;
        q = (int *)realloc(q, sizeof(int) * (size_t)qcap);
      }
      for (int i = 0; i < nc; i++) q[tail++] = ch[i];
    }
  }
  *res_h = p.nodes[best].h;
  *res_depth = p.nodes[best].depth;
done:
  dbg_iters = iters;
  dbg_exhausted = !won && head >= tail;
  free(q);
  free(key);
  vset_free(&vs);
  pool_free(&p);
  return won;
}

/* Node.__lt__ engine.py:49-50 with the class-level balance */
static inline int node_lt(const pool_t *p, int a, int b, double balance) {
  return (double)p->nodes[a].h + balance * (double)p->nodes[a].depth < (double)p->nodes[b].h + balance * (double)p->nodes[b].depth;
}

/* AStarAgent.getSolution engine.py:96-119; open list = CPython heapq (Lib/heapq.py heappush/heappop). */
static int solve_astar(const level_t *lv, int px, int py, const uint8_t *crates, double balance, int max_iter, int *res_h, int *res_depth) {
  pool_t p;
  vset_t vs;
  pool_init(&p, lv);
  vset_init(&vs, 2 + 2 * lv->ncr);
  int hcap = 4096, hn = 0, best = -1, iters = 0, won = 0;
  int *heap = (int *)malloc(sizeof(int) * (size_t)hcap);
  uint8_t *key = (uint8_t *)malloc((size_t)vs.klen);
#define SIFTDOWN(startpos, pos0)                     \
  do {                                               \
    int sd_pos = (pos0), sd_item = heap[sd_pos];     \
    while (sd_pos > (startpos)) {                    \
      int sd_pp = (sd_pos - 1) >> 1;                 \
      int sd_parent = heap[sd_pp];                   \
      if (node_lt(&p, sd_item, sd_parent, balance)) {\
        heap[sd_pos] = sd_parent;                    \
        sd_pos = sd_pp;                              \
        continue;                                    \
      }                                              \
      break;                                         \
    }                                                \
    heap[sd_pos] = sd_item;                          \
  } while (0)
#define HEAPPUSH(item)                                               \
  do {                                                               \
    if (hn == hcap) {                                                \
      hcap *= 2;                                                     \
      heap = (int *)realloc(heap, sizeof(int) * (size_t)hcap);       \
    }                                                                \
    heap[hn++] = (item);                                             \
    SIFTDOWN(0, hn - 1);                                             \
  } while (0)
  HEAPPUSH(make_root(&p, px, py, crates));
  while (iters < max_iter && hn > 0) {
    iters++;
    /* heappop */
    int last = heap[--hn], cur;
    if (hn > 0) {
      cur = heap[0];
      heap[0] = last;
      int pos = 0, endpos = hn, newitem = heap[0], childpos = 1;
      while (childpos < endpos) {
        int rightpos = childpos + 1;
        if (rightpos < endpos && !node_lt(&p, heap[childpos], heap[rightpos], balance)) childpos = rightpos;
        heap[pos] = heap[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
      }
      heap[pos] = newitem;
      SIFTDOWN(0, pos);
    } else {
      cur = last;
    }
    if (check_win(lv, crates_of(&p, cur))) {
      *res_h = p.nodes[cur].h;
      *res_depth = p.nodes[cur].depth;
      won = 1;
      goto done;
    }
    make_key(&p, cur, key);
    if (!vset_contains(&vs, key)) {
      if (BETTER(&p, cur, best)) best = cur;
      vset_insert(&vs, key);
      int ch[4];
      int nc = get_children(&p, cur, ch);
      for (int i = 0; i < nc; i++) HEAPPUSH(ch[i]);
    }
  }
  *res_h = p.nodes[best].h;
  *res_depth = p.nodes[best].depth;
done:
  dbg_iters = iters;
  dbg_exhausted = !won && hn == 0;
  free(heap);
  free(key);
  vset_free(&vs);
  pool_free(&p);
  return won;
#undef SIFTDOWN
#undef HEAPPUSH
}

/* development counters (tools/solver_hist.py): calls, BFS pops, BFS wins, BFS exhausted, A* stages run, A* pops, A* wins,
 * A* exhausted; [8 + b]: calls whose total pops fall into [2^b, 2^(b+1)) */
long long orc_solver_hist[40];
static void hist_add(int i, long long v) {
#pragma omp atomic
  orc_solver_hist[i] += v;
}

/* sokoban_prob.py:99-148 _run_game.  grid tiles: 0 empty 1 solid 2 player 3 crate 4 target.
 * The level is wrapped in a one-tile solid border (:107-124), so level coords = map coords + 1. */
void orc_sokoban_solve(const uint8_t *grid, int H, int W, int power, int *dist_win, int *sol_len) {
  static __thread level_t lv;
  uint8_t *crates = (uint8_t *)malloc((size_t)H * W * 2 + 2);
  int px = 0, py = 0;
  lv.w = W + 2;
  lv.h = H + 2;
  lv.ntg = lv.ncr = 0;
  memset(lv.is_target, 0, sizeof(lv.is_target));
  for (int y = 0; y < lv.h; y++)
    for (int x = 0; x < lv.w; x++) {
      int border = (x == 0 || y == 0 || x == lv.w - 1 || y == lv.h - 1);
      int t = border ? 1 : grid[(y - 1) * W + (x - 1)];
      lv.solid[y * lv.w + x] = (t == 1);
      if (t == 2) { px = x; py = y; }
      if (t == 3) { crates[2 * lv.ncr] = (uint8_t)x; crates[2 * lv.ncr + 1] = (uint8_t)y; lv.ncr++; }
      if (t == 4) { lv.tx[lv.ntg] = (uint8_t)x; lv.ty[lv.ntg] = (uint8_t)y; lv.ntg++; lv.is_target[y * lv.w + x] = 1; }
    }
  init_deadlocks(&lv);
  int h = 0, depth = 0;
  *sol_len = 0;
  long long total = 0;
  int won = solve_bfs(&lv, px, py, crates, power, &h, &depth);
  hist_add(0, 1);
  hist_add(1, dbg_iters);
  hist_add(2, won);
  hist_add(3, dbg_exhausted);
  total += dbg_iters;
  static const double BAL[3] = {1.0, 0.5, 0.0};
  for (int k = 0; k < 3 && !won; k++) {
    won = solve_astar(&lv, px, py, crates, BAL[k], power, &h, &depth);
    hist_add(4, 1);
    hist_add(5, dbg_iters);
    hist_add(6, won);
    hist_add(7, dbg_exhausted);
    total += dbg_iters;
  }
  {
    int b = 0;
    while ((1ll << (b + 1)) <= total && b < 30) b++;
    hist_add(8 + b, 1);
  }
  if (won) {
    *dist_win = 0;
    *sol_len = depth;
  } else {
    *dist_win = h; /* heuristic of the last stage's best node (:147) */
  }
  free(crates);
}
