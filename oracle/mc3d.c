/*
 * mc3d.c -- restatement of the reference's 3-D maze statistics (minecraft_3D_maze).
 *
 * TEST INFRASTRUCTURE ONLY (part of the oracle, see pcgrl_oracle.h).
 *
 * Follows envs/probs/minecraft/minecraft_3D_maze_prob.py:143-181 (get_stats) and envs/helper_3D.py:
 *   _passable :214-319, _flood_fill :354-383, calc_num_regions :396-406, run_dijkstra :422-490,
 *   calc_longest_path :503-563, remove_stacked_path_tiles :657-675.
 * including the behaviours that matter for parity:
 *   - starts must stand on something and z == 0 is never a start (:525-526);
 *   - the FIFO label-correcting search accepts a cell again when a strictly shorter path arrives (:437-440);
 *   - "farthest" = first maximum in first-insertion order of the paths dict (:538-541);
 *   - visited_map[np.array(list(paths.keys()))] = 1 marks whole z-planes for every coordinate VALUE that occurs in a
 *     reached key (:531);
 *   - n_jump is overwritten by every processed component, not only the best (:553).
 * Tiles: 0 = AIR (passable), 1 = DIRT.  Grid index (z*Y + y)*X + x.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int Z, Y, X;
  const uint8_t *g;
} map3_t;

#define AIR(m, x, y, z) ((m)->g[((z) * (m)->Y + (y)) * (m)->X + (x)] == 0)

/* helper_3D.py:354-406: 6-neighbour flood fill over AIR */
static int regions3d(const map3_t *m) {
  int n = m->Z * m->Y * m->X, regions = 0;
  int16_t *color = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
  int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  static const int D[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
  for (int i = 0; i < n; i++) color[i] = -1;
  for (int s = 0; s < n; s++) {
    if (m->g[s] != 0 || color[s] != -1) continue;
    int head = 0, tail = 0;
    q[tail++] = s;
    color[s] = 1;
    while (head < tail) {
      int c = q[head++];
      int x = c % m->X, y = (c / m->X) % m->Y, z = c / (m->X * m->Y);
      for (int d = 0; d < 6; d++) {
        int nx = x + D[d][0], ny = y + D[d][1], nz = z + D[d][2];
        if (nx < 0 || ny < 0 || nz < 0 || nx >= m->X || ny >= m->Y || nz >= m->Z) continue;
        int ni = (nz * m->Y + ny) * m->X + nx;
        if (m->g[ni] != 0 || color[ni] != -1) continue;
        color[ni] = 1;
        q[tail++] = ni;
      }
    }
    regions++;
  }
  free(color);
  free(q);
  return regions;
}

/* one queue entry of run_dijkstra: foothold + how it was reached (path = parent's path + traversed + foothold) */
typedef struct {
  int16_t x, y, z;
  int16_t len;     /* len(path) */
  int16_t njump;
  int8_t ntrav;
  int16_t trav[2][3];
  int32_t parent;
} entry_t;

typedef struct {
  entry_t *e;
  int n, cap;
} evec_t;

static int epush(evec_t *v, entry_t en) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 256;
    v->e = (entry_t *)realloc(v->e, sizeof(entry_t) * (size_t)v->cap);
  }
  v->e[v->n] = en;
  return v->n++;
}

typedef struct {
  int32_t *best;   /* per cell: entry id of the accepted path, -1 = none (the `paths` dict) */
  int32_t *order;  /* cells in first-insertion order */
  int n_order;
  evec_t ev;
} search_t;

/* helper_3D.py:214-319 _passable: successors of foothold (x,y,z) in direction order (1,0),(0,1),(-1,0),(0,-1) */
static void successors(const map3_t *m, evec_t *ev, int cur) {
  static const int DIR[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
  const entry_t c = ev->e[cur];
  const int x = c.x, y = c.y, z = c.z, Z = m->Z;
  for (int d = 0; d < 4; d++) {
    int nx = x + DIR[d][0], ny = y + DIR[d][1], nz = z;
    int jx = x + 2 * DIR[d][0], jy = y + 2 * DIR[d][1], jz = z;
    if (nx < 0 || ny < 0 || nx >= m->X || ny >= m->Y) continue;
    entry_t s;
    memset(&s, 0, sizeof(s));
    s.parent = cur;
    s.njump = c.njump;
    int ok = 0;
    if ((nz == 0 || (nz > 0 && !AIR(m, nx, ny, nz - 1))) && AIR(m, nx, ny, nz) && AIR(m, nx, ny, nz + 1)) {
      /* walk forward (:232-242) */
      s.x = nx; s.y = ny; s.z = nz; s.ntrav = 0; ok = 1;
    } else if ((nz - 1 == 0 || (nz - 1 > 0 && !AIR(m, nx, ny, nz - 2))) && nz - 1 >= 0 && AIR(m, nx, ny, nz - 1) &&
               AIR(m, nx, ny, nz) && AIR(m, nx, ny, nz + 1)) {
      /* step down (:246-255) */
      s.x = nx; s.y = ny; s.z = nz - 1; s.ntrav = 1;
      s.trav[0][0] = nx; s.trav[0][1] = ny; s.trav[0][2] = nz; ok = 1;
    } else if (nz + 2 < Z && !AIR(m, nx, ny, nz) && AIR(m, nx, ny, nz + 1) && AIR(m, nx, ny, nz + 2) && AIR(m, x, y, nz + 2)) {
      /* step up (:262-269) */
      s.x = nx; s.y = ny; s.z = nz + 1; s.ntrav = 1;
      s.trav[0][0] = x; s.trav[0][1] = y; s.trav[0][2] = nz + 1; ok = 1;
    } else if (nz - 2 >= 0 && nz + 2 < Z && AIR(m, nx, ny, nz + 2) && AIR(m, nx, ny, nz + 1) && AIR(m, nx, ny, nz) &&
               AIR(m, nx, ny, nz - 1) && AIR(m, nx, ny, nz - 2) && AIR(m, x, y, nz + 2) && jx >= 0 && jy >= 0 && jx < m->X &&
               jy < m->Y) {
      /* jump over a one-tile gap (:283-319) */
      s.njump = c.njump + 1;
      s.trav[0][0] = nx; s.trav[0][1] = ny; s.trav[0][2] = nz;
      if (AIR(m, jx, jy, jz + 1) && AIR(m, jx, jy, jz + 2) && AIR(m, jx, jy, jz) && !AIR(m, jx, jy, jz - 1)) {
        s.x = jx; s.y = jy; s.z = jz; s.ntrav = 1; ok = 1;
      } else if (jz + 3 < Z && AIR(m, jx, jy, jz + 3) && AIR(m, jx, jy, jz + 2) && AIR(m, jx, jy, jz + 1) && !AIR(m, jx, jy, jz)) {
        s.x = jx; s.y = jy; s.z = jz + 1; s.ntrav = 2;
        s.trav[1][0] = nx; s.trav[1][1] = ny; s.trav[1][2] = nz + 1; ok = 1;
      } else if (AIR(m, jx, jy, jz) && AIR(m, jx, jy, jz + 1) && AIR(m, jx, jy, jz - 1) && !AIR(m, jx, jy, jz - 2)) {
        s.x = jx; s.y = jy; s.z = jz - 1; s.ntrav = 2;
        s.trav[1][0] = nx; s.trav[1][1] = ny; s.trav[1][2] = nz - 1; ok = 1;
      }
    }
    if (!ok) continue;
    s.len = (int16_t)(c.len + s.ntrav + 1);
    epush(ev, s);
  }
}

/* helper_3D.py:422-490 run_dijkstra */
static void run_search(const map3_t *m, search_t *S, int sx, int sy, int sz) {
  int n = m->Z * m->Y * m->X;
  for (int i = 0; i < n; i++) S->best[i] = -1;
  S->n_order = 0;
  S->ev.n = 0;
  entry_t root;
  memset(&root, 0, sizeof(root));
  root.x = sx; root.y = sy; root.z = sz; root.len = 1; root.parent = -1;
  epush(&S->ev, root);
  for (int head = 0; head < S->ev.n; head++) { /* the entry vector IS the FIFO queue */
    entry_t c = S->ev.e[head];
    int ci = (c.z * m->Y + c.y) * m->X + c.x;
    if (S->best[ci] >= 0 && S->ev.e[S->best[ci]].len <= c.len) continue; /* :437-440 */
    if (c.z + 1 == m->Z || !AIR(m, c.x, c.y, c.z + 1)) continue;         /* :443-445 no head-room */
    if (S->best[ci] < 0) S->order[S->n_order++] = ci;
    S->best[ci] = head;
    successors(m, &S->ev, head);
  }
}

static int farthest(const search_t *S) { /* :538-541 first maximum in first-insertion order */
  int bi = -1, bl = -1;
  for (int k = 0; k < S->n_order; k++) {
    int l = S->ev.e[S->best[S->order[k]]].len;
    if (l > bl) {
      bl = l;
      bi = S->order[k];
    }
  }
  return bi;
}

void orc_mc3d_stats(const uint8_t *grid, int Z, int Y, int X, int32_t *stats, int16_t *path_xyz, int32_t *path_len) {
  map3_t m = {Z, Y, X, grid};
  int n = Z * Y * X;
  search_t S;
  S.best = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  S.order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  S.ev.e = NULL;
  S.ev.n = S.ev.cap = 0;
  uint8_t *final_visited = (uint8_t *)calloc((size_t)n, 1);
  int final_value = 0, n_jump = 0;
  int16_t *best_path = (int16_t *)malloc(sizeof(int16_t) * 3 * (size_t)(4 * n + 8));
  int best_path_n = 0;

  for (int s = 0; s < n; s++) { /* AIR cells in (z, y, x)-major order (helper_3D.py:22-30) */
    if (grid[s] != 0) continue;
    int x = s % X, y = (s / X) % Y, z = s / (X * Y);
    if (final_visited[s]) continue;
    if (z + 1 == Z || !AIR(&m, x, y, z + 1)) { /* :520-522 */
      final_visited[s] = 1;
      continue;
    }
    if (z - 1 < 0 || AIR(&m, x, y, z - 1)) continue; /* :525-526 */
    run_search(&m, &S, x, y, z);
    /* :530-533 with the fancy-index bug: every coordinate value v of every reached key marks plane z = v */
    for (int k = 0; k < S.n_order; k++) {
      int ci = S.order[k];
      int c[3] = {ci % X, (ci / X) % Y, ci / (X * Y)};
      for (int a = 0; a < 3; a++)
        if (c[a] < Z) memset(final_visited + (size_t)c[a] * Y * X, 1, (size_t)Y * X);
    }
    int far = farthest(&S);
    run_search(&m, &S, far % X, (far / X) % Y, far / (X * Y));
    int far2 = farthest(&S);
    const entry_t *fe = &S.ev.e[S.best[far2]];
    int max_dist = fe->len;
    n_jump = fe->njump; /* :553 overwritten for every component */
    if (max_dist > final_value) {
      final_value = max_dist;
      /* materialise paths[(mx,my,mz)]: root ... (traversed tiles, foothold) ... */
      best_path_n = 0;
      int chain[4096], cn = 0;
      for (int id = S.best[far2]; id >= 0; id = S.ev.e[id].parent) chain[cn++] = id;
      for (int k = cn - 1; k >= 0; k--) {
        const entry_t *e = &S.ev.e[chain[k]];
        for (int t = 0; t < e->ntrav; t++) {
          best_path[3 * best_path_n] = e->trav[t][0];
          best_path[3 * best_path_n + 1] = e->trav[t][1];
          best_path[3 * best_path_n + 2] = e->trav[t][2];
          best_path_n++;
        }
        best_path[3 * best_path_n] = e->x;
        best_path[3 * best_path_n + 1] = e->y;
        best_path[3 * best_path_n + 2] = e->z;
        best_path_n++;
      }
    }
  }
  /* remove_stacked_path_tiles (:657-675): as a set, drop every tile whose lower neighbour is also on the path */
  int out_n = 0;
  uint8_t *inpath = (uint8_t *)calloc((size_t)n, 1);
  for (int k = 0; k < best_path_n; k++)
    inpath[(best_path[3 * k + 2] * Y + best_path[3 * k + 1]) * X + best_path[3 * k]] = 1;
  for (int ci = 0; ci < n; ci++) {
    if (!inpath[ci]) continue;
    int x = ci % X, y = (ci / X) % Y, z = ci / (X * Y);
    if (z > 0 && inpath[((z - 1) * Y + y) * X + x]) continue;
    path_xyz[3 * out_n] = (int16_t)x;
    path_xyz[3 * out_n + 1] = (int16_t)y;
    path_xyz[3 * out_n + 2] = (int16_t)z;
    out_n++;
  }
  *path_len = out_n;
  stats[0] = regions3d(&m);
  stats[1] = final_value;
  stats[2] = n_jump;
  free(inpath);
  free(best_path);
  free(final_visited);
  free(S.best);
  free(S.order);
  free(S.ev.e);
}
