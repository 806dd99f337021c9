// pcgrl_kernels3d.h -- gfx950 kernels for minecraft_3D_maze (narrow representation), one wavefront per env.
//
// Reference (paths relative to control_pcgrl/): envs/probs/minecraft/minecraft_3D_maze_prob.py:143-181 get_stats,
// :84-93 process_observation; envs/helper_3D.py: _passable :214-319, _flood_fill :354-383, calc_num_regions :396-406,
// run_dijkstra :422-490, calc_longest_path :503-563, remove_stacked_path_tiles :657-675; envs/pcgrl_env.py:267-342.
//
// Lane roles inside the wavefront:
//   lanes 0..Z-1   one z-plane each as a (Y*X)-bit mask: 6-neighbour flood fill = shifts by 1 / X inside the lane and a
//                  DPP row_shr/row_shl between planes; start-candidate masks for the path search
//   lanes 0..3     the four move directions of helper_3D._passable, evaluated together for each queue entry
//   all 64 lanes   grid/overlay conversion, farthest-cell arg-max, path-overlay post-processing, reset RNG (LCG skip-ahead
//                  per lane) and the observation (one 16-byte chunk = 4 cells x 4 one-hot channels per lane per store,
//                  1 KiB contiguous per wave instruction)
// The path search is the reference's FIFO label-correcting search and must keep its pop order (tie-breaks decide
// n_jump, the farthest cell and the path drawn into the next observation), so its queue, the per-cell best-entry table
// and the first-insertion order list live in LDS and one entry is popped per iteration.
#pragma once
#include <hip/hip_runtime.h>

#include "pcgrl_common.h"
#include "pcgrl_kernels2d.h"

namespace pcgrl {

constexpr int M3_MAXCELLS = 512;
constexpr int M3_MAXW = M3_MAXCELLS / 32;  // bit words
constexpr int M3_ENT_CAP = 1536;           // queue entries per search (LDS)
constexpr int M3_NS = 3;

struct M3Lds {
  uint4 ent[M3_ENT_CAP];        // x | y<<8 | z<<16 | kind<<24 ; len | njump<<16 ; parent ; unused
  uint32_t best[M3_MAXCELLS];   // per cell: len << 16 | accepted entry id (the `paths` dict), 0xFFFFFFFF = none
  uint16_t order[M3_MAXCELLS];  // cells in first-insertion order
  uint32_t claim[M3_MAXCELLS];  // scratch of m3_search: lowest batch slot that wants to accept a cell (0xFFFFFFFF = none)
  uint32_t dirt[M3_MAXW + 2];   // tile bit per cell (1 = DIRT), flat index (z*Y + y)*X + x
  uint32_t pathm[M3_MAXW + 2];  // tiles of the best path
  uint32_t over[M3_MAXW + 2];   // overlay mask (transposed index) for the observation
  uint8_t col[64];              // per (y,x): AIR bits over z
};

struct M3Ctx {
  int lane, Z, Y, X, n_cells, nw;
};

enum { M3_WALK = 0, M3_DOWN = 1, M3_UP = 2, M3_JFLAT = 3, M3_JUP = 4, M3_JDOWN = 5, M3_ROOT = 6 };

__device__ inline bool m3_dirt(const M3Lds &L, int cell) { return (L.dirt[cell >> 5] >> (cell & 31)) & 1u; }

// (Y*X)-bit AIR mask of plane z from the flat bit string
__device__ inline uint64_t m3_plane_air(const M3Lds &L, const M3Ctx &c, int z) {
  const int pbits = c.Y * c.X, b0 = z * pbits;
  const int w = b0 >> 5, s = b0 & 31;
  uint64_t lo = (uint64_t)L.dirt[w] | ((uint64_t)L.dirt[w + 1] << 32);
  uint64_t v = lo >> s;
  if (s) v |= (uint64_t)L.dirt[w + 2] << (64 - s);
  const uint64_t pm = pbits >= 64 ? ~0ull : ((1ull << pbits) - 1ull);
  return ~v & pm;
}

__device__ inline uint64_t dpp64_up(uint64_t v) {  // from lane-1 (0 into lane 0 of a DPP row)
  uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x111, 0xF, 0xF, true);
  uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x111, 0xF, 0xF, true);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ inline uint64_t dpp64_down(uint64_t v) {  // from lane+1
  uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x101, 0xF, 0xF, true);
  uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x101, 0xF, 0xF, true);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// helper_3D.py:396-406 calc_num_regions (6-neighbour components of AIR)
__device__ inline int m3_regions(const M3Ctx &c, uint64_t air) {
  uint64_t notx0 = 0, notxl = 0;  // plane bits whose x is not 0 / not X-1
  for (int y = 0; y < c.Y; y++) {
    uint64_t rowm = ((1ull << c.X) - 1ull) << (y * c.X);
    notx0 |= rowm & ~(1ull << (y * c.X));
    notxl |= rowm & ~(1ull << (y * c.X + c.X - 1));
  }
  uint64_t remaining = c.lane < c.Z ? air : 0ull;
  int n = 0;
  while (true) {
    uint64_t b = __ballot(remaining != 0);
    if (b == 0) break;
    int fl = __builtin_ctzll(b);
    uint64_t f = c.lane == fl ? (remaining & (0ull - remaining)) : 0ull;
    while (true) {
      uint64_t d = ((f & notxl) << 1) | ((f & notx0) >> 1) | (f << c.X) | (f >> c.X) | dpp64_up(f) | dpp64_down(f);
      uint64_t nf = d & remaining & ~f;
      if (__ballot(nf != 0) == 0) break;
      f |= nf;
    }
    remaining &= ~f;
    n++;
  }
  return n;
}

__device__ inline int m3_cell(const M3Ctx &c, int x, int y, int z) { return (z * c.Y + y) * c.X + x; }

// One search of helper_3D.run_dijkstra from (sx,sy,sz).  Uniform over the wave; returns number of entries.
// On overflow of the LDS queue sets *overflow.
//
// The reference pops one queue entry at a time.  Here up to 16 consecutive entries are taken per trip, lane 4*i + d
// working on direction d of entry i, which is exact because:
//   * whether entry i is accepted (:437-445) depends on earlier entries only through `best` of ITS OWN cell, so a trip is
//     cut before the second accept candidate of one cell (slots are claimed with an LDS atomic-min; rare);
//   * a successor is queued unless it would be a no-op when popped (cell without head-room, or `best` of its cell not
//     longer).  `best` only ever decreases and every entry of this trip is popped before anything queued now, so testing
//     against `best` AFTER the whole trip's accepts drops exactly entries the sequential run would reject later;
//   * first-visit order and queue order are kept with prefix counts over the lanes (entry-major, direction-minor).
__device__ inline int m3_search(M3Lds &L, const M3Ctx &c, int sx, int sy, int sz, int &n_order, bool &overflow) {
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  for (int i = c.lane; i < c.n_cells; i += 64) {
    L.best[i] = NONE;
    L.claim[i] = NONE;
  }
  if (c.lane == 0) L.ent[0] = make_uint4((uint32_t)sx | ((uint32_t)sy << 8) | ((uint32_t)sz << 16) | ((uint32_t)M3_ROOT << 24), 1u, NONE, 0u);
  int head = 0, tail = 1;
  n_order = 0;
  const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};  // helper_3D.py:220
  const int slot_i = c.lane >> 2, d = c.lane & 3;
  const int dxl = DX[d], dyl = DY[d];
  const uint64_t lt = (1ull << c.lane) - 1ull;
  while (head < tail) {
    const int nb = min(16, tail - head);
    const bool live = slot_i < nb;
    const int id = head + (live ? slot_i : 0);
    const uint4 e = L.ent[id];
    const int x = e.x & 255, y = (e.x >> 8) & 255, z = (e.x >> 16) & 255;
    const int len = e.y & 0xFFFF, nj = e.y >> 16;
    const int ci = m3_cell(c, x, y, z);
    // everything that depends only on the entry is requested in one LDS round trip: its cell's `best`, its column and
    // the columns of this lane's neighbour and jump landing (column 0 stands in for cells outside the map)
    const int nx = x + dxl, ny = y + dyl, jx = x + 2 * dxl, jy = y + 2 * dyl;
    const bool n_in = nx >= 0 && ny >= 0 && nx < c.X && ny < c.Y, j_in = jx >= 0 && jy >= 0 && jx < c.X && jy < c.Y;
    const uint32_t b = L.best[ci];
    const uint32_t cc = L.col[y * c.X + x];
    const uint32_t cn = L.col[n_in ? ny * c.X + nx : 0];
    const uint32_t cj = L.col[j_in ? jy * c.X + jx : 0];
    bool accept = live;
    if (b != NONE && (int)(b >> 16) <= len) accept = false;                     // :437-440
    if (z + 1 == c.Z || !((cc >> (z + 1)) & 1u)) accept = false;                // :443-445 no head-room
    // cut the trip before the second accept candidate of one cell (nothing to check for a single entry)
    int nproc = nb;
    if (nb > 1) {
      if (accept && d == 0) atomicMin(&L.claim[ci], (uint32_t)slot_i);
      const bool dup = accept && L.claim[ci] != (uint32_t)slot_i;
      const uint64_t dupb = __ballot(dup);
      if (dupb) nproc = __builtin_ctzll(dupb) >> 2;  // >= 1: slot 0 always owns its cell
      if (accept && d == 0) L.claim[ci] = NONE;
    }
    const bool doit = accept && slot_i < nproc;
    const bool first = doit && d == 0 && b == NONE;
    const uint64_t fb = __ballot(first);
    if (first) L.order[n_order + __popcll(fb & lt)] = (uint16_t)ci;
    n_order += __popcll(fb);
    if (doit && d == 0) L.best[ci] = ((uint32_t)len << 16) | (uint32_t)id;
    // successors: direction d of entry slot_i (helper_3D.py:214-319)
    bool ok = false;
    int tx = 0, ty = 0, tz = 0, kind = 0, add = 0, nj2 = nj;
    uint32_t ct = 0;  // column mask of the target cell
    if (doit) {
      const int nz = z;
      if (n_in) {
        ct = cn;
        auto A = [&](uint32_t col, int k) -> bool { return (col >> k) & 1u; };
        if ((nz == 0 || !A(cn, nz - 1)) && A(cn, nz) && A(cn, nz + 1)) {
          ok = true; tx = nx; ty = ny; tz = nz; kind = M3_WALK; add = 1;
        } else if (nz >= 1 && (nz - 1 == 0 || !A(cn, nz - 2)) && A(cn, nz - 1) && A(cn, nz) && A(cn, nz + 1)) {
          ok = true; tx = nx; ty = ny; tz = nz - 1; kind = M3_DOWN; add = 2;
        } else if (nz + 2 < c.Z && !A(cn, nz) && A(cn, nz + 1) && A(cn, nz + 2) && A(cc, nz + 2)) {
          ok = true; tx = nx; ty = ny; tz = nz + 1; kind = M3_UP; add = 2;
        } else if (nz - 2 >= 0 && nz + 2 < c.Z && A(cn, nz + 2) && A(cn, nz + 1) && A(cn, nz) && A(cn, nz - 1) && A(cn, nz - 2) &&
                   A(cc, nz + 2) && j_in) {
          ct = cj;
          const int jz = z;
          if (A(cj, jz + 1) && A(cj, jz + 2) && A(cj, jz) && !A(cj, jz - 1)) {
            ok = true; tx = jx; ty = jy; tz = jz; kind = M3_JFLAT; add = 2; nj2 = nj + 1;
          } else if (jz + 3 < c.Z && A(cj, jz + 3) && A(cj, jz + 2) && A(cj, jz + 1) && !A(cj, jz)) {
            ok = true; tx = jx; ty = jy; tz = jz + 1; kind = M3_JUP; add = 3; nj2 = nj + 1;
          } else if (A(cj, jz) && A(cj, jz + 1) && A(cj, jz - 1) && !A(cj, jz - 2)) {
            ok = true; tx = jx; ty = jy; tz = jz - 1; kind = M3_JDOWN; add = 3; nj2 = nj + 1;
          }
        }
      }
    }
    if (ok) {  // never queue what would be a no-op when popped
      if (tz + 1 == c.Z || !((ct >> (tz + 1)) & 1u)) ok = false;
      if (ok) {
        const uint32_t bt = L.best[m3_cell(c, tx, ty, tz)];
        if (bt != NONE && (int)(bt >> 16) <= len + add) ok = false;
      }
    }
    const uint64_t okb = __ballot(ok);
    const int npush = __popcll(okb);
    if (tail + npush > M3_ENT_CAP) {
      overflow = true;
      break;
    }
    if (ok)
      L.ent[tail + __popcll(okb & lt)] = make_uint4((uint32_t)tx | ((uint32_t)ty << 8) | ((uint32_t)tz << 16) | ((uint32_t)kind << 24),
                                                    (uint32_t)(len + add) | ((uint32_t)nj2 << 16), (uint32_t)id, 0u);
    tail += npush;
    head += nproc;
  }
  return tail;
}

// first maximum of len(path) in first-insertion order (helper_3D.py:538-541); returns the cell, sets entry id
__device__ inline int m3_farthest(const M3Lds &L, const M3Ctx &c, int n_order, int &entry) {
  uint32_t key = 0;  // len << 16 | (0xFFFF - k): max key = longest, earliest
  for (int k = c.lane; k < n_order; k += 64) {
    uint32_t len = L.best[L.order[k]] >> 16;
    uint32_t kk = (len << 16) | (uint32_t)(0xFFFF - k);
    key = kk > key ? kk : key;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t other = (uint32_t)__shfl_xor((int)key, o, 64);
    key = other > key ? other : key;
  }
  const int k = 0xFFFF - (int)(key & 0xFFFF);
  const int cell = L.order[k];
  entry = (int)(L.best[cell] & 0xFFFFu);
  return cell;
}

// helper_3D.calc_longest_path + remove_stacked_path_tiles + minecraft_3D_maze_prob.get_stats
// air: this lane's plane (lanes < Z).  Results uniform over the wave.  L.over receives the new overlay mask.
__device__ inline void m3_stats(M3Lds &L, const M3Ctx &c, uint64_t air, int32_t *st, bool &overflow PHASE_ARG) {
  // per-(y,x) column masks for the move rules
  if (c.lane < 64) {
    for (int q = c.lane; q < c.Y * c.X; q += 64) {
      uint32_t m = 0;
      for (int z = 0; z < c.Z; z++) m |= (uint32_t)(!m3_dirt(L, z * c.Y * c.X + q)) << z;
      L.col[q] = (uint8_t)m;
    }
  }
  PHASE_MARK(2);  // column masks
  st[0] = m3_regions(c, air);
  PHASE_MARK(3);  // regions
  // start candidates per plane: AIR with head-room, standing on something, z >= 1 (:520-526)
  const uint64_t above = dpp64_down(air), below = dpp64_up(air);
  uint64_t cand = (c.lane >= 1 && c.lane + 1 < c.Z) ? (air & above & ~below) : 0ull;
  uint32_t marked = 0;  // z-planes of final_visited_map that are fully set (the fancy-index bug, :531)
  int final_value = 0, n_jump = 0;
  for (int i = c.lane; i < c.nw + 2; i += 64) {
    L.pathm[i] = 0;
    L.over[i] = 0;
  }
  while (true) {
    const bool mine = c.lane < c.Z && cand != 0 && !((marked >> c.lane) & 1u);
    const uint64_t b = __ballot(mine);
    if (b == 0) break;
    const int sz = __builtin_ctzll(b);
    const int bit = (int)__shfl((int)__builtin_ctzll(cand | (1ull << 63)), sz, 64);
    const int sy = bit / c.X, sx = bit - sy * c.X;
    int n_order = 0, e1 = 0, e2 = 0;
    m3_search(L, c, sx, sy, sz, n_order, overflow);
    if (overflow) break;
    // mark planes z = v for every coordinate value v of every reached cell
    uint32_t mk = 0;
    for (int k = c.lane; k < n_order; k += 64) {
      int ci = L.order[k];
      int x = ci % c.X, y = (ci / c.X) % c.Y, z = ci / (c.X * c.Y);
      mk |= (1u << x) | (1u << y) | (1u << z);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mk |= (uint32_t)__shfl_xor((int)mk, o, 64);
    marked |= mk & ((1u << c.Z) - 1u);
    const int far = m3_farthest(L, c, n_order, e1);
    const int fx = far % c.X, fy = (far / c.X) % c.Y, fz = far / (c.X * c.Y);
    m3_search(L, c, fx, fy, fz, n_order, overflow);
    if (overflow) break;
    (void)m3_farthest(L, c, n_order, e2);
    const uint4 fe = L.ent[e2];
    const int max_dist = fe.y & 0xFFFF;
    n_jump = fe.y >> 16;  // :553 overwritten by every processed component
    if (max_dist > final_value) {
      final_value = max_dist;
      // materialise the tiles of paths[(mx,my,mz)] into a bit mask (lane 0 walks the parent chain)
      for (int i = c.lane; i < c.nw + 2; i += 64) L.pathm[i] = 0;
      if (c.lane == 0) {
        int id = e2;
        while (true) {
          const uint4 e = L.ent[id];
          const int x = e.x & 255, y = (e.x >> 8) & 255, z = (e.x >> 16) & 255, kind = e.x >> 24;
          auto mark = [&](int mx, int my, int mz) {
            int ci = m3_cell(c, mx, my, mz);
            L.pathm[ci >> 5] |= 1u << (ci & 31);
          };
          mark(x, y, z);
          if (kind == M3_ROOT) break;
          const uint4 pe = L.ent[e.z];
          const int px = pe.x & 255, py = (pe.x >> 8) & 255, pz = (pe.x >> 16) & 255;
          const int mxx = (px + x) >> 1, myy = (py + y) >> 1;  // the jumped-over column
          switch (kind) {
            case M3_DOWN: mark(x, y, pz); break;                              // [(nx, ny, nz)]
            case M3_UP: mark(px, py, pz + 1); break;                          // [(x, y, nz+1)]
            case M3_JFLAT: mark(mxx, myy, pz); break;                         // [(nx, ny, nz)]
            case M3_JUP: mark(mxx, myy, pz); mark(mxx, myy, pz + 1); break;   // [(nx,ny,nz), (nx,ny,nz+1)]
            case M3_JDOWN: mark(mxx, myy, pz); mark(mxx, myy, pz - 1); break; // [(nx,ny,nz), (nx,ny,nz-1)]
            default: break;
          }
          id = (int)e.z;
        }
      }
    }
  }
  PHASE_MARK(4);  // path searches
  // remove_stacked_path_tiles (:657-675) then the transposed overlay of process_observation (:84-93):
  // path tile (x,y,z) is drawn at array index [x][y][z]
  const int pbits = c.Y * c.X;
  for (int ci = c.lane; ci < c.n_cells; ci += 64) {
    bool in = (L.pathm[ci >> 5] >> (ci & 31)) & 1u;
    if (in && ci >= pbits) {
      int lo = ci - pbits;
      if ((L.pathm[lo >> 5] >> (lo & 31)) & 1u) in = false;
    }
    if (in) {
      int x = ci % c.X, y = (ci / c.X) % c.Y, z = ci / pbits;
      int oi = (x * c.Y + y) * c.X + z;
      if (x < c.Z && y < c.Y && z < c.X) atomicOr(&L.over[oi >> 5], 1u << (oi & 31));
    }
  }
  st[1] = final_value;
  st[2] = n_jump;
  PHASE_MARK(5);  // overlay post-processing
}

// observation: (o0, o1, o2, 4) uint8, channel 0 = out of bounds, 1 = AIR, 2 = DIRT, 3 = path overlay
__device__ inline void m3_encode_obs(const M3Lds &L, const M3Ctx &c, const Params &p, int env, const int *pos, bool show_path,
                                     uint8_t *obs_base = nullptr) {
  if (p.obs == nullptr) return;
  if (obs_base == nullptr) obs_base = p.obs;
  const int o0 = p.cfg.obs_window[0], o1 = p.cfg.obs_window[1], o2 = p.cfg.obs_window[2];
  const int total = o0 * o1 * o2, chunks = total >> 2;
  uint4 *dst = (uint4 *)(obs_base + (size_t)env * total * 4);
  const int t0 = pos[0] - o0 / 2, t1 = pos[1] - o1 / 2, t2 = pos[2] - o2 / 2;
  const int o12 = o1 * o2;
  const float inv12 = 1.0f / (float)o12, inv2 = 1.0f / (float)o2;
  for (int ch = c.lane; ch < chunks; ch += 64) {
    uint32_t w[4];
    // (i, j, k) of the chunk's first cell: floor((q + 0.5) / d) is exact in fp32 for these sizes (q < 2^20)
    const int q0 = ch * 4;
    int i = (int)(((float)q0 + 0.5f) * inv12);
    const int r = q0 - i * o12;
    int j = (int)(((float)r + 0.5f) * inv2);
    int k = r - j * o2;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int a = t0 + i, b = t1 + j, d = t2 + k;
      int v = 0;
      if ((unsigned)a < (unsigned)c.Z && (unsigned)b < (unsigned)c.Y && (unsigned)d < (unsigned)c.X) {
        const int ci = (a * c.Y + b) * c.X + d;
        v = 1 + (int)m3_dirt(L, ci);
        if (show_path && ((L.over[ci >> 5] >> (ci & 31)) & 1u)) v = 3;
      }
      w[t] = 1u << (8 * v);
      if (++k == o2) {  // raster order carry
        k = 0;
        if (++j == o1) {
          j = 0;
          ++i;
        }
      }
    }
    store_obs16(dst + ch, make_uint4(w[0], w[1], w[2], w[3]));
  }
}

// reset from the env's RNG streams (envs/pcgrl_env.py:158-188; probabilities, then the map in (z,y,x) order)
__device__ inline void m3_reset_rng(M3Lds &L, const M3Ctx &c, const Params &p, int env, int cpl) {
  Pcg rp, rr;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // rollout kernel: the state stored by lane 0 at the previous reset of this wave
  rp.load(p.rng[env].prob);
  rr.load(p.rng[env].rep);
  double p0 = rp.next_double(), p1 = rp.next_double();
  double total = 0.0;
  total += p0;
  total += p1;
  double c0 = p0 / total, c1 = c0 + p1 / total;
  c0 /= c1;  // cdf /= cdf[-1]
  for (int i = c.lane; i < c.nw + 2; i += 64) L.dirt[i] = 0;
  Pcg end = rr;
  end.jump(p.jump[64]);
  rr.jump(p.jump[c.lane]);
  uint32_t bits = 0;
  const int first = c.lane * cpl;
  for (int k = 0; k < cpl; k++) {
    int ci = first + k;
    if (ci < c.n_cells) {
      double u = rr.next_double();
      int idx = (c0 <= u ? 1 : 0) + (1.0 <= u ? 1 : 0);  // searchsorted(cdf, u, 'right') with cdf[-1] == 1.0
      if (idx >= 1) bits |= 1u << k;
    }
  }
  for (int k = 0; k < cpl; k++) {
    int ci = first + k;
    if (ci < c.n_cells && ((bits >> k) & 1u)) atomicOr(&L.dirt[ci >> 5], 1u << (ci & 31));
  }
  if (c.lane == 0) {
    end.store(p.rng[env].rep);
    rp.store(p.rng[env].prob);
  }
  __atomic_thread_fence(__ATOMIC_RELEASE);
}

// M3_ROLLOUT: pcgrl_rollout, p.n_steps steps per launch with the env state in LDS / registers (see rollout_kernel)
enum M3Mode { M3_STEP = 0, M3_RESET = 1, M3_OBSERVE = 2, M3_STATS_FOR_GRIDS = 3, M3_GET_STATE = 4, M3_ROLLOUT = 5 };

template <int MODE>
__global__ __launch_bounds__(64) void m3_kernel(Params p, int cpl) {
  __shared__ M3Lds L;
  M3Ctx c;
  c.lane = (int)__lane_id();
  c.Z = p.cfg.dims[0];
  c.Y = p.cfg.dims[1];
  c.X = p.cfg.dims[2];
  c.n_cells = c.Z * c.Y * c.X;
  c.nw = (c.n_cells + 31) >> 5;
  const int env = blockIdx.x;
  constexpr int NS = M3_NS;
  PHASE_DECL();
  TRACE_DECL();
  uint32_t *gd = (uint32_t *)p.planes + (size_t)env * 2 * M3_MAXW;  // [dirt words | overlay words]
  EnvState *S = &p.st[env];

  if constexpr (MODE == M3_STATS_FOR_GRIDS) {
    for (int i = c.lane; i < c.nw + 2; i += 64) L.dirt[i] = 0;
    const uint8_t *src = p.init_grids + (size_t)env * c.n_cells;
    for (int ci = c.lane; ci < c.n_cells; ci += 64)
      if (src[ci]) atomicOr(&L.dirt[ci >> 5], 1u << (ci & 31));
    uint64_t air = c.lane < c.Z ? m3_plane_air(L, c, c.lane) : 0ull;
    int32_t st[NS];
    bool ovf = false;
    m3_stats(L, c, air, st, ovf PHASE_PASS);
    if (ovf && c.lane == 0) atomicOr(p.err, 4);
    if (c.lane == 0)
      for (int k = 0; k < NS; k++) p.stats_out[(size_t)env * NS + k] = st[k];
    return;
  }
  if constexpr (MODE == M3_GET_STATE) {
    if (p.out_grids)
      for (int ci = c.lane; ci < c.n_cells; ci += 64) p.out_grids[(size_t)env * c.n_cells + ci] = (gd[ci >> 5] >> (ci & 31)) & 1u;
    if (c.lane == 0) {
      if (p.out_pos)
        for (int d = 0; d < 3; d++) p.out_pos[(size_t)env * 3 + d] = S->pos[d];
      if (p.out_counters) {
        p.out_counters[(size_t)env * 4 + 0] = S->iteration;
        p.out_counters[(size_t)env * 4 + 1] = S->changes;
        p.out_counters[(size_t)env * 4 + 2] = S->n_step;
        p.out_counters[(size_t)env * 4 + 3] = S->iteration;  // episode length so far
      }
      if (p.stats_out)
        for (int k = 0; k < NS; k++) p.stats_out[(size_t)env * NS + k] = S->stats[k];
      if (p.out_last_loss) p.out_last_loss[env] = S->last_loss;
      if (p.out_ep_return) p.out_ep_return[env] = S->ep_return;
    }
    return;
  }

  // load grid + overlay
  for (int i = c.lane; i < c.nw + 2; i += 64) {
    L.dirt[i] = i < c.nw ? gd[i] : 0u;
    L.over[i] = i < c.nw ? gd[M3_MAXW + i] : 0u;
  }
  int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};

  if constexpr (MODE == M3_OBSERVE) {
    // reset()/observe(): no path overlay (PcgrlEnv.reset does not call process_observation)
    m3_encode_obs(L, c, p, env, pos, false);
    return;
  }

  int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
  double last_loss = S->last_loss, ep_return = S->ep_return;
  int32_t st[NS];
  for (int k = 0; k < NS; k++) st[k] = S->stats[k];
  bool ovf = false;
  EnvTargets<NS> trg;
  trg.load(p, env, false);

  if constexpr (MODE == M3_RESET) {
    if (p.mask != nullptr && p.mask[env] == 0) return;
    if (p.refresh_only) {  // statistics (and the path overlay) of the current map, nothing else
      uint64_t air0 = c.lane < c.Z ? m3_plane_air(L, c, c.lane) : 0ull;
      m3_stats(L, c, air0, st, ovf PHASE_PASS);
      if (ovf && c.lane == 0) atomicOr(p.err, 4);
      for (int i = c.lane; i < c.nw; i += 64) gd[M3_MAXW + i] = L.over[i];
      if (c.lane == 0) {
        S->last_loss = trg.loss(p.cfg, st);
        for (int k = 0; k < NS; k++) {
          S->stats[k] = st[k];
          if (p.stats_out) p.stats_out[(size_t)env * NS + k] = st[k];
        }
      }
      return;
    }
    if (p.init_grids) {
      for (int i = c.lane; i < c.nw + 2; i += 64) L.dirt[i] = 0;
      const uint8_t *src = p.init_grids + (size_t)env * c.n_cells;
      for (int ci = c.lane; ci < c.n_cells; ci += 64)
        if (src[ci]) atomicOr(&L.dirt[ci >> 5], 1u << (ci & 31));
      pos[0] = pos[1] = pos[2] = 0;
      if (p.init_pos)
        for (int d = 0; d < 3; d++) pos[d] = p.init_pos[(size_t)env * 3 + d];
    } else {
      m3_reset_rng(L, c, p, env, cpl);
      pos[0] = pos[1] = pos[2] = 0;
    }
    uint64_t air = c.lane < c.Z ? m3_plane_air(L, c, c.lane) : 0ull;
    m3_stats(L, c, air, st, ovf PHASE_PASS);
    n_step = iteration = changes = 0;
    ep_return = 0.0;
    if (p.set_state) {  // pcgrl_set_state: injected map, the caller's counters / return
      if (p.in_counters) {
        iteration = p.in_counters[(size_t)env * 4 + 0];
        changes = p.in_counters[(size_t)env * 4 + 1];
        n_step = p.in_counters[(size_t)env * 4 + 2];
      }
      if (p.in_ep_return) ep_return = p.in_ep_return[env];
    }
    trg.load(p, env, true);
    last_loss = trg.loss(p.cfg, st);
  } else {
   const int K = MODE == M3_ROLLOUT ? p.n_steps : 1;
   const size_t N = (size_t)p.n_envs;
   for (int k = 0; k < K; k++) {
    const size_t o = (size_t)k * N + (size_t)env;  // index of this step's outputs
    uint8_t *obs_k = p.obs == nullptr ? nullptr
                     : (MODE == M3_ROLLOUT && !p.obs_last_only ? p.obs + (size_t)k * N * (size_t)p.obs_env_bytes : p.obs);
    const bool want_obs = MODE != M3_ROLLOUT || !p.obs_last_only || k == K - 1;
    // ---- step (envs/pcgrl_env.py:267-342 with narrow_rep.py:89-102)
    const int action = p.actions[o];
    const bool bad = action < 0 || action >= 2;
    const bool upd_only = p.update_only != 0;
    iteration += upd_only ? 0 : 1;
    bool change = false;
    if (!bad) {
      const int ci = m3_cell(c, pos[2], pos[1], pos[0]);  // pos = (z, y, x)
      const bool old = m3_dirt(L, ci);
      change = old != (action != 0);
      if (change && c.lane == 0) L.dirt[ci >> 5] ^= 1u << (ci & 31);
      const int idx = n_step % c.n_cells;  // Q1
      pos[0] = idx / (c.Y * c.X);
      pos[1] = (idx / c.X) % c.Y;
      pos[2] = idx % c.X;
      n_step++;
    } else if (c.lane == 0) {
      atomicOr(p.err, 1);
    }
    if (upd_only) {  // rep.update() only: map, position, observation (with the stale overlay)
      m3_encode_obs(L, c, p, env, pos, true);
      for (int i = c.lane; i < c.nw; i += 64) gd[i] = L.dirt[i];
      if (c.lane == 0) {
        S->pos[0] = pos[0];
        S->pos[1] = pos[1];
        S->pos[2] = pos[2];
        S->n_step = n_step;
      }
      return;
    }
    changes += change ? 1 : 0;
    bool done = iteration > p.cfg.max_iterations;
    if (p.cfg.max_changes >= 0) done = done || changes > p.cfg.max_changes;
    const bool do_reset = done && p.auto_reset != 0;
    // the observation is assembled BEFORE the stats refresh (pcgrl_env.py:298-299 vs :314-323): it shows the path of
    // the previous stats update on the already edited map
    PHASE_MARK(0);  // loads + action
    if (!do_reset && want_obs) m3_encode_obs(L, c, p, env, pos, true, obs_k);
    PHASE_MARK(1);  // observation
    if (change) {
      uint64_t air = c.lane < c.Z ? m3_plane_air(L, c, c.lane) : 0ull;
      m3_stats(L, c, air, st, ovf PHASE_PASS);
    }
    const double loss = trg.loss(p.cfg, st);
    const double rew = loss - last_loss;
    last_loss = loss;
    ep_return += rew;
    if (c.lane == 0) {
      if (p.reward) p.reward[o] = (float)rew;
      if (p.reward64) p.reward64[o] = rew;
      if (p.done) p.done[o] = done ? 1 : 0;
      if (p.stats_out)
        for (int i = 0; i < NS; i++) p.stats_out[o * NS + i] = st[i];
    }
    if (do_reset) {
      if (c.lane == 0) latch_episode<NS>(p, env, S, ep_return, iteration, st);
      m3_reset_rng(L, c, p, env, cpl);
      pos[0] = pos[1] = pos[2] = 0;
      uint64_t air = c.lane < c.Z ? m3_plane_air(L, c, c.lane) : 0ull;
      m3_stats(L, c, air, st, ovf PHASE_PASS);
      n_step = iteration = changes = 0;
      ep_return = 0.0;
      trg.load(p, env, true);
      last_loss = trg.loss(p.cfg, st);
      if (want_obs) m3_encode_obs(L, c, p, env, pos, false, obs_k);
    }
   }
  }
  if (ovf && c.lane == 0) atomicOr(p.err, 4);
  // write back
  for (int i = c.lane; i < c.nw; i += 64) {
    gd[i] = L.dirt[i];
    gd[M3_MAXW + i] = L.over[i];
  }
  if (c.lane == 0) {
    trg.write_ctrl_obs(p, env, st);
    trg.commit(p, env);
    S->pos[0] = pos[0];
    S->pos[1] = pos[1];
    S->pos[2] = pos[2];
    S->n_step = n_step;
    S->iteration = iteration;
    S->changes = changes;
    S->flags = 0;
    S->last_loss = last_loss;
    S->ep_return = ep_return;
    for (int k = 0; k < NS; k++) S->stats[k] = st[k];
  }
  PHASE_MARK(6);
  PHASE_FLUSH();
  TRACE_PUT(0, _tr0);
  TRACE_PUT(1, TRACE_NOW());
  TRACE_DRAIN();
  TRACE_PUT(2, TRACE_NOW());
}

}  // namespace pcgrl
