// pcgrl_kernels3d.h -- gfx950 kernels for minecraft_3D_maze (narrow representation).
//
// Reference (paths relative to control_pcgrl/): envs/probs/minecraft/minecraft_3D_maze_prob.py:143-181 get_stats,
// :84-93 process_observation; envs/helper_3D.py: _passable :214-319, _flood_fill :354-383, calc_num_regions :396-406,
// run_dijkstra :422-490, calc_longest_path :503-563, remove_stacked_path_tiles :657-675; envs/pcgrl_env.py:267-342.
//
// One workgroup per env.  pcgrl_step runs TWO specialised wavefronts over the env (like the 2-D step kernel):
//   wave 0 "simulate"  action -> statistics (regions, path searches) -> reward / done -> auto-reset -> state write-back
//   wave 1 "observe"   replays the (trivial) action / reset on its own copy and streams the observation, which shows the
//                      path overlay of the PREVIOUS statistics update (pcgrl_env.py:298-299 vs :314-323), so it does
//                      not depend on this step's searches.
// Lane roles inside the simulate wave:
//   lanes 0..Z-1   one z-plane each as a (Y*X)-bit mask: 6-neighbour flood fill = shifts by 1 / X inside the lane and a
//                  DPP row_shr/row_shl between planes; start-candidate masks for the path search
//   lane 4*i + d   move direction d of queue entry i of the current trip of the path search (16 entries per trip)
//   all 64 lanes   farthest-cell arg-max, overlay post-processing, reset RNG (LCG skip-ahead per lane)
//
// PATH SEARCH.  helper_3D.run_dijkstra is a FIFO label-correcting search whose pop order decides n_jump, the farthest
// cell and the path drawn into the next observation, so the queue order is kept exactly (see m3_search).
//
// SLOT CACHE.  calc_longest_path (:503-563) starts one pair of searches per start candidate, and its whole-plane visited
// marking (:531) leaves at most ONE processed candidate per z-plane: the first candidate of the plane in (y, x) order.
// Everything a plane's pair of searches produces -- the marks, max_dist, n_jump, the path tiles -- is a function of the
// cells the searches READ.  The engine keeps, per env and plane, that result together with the read set (a superset: the
// columns the move rules looked at and the range of heights, z-2 .. z+3 around every accepted cell) in HBM.  A step edits one cell: slots whose read set contains it are
// dropped, every other slot is still exact, and the sequential candidate walk re-runs only the searches it actually
// needs.  Under random edits most steps re-run no or one pair instead of all of them.
#pragma once
#include <hip/hip_runtime.h>

#include "pcgrl_common.h"
#include "pcgrl_kernels2d.h"

namespace pcgrl {

constexpr int M3_MAXCELLS = 512;
constexpr int M3_MAXW = M3_MAXCELLS / 32;  // bit words
constexpr int M3_ENT_CAP = 1536;           // queue entries per search (LDS)
constexpr int M3_NS = 3;
constexpr int M3_SLOTS = 6;                // start planes z = 1 .. Z-2 (Z <= 8)

// cached result of one start plane (see SLOT CACHE)
struct alignas(8) M3Slot {
  uint8_t start;   // bit index (y*X + x) of the start cell in its plane
  uint8_t valid;
  uint16_t max_dist;
  uint16_t n_jump;
  uint8_t mk;      // z-planes marked visited by the first search (the fancy-index bug, :531)
  uint8_t zr;      // read set, heights: lowest | highest << 4 plane the move rules looked at
  uint32_t rs[2];  // read set, columns: bit q = the searches looked at column q = y*X + x
  uint32_t pathm[M3_MAXW];  // tiles of paths[farthest] of the second search
};
static_assert(sizeof(M3Slot) == 80, "slot layout");
constexpr int M3_SLOT_WORDS = (int)(sizeof(M3Slot) / 4) * M3_SLOTS;

struct M3Lds {
  uint2 ent[M3_ENT_CAP];        // cell | kind<<9 | njump<<12 | parent<<20 ; len | x<<12 | y<<18 | z<<24 | direction<<28
  uint32_t best[M3_MAXCELLS];   // per cell: epoch<<24 | len<<12 | accepted entry id (the `paths` dict of the current search)
  uint16_t order[M3_MAXCELLS];  // cells in first-insertion order
  uint32_t claim[M3_MAXCELLS];  // scratch of m3_search: lowest trip slot popping a cell (0xFFFFFFFF between trips)
  uint32_t dirt[M3_MAXW + 2];   // tile bit per cell (1 = DIRT), flat index (z*Y + y)*X + x
  uint32_t pathm[M3_MAXW + 2];  // tiles of the best path
  uint32_t over[M3_MAXW + 2];   // overlay mask (transposed index) for the observation
  uint8_t col[64];              // per (y,x): AIR bits over z
  M3Slot slot[M3_SLOTS];
  uint32_t epoch;               // current search id in `best`
#ifdef PCGRL_PHASE_TIMING
  uint32_t dbg[8];              // development counters: trips, queue entries, searches, cycles of search 1 / farthest / search 2 / chain walk
#endif
};
struct M3ObsLds {  // the observe wave's own copy
  uint32_t dirt[M3_MAXW + 2];
  uint32_t over[M3_MAXW + 2];
};

struct M3Ctx {
  int lane, Z, Y, X, n_cells, nw;
};

enum { M3_WALK = 0, M3_DOWN = 1, M3_UP = 2, M3_JFLAT = 3, M3_JUP = 4, M3_JDOWN = 5, M3_ROOT = 6 };
constexpr uint32_t M3_NOPARENT = 0xFFFu;

__device__ inline bool m3_bit(const uint32_t *w, int i) { return (w[i >> 5] >> (i & 31)) & 1u; }

// (Y*X)-bit AIR mask of plane z from the flat bit string
__device__ inline uint64_t m3_plane_air(const uint32_t *dirt, const M3Ctx &c, int z) {
  const int pbits = c.Y * c.X, b0 = z * pbits;
  const int w = b0 >> 5, s = b0 & 31;
  uint64_t lo = (uint64_t)dirt[w] | ((uint64_t)dirt[w + 1] << 32);
  uint64_t v = lo >> s;
  if (s) v |= (uint64_t)dirt[w + 2] << (64 - s);
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

// maximum over the 64 lanes (result uniform): DPP butterfly inside the 16-lane rows, the four rows through SGPRs
__device__ inline uint32_t wave_max(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
// OR over the 64 lanes (result uniform), same shape
__device__ inline uint32_t wave_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
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
    while (true) {  // two expansion rounds per trip (one ballot per two rounds)
      uint64_t d = ((f & notxl) << 1) | ((f & notx0) >> 1) | (f << c.X) | (f >> c.X) | dpp64_up(f) | dpp64_down(f);
      f |= d & remaining;
      d = ((f & notxl) << 1) | ((f & notx0) >> 1) | (f << c.X) | (f >> c.X) | dpp64_up(f) | dpp64_down(f);
      const uint64_t nf = d & remaining & ~f;
      if (__ballot(nf != 0) == 0) break;
      f |= nf;
    }
    remaining &= ~f;
    n++;
  }
  return n;
}

__device__ inline int m3_cell(const M3Ctx &c, int x, int y, int z) { return (z * c.Y + y) * c.X + x; }

// One search of helper_3D.run_dijkstra from (sx,sy,sz).  Uniform over the wave; returns the number of queue entries.
// On overflow of the LDS queue sets `overflow`.
//
// The reference pops one queue entry at a time.  Here up to 16 consecutive entries are taken per trip, lane 4*i + d
// working on direction d of entry i, which is exact because:
//   * whether entry i is accepted (:437-445) depends on earlier entries only through `best` of ITS OWN cell.  Every
//     popped entry registers its trip slot for its cell (an LDS min issued BEFORE the reads of the trip, so one round trip
//     returns `best`, the columns and the first slot of the cell together); the trip is cut before an accept candidate
//     that is not the first entry of its cell in the trip (rare), which then runs first in the next trip;
//   * a successor is queued unless it is known to be a no-op when popped (cell without head-room, or `best` of its cell --
//     as read at the start of the trip: `best` only decreases -- not longer).  Entries that turn out to be no-ops later are
//     rejected when popped, like in the reference;
//   * first-visit order and queue order are kept with prefix counts over the lanes (entry-major, direction-minor).
// `best` entries carry the search's epoch, so nothing is cleared between searches.
// mk: coordinate values of all reached cells (bit v set if some reached cell has x, y or z == v);  rs: read set (one bit
// per column) accumulated PER LANE for the slot cache (the caller ORs the lanes together).
__device__ inline int m3_search(M3Lds &L, const M3Ctx &c, int sx, int sy, int sz, int &n_order, uint32_t &mk, uint64_t &rs,
                                int &zlo, int &zhi, bool &overflow) {
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  uint32_t epoch = L.epoch + 1;  // (uniform: every lane reads the same word)
  if (epoch > 255u) {            // wrapped: clear the table once
    for (int i = c.lane; i < c.n_cells; i += 64) L.best[i] = 0;
    epoch = 1;
  }
  if (c.lane == 0) {
    L.epoch = epoch;
    L.ent[0] = make_uint2((uint32_t)m3_cell(c, sx, sy, sz) | ((uint32_t)M3_ROOT << 9) | (M3_NOPARENT << 20),
                          1u | ((uint32_t)sx << 12) | ((uint32_t)sy << 18) | ((uint32_t)sz << 24));
  }
  int head = 0, tail = 1;
  n_order = 0;
  const int slot_i = c.lane >> 2, d = c.lane & 3;
  const int dxl = d == 0 ? 1 : (d == 2 ? -1 : 0), dyl = d == 1 ? 1 : (d == 3 ? -1 : 0);  // helper_3D.py:220
  const int dq = dyl * c.X + dxl;  // column-index step of this lane's direction
  const uint64_t lt = (1ull << c.lane) - 1ull;
  const int YX = c.Y * c.X;
  uint32_t mkl = 0;
#ifdef PCGRL_PHASE_TIMING
  int dbg_trips = 0;
#endif
  while (head < tail) {
#ifdef PCGRL_PHASE_TIMING
    dbg_trips++;
#endif
#ifdef PCGRL_PHASE_TIMING
    uint64_t tt0_ = __builtin_readcyclecounter();
#define M3_TT(i)                                                    \
  do {                                                              \
    uint64_t tt1_ = __builtin_readcyclecounter();                   \
    if (c.lane == 0) L.dbg[i] += (uint32_t)(tt1_ - tt0_);           \
    tt0_ = tt1_;                                                    \
  } while (0)
#else
#define M3_TT(i) \
  do {           \
  } while (0)
#endif
    const int nb = min(16, tail - head);
    const bool live = slot_i < nb;
    const int id = head + (live ? slot_i : 0);
    const uint2 e = L.ent[id];
    const int ci = e.x & 511, nj = (e.x >> 12) & 255, len = (int)(e.y & 0xFFFu);
    const int x = (e.y >> 12) & 63, y = (e.y >> 18) & 63, z = (int)((e.y >> 24) & 15u);
    if (live && d == 0) atomicMin(&L.claim[ci], (uint32_t)slot_i);
    // everything that depends only on the entry comes back in one LDS round trip: `best` and first slot of its cell, its
    // column and the columns of this lane's neighbour and jump landing (column 0 stands in for cells outside the map)
    const int nx = x + dxl, ny = y + dyl, jx = nx + dxl, jy = ny + dyl;
    const bool n_in = ((unsigned)nx < (unsigned)c.X) & ((unsigned)ny < (unsigned)c.Y);
    const bool j_in = ((unsigned)jx < (unsigned)c.X) & ((unsigned)jy < (unsigned)c.Y);
    const int qc = ci - z * YX, qn = n_in ? qc + dq : 0, qj = j_in ? qc + 2 * dq : 0;
    const uint32_t b = L.best[ci];
    const uint32_t first_slot = L.claim[ci];
    const uint32_t cc = L.col[qc], cn = L.col[qn], cj = L.col[qj];
    const bool seen = (b >> 24) == epoch;
    M3_TT(3);  // entry read + second round of reads issued (wait happens at first use)
    // :437-440 (an entry that is not shorter is dropped) and :443-445 (no head-room); cells >= Z read as not-AIR
    // (bitwise on purpose, here and below: `&&` / `||` compile to exec-masked branches, these to mask arithmetic)
    const bool accept = live & !(seen & ((int)((b >> 12) & 0xFFFu) <= len)) & (((cc >> (z + 1)) & 1u) != 0u);
    // cut the trip before an accept candidate that is not the first popped entry of its cell in this trip
    const uint64_t dupb = __ballot(accept & (d == 0) & (first_slot != (uint32_t)slot_i));
    const int nproc = dupb ? (__builtin_ctzll(dupb) >> 2) : nb;  // >= 1: slot 0 is always the first of its cell
    if (live && d == 0) L.claim[ci] = NONE;
    const bool doit = accept & (slot_i < nproc);
    const bool first = doit & (d == 0) & !seen;
    const uint64_t fb = __ballot(first);
    if (first) L.order[n_order + __popcll(fb & lt)] = (uint16_t)ci;
    n_order += __popcll(fb);
    const bool acc0 = doit & (d == 0);
    if (acc0) L.best[ci] = (epoch << 24) | ((uint32_t)len << 12) | (uint32_t)id;
    M3_TT(4);  // accept / claim / order / best
    // Successor in direction d (helper_3D._passable :214-319), branch-free on 6-bit windows of the columns: bit i of
    // a window = AIR at height z-2+i (below the floor and above the ceiling read as not-AIR, which is what every rule's
    // explicit bounds check amounts to).  The six rules are mutually exclusive.
    const uint32_t wn = ((cn << 2) >> z) & 0x3Fu, wj = ((cj << 2) >> z) & 0x3Fu, c4 = (cc >> (z + 2)) & 1u;
    const bool walk = (wn & 0x0Eu) == 0x0Cu;                      // stands at z: !n[z-1], n[z], n[z+1]
    const bool down = (z >= 1) & ((wn & 0x0Fu) == 0x0Eu);            // stands at z-1: !n[z-2], n[z-1], n[z], n[z+1]
    const bool up = ((wn & 0x1Cu) == 0x18u) & (c4 != 0u);                  // stands at z+1: !n[z], n[z+1], n[z+2], own z+2 free
    const bool gap = (z >= 2) & ((wn & 0x1Fu) == 0x1Fu) & (c4 != 0u) & j_in;  // n[z-2..z+2] all AIR: a gap to jump over
    const bool jflat = gap & ((wj & 0x1Eu) == 0x1Cu);              // !j[z-1], j[z], j[z+1], j[z+2]
    const bool jup = gap & ((wj & 0x3Cu) == 0x38u);                // !j[z], j[z+1], j[z+2], j[z+3]
    const bool jdown = gap & ((wj & 0x0Fu) == 0x0Eu);              // !j[z-2], j[z-1], j[z], j[z+1]
    const bool jump = jflat | jup | jdown;
    bool ok = doit & n_in & (walk | down | up | jump);
    const int kind = walk ? M3_WALK : (down ? M3_DOWN : (up ? M3_UP : (jflat ? M3_JFLAT : (jup ? M3_JUP : M3_JDOWN))));
    const int add = walk ? 1 : ((jup | jdown) ? 3 : 2);
    const int tz = z + ((up | jup) ? 1 : 0) - ((down | jdown) ? 1 : 0);
    const int tq = jump ? qj : qn, tx = jump ? jx : nx, ty = jump ? jy : ny;
    const int tcell = tz * YX + tq;
    M3_TT(5);  // move rules + read set
    // Never queue what is known to be a no-op when popped (the target always has head-room: every rule checks it) --
    // but only while the queue is long: the check is one more dependent LDS round trip per trip, and with a short queue
    // (corridors: the searches that make a launch wait) the few useless entries are dropped for free when popped.
    auto prune = [&]() {
      const uint32_t bt = L.best[ok ? tcell : 0];
      ok &= !(((bt >> 24) == epoch) & ((int)((bt >> 12) & 0xFFFu) <= len + add));
    };
    const bool lazy = tail - head <= 32;
    if (!lazy) prune();
    uint64_t okb = __ballot(ok);
    int npush = __popcll(okb);
    if (tail + npush > M3_ENT_CAP) {
      if (lazy) {  // (never an overflow that the check would have avoided)
        prune();
        okb = __ballot(ok);
        npush = __popcll(okb);
      }
      if (tail + npush > M3_ENT_CAP) {
        overflow = true;
        break;
      }
    }
    if (ok)
      L.ent[tail + __popcll(okb & lt)] =
          make_uint2((uint32_t)tcell | ((uint32_t)kind << 9) | ((uint32_t)(nj + (jump ? 1 : 0)) << 12) | ((uint32_t)id << 20),
                     (uint32_t)(len + add) | ((uint32_t)tx << 12) | ((uint32_t)ty << 18) | ((uint32_t)tz << 24) | ((uint32_t)d << 28));
    tail += npush;
    head += nproc;
    M3_TT(6);  // prune read + push
  }
#undef M3_TT
  // Coordinate marks, read set and height range of the accepted cells (= L.order: every accepted entry belongs to one of
  // them), after the loop instead of ~25 instructions in every trip: a cell's moves look at its own column, the four
  // neighbour columns and the four landing columns two steps away.
  for (int i = c.lane; i < n_order; i += 64) {
    const int ci = L.order[i];
    const int z = ci / YX, q = ci - z * YX, y = q / c.X, x = q - y * c.X;
    mkl |= (x < 8 ? 1u << x : 0u) | (y < 8 ? 1u << y : 0u) | (1u << z);
    uint64_t m = 1ull << q;
    m |= x + 1 < c.X ? 1ull << (q + 1) : 0ull;
    m |= x + 2 < c.X ? 1ull << (q + 2) : 0ull;
    m |= x >= 1 ? 1ull << (q - 1) : 0ull;
    m |= x >= 2 ? 1ull << (q - 2) : 0ull;
    m |= y + 1 < c.Y ? 1ull << (q + c.X) : 0ull;
    m |= y + 2 < c.Y ? 1ull << (q + 2 * c.X) : 0ull;
    m |= y >= 1 ? 1ull << (q - c.X) : 0ull;
    m |= y >= 2 ? 1ull << (q - 2 * c.X) : 0ull;
    rs |= m;
    zlo = min(zlo, z);
    zhi = max(zhi, z);
  }
  mk = wave_or(mkl);
#ifdef PCGRL_PHASE_TIMING
  if (c.lane == 0) {
    L.dbg[0] += (uint32_t)dbg_trips;
    L.dbg[1] += (uint32_t)tail;
    L.dbg[2] += 1;
  }
#endif
  return tail;
}

// first maximum of len(path) in first-insertion order (helper_3D.py:538-541); returns the cell, sets entry id
__device__ inline int m3_farthest(const M3Lds &L, const M3Ctx &c, int n_order, int &entry) {
  uint32_t key = 0;  // len << 16 | (0xFFFF - k): max key = longest, earliest
  for (int k = c.lane; k < n_order; k += 64) {
    const uint32_t len = (L.best[L.order[k]] >> 12) & 0xFFFu;
    const uint32_t kk = (len << 16) | (uint32_t)(0xFFFF - k);
    key = kk > key ? kk : key;
  }
  key = wave_max(key);
  const int k = 0xFFFF - (int)(key & 0xFFFF);
  const int cell = L.order[k];
  entry = (int)(L.best[cell] & 0xFFFu);
  return cell;
}

// The pair of searches of one start candidate (helper_3D.py:527-553) -> slot s (result + read set).
__device__ inline void m3_fill_slot(M3Lds &L, const M3Ctx &c, int s, int sx, int sy, int sz, bool &overflow) {
  M3Slot &S = L.slot[s];
  if (c.lane == 0) S.valid = 0;
  int n_order = 0, e1 = 0, e2 = 0;
  uint32_t mk = 0, mk2 = 0;
  uint64_t rs = 0;
  int zlo = 15, zhi = 0;
#ifdef PCGRL_PHASE_TIMING
  uint64_t t0_ = __builtin_readcyclecounter();
#define M3_T(i)                                                     \
  do {                                                              \
    uint64_t t1_ = __builtin_readcyclecounter();                    \
    if (c.lane == 0) L.dbg[i] += (uint32_t)(t1_ - t0_);             \
    t0_ = t1_;                                                      \
  } while (0)
#else
#define M3_T(i) \
  do {          \
  } while (0)
#endif
  m3_search(L, c, sx, sy, sz, n_order, mk, rs, zlo, zhi, overflow);
  if (overflow) return;
  const int YX = c.Y * c.X;
  (void)m3_farthest(L, c, n_order, e1);
  const uint32_t f1 = L.ent[e1].y;
  m3_search(L, c, (int)((f1 >> 12) & 63u), (int)((f1 >> 18) & 63u), (int)((f1 >> 24) & 15u), n_order, mk2, rs, zlo, zhi, overflow);
  if (overflow) return;
  (void)m3_farthest(L, c, n_order, e2);
  // OR of the lanes' read sets (DPP inside the 16-lane rows, then across); min / max of the heights ride along as a
  // unary mask of the planes seen
  // OR of the lanes' read sets; min / max of the heights ride along as a unary mask of the planes seen
  const uint32_t lo = wave_or((uint32_t)rs), hi = wave_or((uint32_t)(rs >> 32));
  uint32_t zm = wave_or(zlo <= zhi ? ((2u << zhi) - (1u << zlo)) : 0u);
  // The tiles of paths[(mx,my,mz)] as a bit mask: every lane walks the parent chain (uniform reads), lane w keeps word
  // w of the mask.  An entry knows its move kind and direction, so the parent's cell and the intermediate tiles of the
  // move (helper_3D.py:214-319) follow without reading the parent: +-YX = one plane up / down.
  uint32_t my = 0;
  auto mark = [&](int cell) { my |= (cell >> 5) == c.lane ? 1u << (cell & 31) : 0u; };
  int id = e2;
  const uint2 fe = L.ent[e2];
  while (true) {
    const uint2 e = L.ent[id];
    const int ci = e.x & 511, kind = (e.x >> 9) & 7, d = (int)(e.y >> 28);
    mark(ci);
    if (kind == M3_ROOT) break;
    const int dq = (d == 0 ? 1 : (d == 2 ? -1 : 0)) + (d == 1 ? c.X : (d == 3 ? -c.X : 0));  // column step of the move
    const int mid = ci - dq;  // jumps: the jumped-over column, at the landing's height
    switch (kind) {
      case M3_DOWN: mark(ci + YX); break;                          // the target column at the parent's height
      case M3_UP: mark(ci - dq); break;                            // above the parent: (x, y, nz+1)
      case M3_JFLAT: mark(mid); break;                             // (nx, ny, nz)
      case M3_JUP: mark(mid - YX); mark(mid); break;               // (nx,ny,nz), (nx,ny,nz+1): landing is one higher
      case M3_JDOWN: mark(mid + YX); mark(mid); break;             // (nx,ny,nz), (nx,ny,nz-1): landing is one lower
      default: break;
    }
    id = (int)(e.x >> 20);
  }
  if (c.lane < M3_MAXW) S.pathm[c.lane] = my;
  if (c.lane == 0) {
    S.start = (uint8_t)(sy * c.X + sx);
    S.valid = 1;
    S.max_dist = (uint16_t)(fe.y & 0xFFFu);
    S.n_jump = (uint16_t)((fe.x >> 12) & 255u);
    S.mk = (uint8_t)(mk & ((1u << c.Z) - 1u));
    zm &= 0xFFu;
    S.zr = (uint8_t)((zm ? __builtin_ctz(zm) : 15) | ((zm ? 31 - __builtin_clz(zm) : 0) << 4));
    S.rs[0] = lo;
    S.rs[1] = hi;
  }
#undef M3_T
}

// helper_3D.calc_longest_path + remove_stacked_path_tiles + minecraft_3D_maze_prob.get_stats
// air: this lane's plane (lanes < Z).  Results uniform over the wave.  L.over receives the new overlay mask.
// Slots that are still valid (see SLOT CACHE) are reused; pass fresh = true to ignore them (reset, caller-provided maps).
__device__ inline void m3_stats(M3Lds &L, const M3Ctx &c, uint64_t air, int32_t *st, bool &overflow, bool fresh PHASE_ARG) {
  const int YX = c.Y * c.X;
  // per-(y,x) column masks for the move rules: lane q collects bit q of every plane
  {
    uint32_t m = 0;
#pragma unroll
    for (int z = 0; z < 8; z++) {
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)air, z, 64), hi = (uint32_t)__shfl((int)(uint32_t)(air >> 32), z, 64);
      const uint64_t a = (uint64_t)lo | ((uint64_t)hi << 32);
      if (z < c.Z) m |= (uint32_t)((a >> (c.lane & 63)) & 1ull) << z;
    }
    L.col[c.lane] = c.lane < YX ? (uint8_t)m : (uint8_t)0;
  }
  if (fresh && c.lane < M3_SLOTS) L.slot[c.lane].valid = 0;
  PHASE_MARK(2);  // column masks
  st[0] = m3_regions(c, air);
  PHASE_MARK(3);  // regions
  // start candidates per plane: AIR with head-room, standing on something, z >= 1 (:520-526)
  const uint64_t above = dpp64_down(air), below = dpp64_up(air);
  const uint64_t cand = (c.lane >= 1 && c.lane + 1 < c.Z) ? (air & above & ~below) : 0ull;
  uint32_t marked = 0;  // z-planes of final_visited_map that are fully set (the fancy-index bug, :531)
  int final_value = 0, n_jump = 0, best_slot = -1;
  while (true) {
    const bool mine = c.lane < c.Z && cand != 0 && !((marked >> c.lane) & 1u);
    const uint64_t b = __ballot(mine);
    if (b == 0) break;
    const int sz = __builtin_ctzll(b);
    const int bit = (int)__shfl((int)__builtin_ctzll(cand | (1ull << 63)), sz, 64);
    const int s = sz - 1;
    const M3Slot &S = L.slot[s];
    if (!(S.valid && S.start == bit)) {
      const int sy = bit / c.X, sx = bit - sy * c.X;
      m3_fill_slot(L, c, s, sx, sy, sz, overflow);
      if (overflow) break;
    }
    marked |= S.mk;
    n_jump = S.n_jump;  // :553 overwritten by every processed component
    if ((int)S.max_dist > final_value) {
      final_value = S.max_dist;
      best_slot = s;
    }
  }
  PHASE_MARK(4);  // path searches
  // remove_stacked_path_tiles (:657-675) then the transposed overlay of process_observation (:84-93):
  // path tile (x,y,z) is drawn at array index [x][y][z]
  for (int i = c.lane; i < c.nw + 2; i += 64) {
    L.pathm[i] = (best_slot >= 0 && i < M3_MAXW) ? L.slot[best_slot < 0 ? 0 : best_slot].pathm[i] : 0u;
    L.over[i] = 0;
  }
  {
    const int q = c.lane, y = q / c.X, x = q - y * c.X;  // this lane's column (lanes < YX)
    for (int z = 0; z < c.Z; z++) {
      const int ci = z * YX + q;
      const bool in = q < YX && m3_bit(L.pathm, ci) && !(z > 0 && m3_bit(L.pathm, ci - YX));
      if (in && x < c.Z && z < c.X) {
        const int oi = (x * c.Y + y) * c.X + z;
        atomicOr(&L.over[oi >> 5], 1u << (oi & 31));
      }
    }
  }
  st[1] = final_value;
  st[2] = n_jump;
  PHASE_MARK(5);  // overlay post-processing
}

// observation: (o0, o1, o2, 4) uint8, channel 0 = out of bounds, 1 = AIR, 2 = DIRT, 3 = path overlay
__device__ inline void m3_encode_obs(const uint32_t *dirt, const uint32_t *over, const M3Ctx &c, const Params &p, int env, const int *pos,
                                     bool show_path, uint8_t *obs_base = nullptr) {
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
        v = 1 + (int)m3_bit(dirt, ci);
        if (show_path && m3_bit(over, ci)) v = 3;
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

// reset from the env's RNG streams (envs/pcgrl_env.py:158-188; probabilities, then the map in (z,y,x) order) into `dirt`.
// rp / rr: the env's streams (in registers); advanced.
__device__ inline void m3_reset_rng(uint32_t *dirt, const M3Ctx &c, const Params &p, int cpl, Pcg &rp, Pcg &rr) {
  double p0 = rp.next_double(), p1 = rp.next_double();
  double total = 0.0;
  total += p0;
  total += p1;
  double c0 = p0 / total, c1 = c0 + p1 / total;
  c0 /= c1;  // cdf /= cdf[-1]
  for (int i = c.lane; i < c.nw + 2; i += 64) dirt[i] = 0;
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
    if (ci < c.n_cells && ((bits >> k) & 1u)) atomicOr(&dirt[ci >> 5], 1u << (ci & 31));
  }
  rr = end;
}

// bytes -> bit string (caller-provided maps)
__device__ inline void m3_load_bytes(uint32_t *dirt, const M3Ctx &c, const uint8_t *src) {
  for (int i = c.lane; i < c.nw + 2; i += 64) dirt[i] = 0;
  for (int ci = c.lane; ci < c.n_cells; ci += 64)
    if (src[ci]) atomicOr(&dirt[ci >> 5], 1u << (ci & 31));
}

// M3_ROLLOUT: pcgrl_rollout, p.n_steps steps per launch with the env state in LDS / registers (one wave does both roles)
enum M3Mode { M3_STEP = 0, M3_RESET = 1, M3_OBSERVE = 2, M3_STATS_FOR_GRIDS = 3, M3_GET_STATE = 4, M3_ROLLOUT = 5 };

// narrow_rep.py:89-102 with Q1 (position of the NEXT edit from the pre-increment counter)
__device__ inline void m3_advance_pos(const M3Ctx &c, int *pos, int &n_step) {
  const int YX = c.Y * c.X;
  const int idx = n_step % c.n_cells;
  pos[0] = idx / YX;
  pos[1] = (idx / c.X) % c.Y;
  pos[2] = idx % c.X;
  n_step++;
}

// D7: the BASELINE map shape 7 x 7 x 7 with compile-time dimensions (the search trip is instruction-bound: constant
// strides and bounds take a fifth of its instructions away)
template <int MODE, bool D7 = false>
__global__ __launch_bounds__(MODE == M3_STEP ? 128 : 64) void m3_kernel(Params p, int cpl) {
  __shared__ M3Lds L;
  __shared__ M3ObsLds O;
  M3Ctx c;
  c.lane = (int)__lane_id();
  c.Z = D7 ? 7 : p.cfg.dims[0];
  c.Y = D7 ? 7 : p.cfg.dims[1];
  c.X = D7 ? 7 : p.cfg.dims[2];
  if (D7) cpl = 6;  // ceil(343 / 64)
  c.n_cells = c.Z * c.Y * c.X;
  c.nw = (c.n_cells + 31) >> 5;
  const int env = blockIdx.x;
  constexpr int NS = M3_NS;
  PHASE_DECL();
  TRACE_DECL();
  uint32_t *gd = (uint32_t *)p.planes + (size_t)env * 2 * M3_MAXW;  // [dirt words | overlay words]
  uint32_t *gslot = (uint32_t *)p.m3cache + (size_t)env * M3_SLOT_WORDS;
  EnvState *S = &p.st[env];

  if constexpr (MODE == M3_STEP) {
    // ------------------------------------------------------------------------------------------ observe wave
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0) {  // (readfirstlane: a scalar branch)
      if (p.obs == nullptr) return;  // (the simulate wave skips the barrier in that case, too)
      for (int i = c.lane; i < c.nw + 2; i += 64) {
        O.dirt[i] = i < c.nw ? gd[i] : 0u;
        O.over[i] = i < c.nw ? gd[M3_MAXW + i] : 0u;
      }
      int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};
      int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
      const int action = p.actions[env];
      const bool upd_only = p.update_only != 0;
      // an auto-reset replays the env's RNG streams in both waves; this wave takes its copy before the barrier
      Pcg rp, rr;
      if (p.auto_reset != 0 && (iteration + 1 > p.cfg.max_iterations || p.cfg.max_changes >= 0)) {
        rp.load(p.rng[env].prob);
        rr.load(p.rng[env].rep);
      }
      __syncthreads();  // both waves hold the old state before wave 0 may overwrite it
      iteration += upd_only ? 0 : 1;
      bool change = false;
      if (action >= 0 && action < 2) {
        const int ci = m3_cell(c, pos[2], pos[1], pos[0]);  // pos = (z, y, x)
        change = m3_bit(O.dirt, ci) != (action != 0);
        if (change && c.lane == 0) O.dirt[ci >> 5] ^= 1u << (ci & 31);
        m3_advance_pos(c, pos, n_step);
      }
      changes += (change && !upd_only) ? 1 : 0;
      bool done = !upd_only && iteration > p.cfg.max_iterations;
      if (p.cfg.max_changes >= 0) done = done || (!upd_only && changes > p.cfg.max_changes);
      if (done && p.auto_reset != 0) {  // first observation of the new episode: no overlay (PcgrlEnv.reset)
        m3_reset_rng(O.dirt, c, p, cpl, rp, rr);
        pos[0] = pos[1] = pos[2] = 0;
        m3_encode_obs(O.dirt, O.over, c, p, env, pos, false);
      } else {
        m3_encode_obs(O.dirt, O.over, c, p, env, pos, true);
      }
      TRACE_PUT(3, TRACE_NOW());
      return;
    }
    __builtin_amdgcn_s_setprio(3);  // the simulate wave's dependent chain issues ahead of the observe wave on its SIMD
  }

  if constexpr (MODE == M3_STATS_FOR_GRIDS) {
    m3_load_bytes(L.dirt, c, p.init_grids + (size_t)env * c.n_cells);
    for (int i = c.lane; i < c.n_cells; i += 64) {
      L.best[i] = 0;
      L.claim[i] = 0xFFFFFFFFu;
    }
    if (c.lane == 0) L.epoch = 0;
    uint64_t air = c.lane < c.Z ? m3_plane_air(L.dirt, c, c.lane) : 0ull;
    int32_t st[NS];
    bool ovf = false;
    m3_stats(L, c, air, st, ovf, true PHASE_PASS);
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

  // load grid + overlay (+ the slot cache) and prepare the search tables
  for (int i = c.lane; i < c.nw + 2; i += 64) {
    L.dirt[i] = i < c.nw ? gd[i] : 0u;
    L.over[i] = i < c.nw ? gd[M3_MAXW + i] : 0u;
  }
  int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};

  if constexpr (MODE == M3_OBSERVE) {
    // reset()/observe(): no path overlay (PcgrlEnv.reset does not call process_observation)
    m3_encode_obs(L.dirt, L.over, c, p, env, pos, false);
    return;
  }
  if constexpr (MODE != M3_RESET) {
    for (int i = c.lane; i < M3_SLOT_WORDS; i += 64) ((uint32_t *)L.slot)[i] = gslot[i];
  }
  for (int i = c.lane; i < c.n_cells; i += 64) {
    L.best[i] = 0;
    L.claim[i] = 0xFFFFFFFFu;
  }
  if (c.lane == 0) L.epoch = 0;
#ifdef PCGRL_PHASE_TIMING
  if (c.lane < 8) L.dbg[c.lane] = 0;
#endif

  int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
  double last_loss = S->last_loss, ep_return = S->ep_return;
  int32_t st[NS];
  for (int k = 0; k < NS; k++) st[k] = S->stats[k];
  bool ovf = false, slots_dirty = false;
  EnvTargets<NS> trg;
  trg.load(p, env, false);
  Pcg rp, rr;

  if constexpr (MODE == M3_RESET) {
    if (p.mask != nullptr && p.mask[env] == 0) return;
    if (p.refresh_only) {  // statistics (and the path overlay) of the current map, nothing else
      uint64_t air0 = c.lane < c.Z ? m3_plane_air(L.dirt, c, c.lane) : 0ull;
      m3_stats(L, c, air0, st, ovf, true PHASE_PASS);
      if (ovf && c.lane == 0) atomicOr(p.err, 4);
      for (int i = c.lane; i < c.nw; i += 64) gd[M3_MAXW + i] = L.over[i];
      for (int i = c.lane; i < M3_SLOT_WORDS; i += 64) gslot[i] = ((uint32_t *)L.slot)[i];
      if (c.lane == 0) {
        S->last_loss = trg.loss(p.cfg, st);
        S->flags = 0;
        for (int k = 0; k < NS; k++) {
          S->stats[k] = st[k];
          if (p.stats_out) p.stats_out[(size_t)env * NS + k] = st[k];
        }
      }
      return;
    }
    if (p.init_grids) {
      m3_load_bytes(L.dirt, c, p.init_grids + (size_t)env * c.n_cells);
      pos[0] = pos[1] = pos[2] = 0;
      if (p.init_pos)
        for (int d = 0; d < 3; d++) pos[d] = p.init_pos[(size_t)env * 3 + d];
    } else {
      rp.load(p.rng[env].prob);
      rr.load(p.rng[env].rep);
      m3_reset_rng(L.dirt, c, p, cpl, rp, rr);
      if (c.lane == 0) {
        rr.store(p.rng[env].rep);
        rp.store(p.rng[env].prob);
      }
      pos[0] = pos[1] = pos[2] = 0;
    }
    uint64_t air = c.lane < c.Z ? m3_plane_air(L.dirt, c, c.lane) : 0ull;
    m3_stats(L, c, air, st, ovf, true PHASE_PASS);
    slots_dirty = true;
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
   rp.load(p.rng[env].prob);  // (only advanced by an auto-reset)
   rr.load(p.rng[env].rep);
   bool any_reset = false;
   if (MODE == M3_STEP && p.obs != nullptr) __syncthreads();  // the observe wave has taken its copy of the old state
   for (int k = 0; k < K; k++) {
    const size_t o = (size_t)k * N + (size_t)env;  // index of this step's outputs
    uint8_t *obs_k = p.obs == nullptr ? nullptr
                     : (MODE == M3_ROLLOUT && !p.obs_last_only ? p.obs + (size_t)k * N * (size_t)p.obs_env_bytes : p.obs);
    const bool want_obs = MODE == M3_ROLLOUT && (!p.obs_last_only || k == K - 1);  // (M3_STEP: the observe wave)
    // ---- step (envs/pcgrl_env.py:267-342 with narrow_rep.py:89-102)
    const int action = p.actions[o];
    const bool bad = action < 0 || action >= 2;
    const bool upd_only = p.update_only != 0;
    iteration += upd_only ? 0 : 1;
    bool change = false;
    if (!bad) {
      const int ci = m3_cell(c, pos[2], pos[1], pos[0]);  // pos = (z, y, x)
      const bool old = m3_bit(L.dirt, ci);
      change = old != (action != 0);
      if (change) {
        if (c.lane == 0) L.dirt[ci >> 5] ^= 1u << (ci & 31);
        // the edit invalidates exactly the cached slots whose searches read this cell
        const int q = pos[1] * c.X + pos[2];
        if (c.lane < M3_SLOTS) {
          const M3Slot &T = L.slot[c.lane];
          const int zr = T.zr;
          if (((T.rs[q >> 5] >> (q & 31)) & 1u) && pos[0] >= (zr & 15) - 2 && pos[0] <= (zr >> 4) + 3) L.slot[c.lane].valid = 0;
        }
        slots_dirty = true;
      }
      m3_advance_pos(c, pos, n_step);
    } else if (c.lane == 0) {
      atomicOr(p.err, 1);
    }
    if (upd_only) {  // rep.update() only: map, position (the observe wave shows the stale overlay)
      for (int i = c.lane; i < c.nw; i += 64) gd[i] = L.dirt[i];
      if (slots_dirty)
        for (int i = c.lane; i < M3_SLOT_WORDS; i += 64) gslot[i] = ((uint32_t *)L.slot)[i];
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
    if (!do_reset && want_obs) m3_encode_obs(L.dirt, L.over, c, p, env, pos, true, obs_k);
    PHASE_MARK(1);  // observation
    if (change) {
      uint64_t air = c.lane < c.Z ? m3_plane_air(L.dirt, c, c.lane) : 0ull;
      m3_stats(L, c, air, st, ovf, false PHASE_PASS);
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
      if (c.lane == 0) {
        latch_episode<NS>(p, env, S, ep_return, iteration, st);
        accumulate_episode<NS>(S);
      }
      m3_reset_rng(L.dirt, c, p, cpl, rp, rr);
      any_reset = true;
      pos[0] = pos[1] = pos[2] = 0;
      uint64_t air = c.lane < c.Z ? m3_plane_air(L.dirt, c, c.lane) : 0ull;
      m3_stats(L, c, air, st, ovf, true PHASE_PASS);
      slots_dirty = true;
      n_step = iteration = changes = 0;
      ep_return = 0.0;
      trg.load(p, env, true);
      last_loss = trg.loss(p.cfg, st);
      if (want_obs) m3_encode_obs(L.dirt, L.over, c, p, env, pos, false, obs_k);
    }
   }
   if (any_reset && c.lane == 0) {
     rr.store(p.rng[env].rep);
     rp.store(p.rng[env].prob);
   }
  }
  if (ovf && c.lane == 0) atomicOr(p.err, 4);
  // write back
  for (int i = c.lane; i < c.nw; i += 64) {
    gd[i] = L.dirt[i];
    gd[M3_MAXW + i] = L.over[i];
  }
  if (slots_dirty)
    for (int i = c.lane; i < M3_SLOT_WORDS; i += 64) gslot[i] = ((uint32_t *)L.slot)[i];
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
#ifdef PCGRL_PHASE_TIMING
  _ph[0] = L.dbg[0];  // (development: trips / queue entries / searches of this launch replace the first phases)
  _ph[1] = L.dbg[6];  // (farthest 2 + path materialisation)
  _ph[2] = L.dbg[2];
  _ph[3] = L.dbg[3];  // cycles: first search, farthest, second search, farthest + path materialisation
  _ph[5] = L.dbg[4];
  _ph[6] = L.dbg[5];
#endif
  PHASE_FLUSH();
  TRACE_PUT(0, _tr0);
  TRACE_PUT(1, TRACE_NOW());
  TRACE_DRAIN();
  TRACE_PUT(2, TRACE_NOW());
}

}  // namespace pcgrl
