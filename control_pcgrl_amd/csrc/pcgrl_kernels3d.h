// pcgrl_kernels3d.h -- gfx950 kernels for minecraft_3D_maze (narrow representation).
//
// Reference (paths relative to control_pcgrl/): envs/probs/minecraft/minecraft_3D_maze_prob.py:143-181 get_stats,
// :84-93 process_observation; envs/helper_3D.py: _passable :214-319, _flood_fill :354-383, calc_num_regions :396-406,
// run_dijkstra :422-490, calc_longest_path :503-563, remove_stacked_path_tiles :657-675; envs/pcgrl_env.py:267-342.
// Map sizes: configs/config.py:153-157 (the stock 15 x 15 x 15) and BASELINE's 7 x 7 x 7.
//
// One workgroup per env.  pcgrl_step runs specialised wavefronts over the env (like the 2-D step kernel):
//   wave 0 "simulate"  action -> move-table update -> path searches -> reward / done -> auto-reset -> state write-back
//   wave 1 "observe"   replays the (trivial) action / reset on its own copy and streams the observation, which shows the
//                      path overlay of the PREVIOUS statistics update (pcgrl_env.py:298-299 vs :314-323), so it does
//                      not depend on this step's searches.
//   wave 2 "helper"    (planes of <= 64 cells) counts the regions and runs the second search of a pair speculatively while
//                      wave 0 runs the first (see SPECULATION); jobs through an LDS mailbox.
// The only barrier of the step sits at the END of the waves (the simulate wave overwrites the state only after the observe
// wave has read it), so no wave's work waits for another's loads.
// Lane roles inside the simulate wave:
//   lanes 0..Z-1   one z-plane each as a (Y*X)-bit mask (1 or 4 64-bit words: size class SC): 6-neighbour flood fill =
//                  shifts by 1 / X inside the lane and a DPP row_shr/row_shl between planes; start candidates
//   lane 4*i + d   move direction d of queue entry i of the current trip of the path search (16 entries per trip)
//   all 64 lanes   move-table maintenance, farthest-cell arg-max, overlay post-processing, reset RNG
//
// MOVE TABLE.  helper_3D._passable (:214-319) is a pure function of the map around a foothold: for every (cell, direction)
// at most one of its six rules applies.  The engine keeps that function as a table in HBM (16 bits per cell and
// direction: path cost, jump flag, height change, target cell - cell; 0 = no move) and MAINTAINS it: an edit of one
// cell can change the moves of at most 48 (cell, direction) pairs -- the cells whose rules read the edited cell -- which 48 lanes re-evaluate at
// once.  The path search then reads one entry per popped queue entry and direction instead of evaluating the rules.
//
// PATH SEARCH.  helper_3D.run_dijkstra is a FIFO label-correcting search whose pop order decides n_jump, the farthest
// cell and the path drawn into the next observation, so the queue order is kept exactly (see m3_search).
//
// SLOT CACHE.  calc_longest_path (:503-563) starts one pair of searches per start candidate, and its whole-plane visited
// marking (:531) leaves at most ONE processed candidate per z-plane: the first candidate of the plane in (y, x) order.
// Everything a plane's pair of searches produces -- the marks, max_dist, n_jump, the path tiles -- is a function of the
// start cell and of the move-table rows of the cells the searches ACCEPTED (nothing else of the map is read).  The engine
// keeps, per env and plane, that result together with the set of accepted cells in the env's record in HBM.  A step edits one cell: a slot is
// dropped iff a move-table entry of one of its accepted cells actually changed; every other slot is still exact, and the
// sequential candidate walk re-runs only the searches it needs.
#pragma once
#include <hip/hip_runtime.h>

#include "pcgrl_common.h"
#include "pcgrl_kernels2d.h"

namespace pcgrl {

// Development builds (-DPCGRL_PHASE_TIMING): the eight per-workgroup counters hold either the search counters
// ([0] trips [1] queue entries [2] searches [3] search-loop cycles [4] regions [5] everything else [6] candidate walk incl.
// the search loops) or, with -DPCGRL_M3_PHASES, the phases of the simulate wave ([0] loads until the barrier [1] columns +
// move-table update [2] regions [3] candidate walk [4] overlay [5] outputs + write-back [6] fresh tables); [7] = wall clock.
#if defined(PCGRL_M3_TAIL)  // simulate wave (tools/tail_3d_walk.py): [0] cached start planes missing when the walk began [1] pairs of
// searches it ran [3] candidate-walk cycles [5] speculative second searches used [6] everything else
#define M3_MARK(coarse, fine) PHASE_MARK((coarse) == 3 ? 3 : 6)
#elif defined(PCGRL_M3_SPEC)  // [0] pairs run [1] with a remembered farthest cell [2] of which the helper's search was used
#define M3_MARK(coarse, fine) PHASE_MARK(6)
#elif defined(PCGRL_M3_TRIPS)  // [0] chain trips [1] general trips [2] / [3] their cycles; the first four phases land in [6]
#define M3_MARK(coarse, fine) PHASE_MARK((coarse) < 4 ? 6 : (coarse))
#elif defined(PCGRL_M3_TAILSPLIT)  // [0] write-back + state stores [1] everything before the walk [2] wait for the helper's region count
// [3] walk [4] overlay [5] loss + outputs [6] closing barrier
#define M3_MARK(coarse, fine) PHASE_MARK((coarse) == 3 ? 3 : ((coarse) == 4 ? 4 : ((coarse) == 5 ? 0 : 1)))
#elif defined(PCGRL_M3_HEADSPLIT)  // [0] loads + LDS fill [1] the edit [2] move-table update + slot-drop test [4] position, observation
// (rollout) [5] region job posted, planes [3] walk [6] everything after it
#define M3_MARK(coarse, fine) PHASE_MARK((coarse) == 0 ? 0 : ((coarse) == 1 ? 4 : ((coarse) == 2 ? 5 : ((coarse) == 3 ? 3 : 6))))
#elif defined(PCGRL_M3_PHASES)
#define M3_MARK(coarse, fine) PHASE_MARK(coarse)
#else
#define M3_MARK(coarse, fine) PHASE_MARK(fine)
#endif

// size classes: SC 0 = planes of up to 64 cells, Z <= 8 (BASELINE's 7^3);  SC 1 = planes of up to 256 cells, Z <= 16 (15^3)
template <int SC>
struct M3C;
template <>
struct M3C<0> {
  static constexpr int CELLS = 512, NW = 16, PW = 1, RING = 512, COLS = 64, SLOTS = 6, ZMAX = 8;
  static constexpr int REC = 2 * NW + COLS / 2 + 2 + SLOTS * (4 + 2 * NW) + 2 * CELLS + 4;  // upper bound of m3_layout().rec_words
};
template <>
struct M3C<1> {
  static constexpr int CELLS = 4096, NW = 128, PW = 4, RING = 4096, COLS = 256, SLOTS = 14, ZMAX = 16;
  static constexpr int REC = 2 * NW + COLS / 2 + 2 + SLOTS * (4 + 2 * NW) + 2 * CELLS + 4;
};
constexpr int M3_NS = 3;
constexpr int M3_SLOT_HDR = 4;  // header words of a cached slot in HBM, followed by the accepted-cell set and the path tiles

__host__ __device__ inline int m3_size_class(int Z, int Y, int X) { return (Z <= 8 && Y * X <= 64 && Z * Y * X <= 512) ? 0 : 1; }
__host__ __device__ inline bool m3_supported(int Z, int Y, int X) {
  return Z >= 1 && Y >= 1 && X >= 1 && Z <= 16 && Y <= 16 && X <= 16 && Y * X <= 256 && Z * Y * X <= 4096;
}
// words of one per-cell bit string of an env in HBM (even, so that 64-bit accesses stay aligned)
__host__ __device__ inline int m3_words(int n_cells) { return (((n_cells + 31) >> 5) + 1) & ~1; }
// One env = one contiguous record of 32-bit words in HBM (Params::planes), mirrored word for word in LDS, so a step reads
// it with a handful of 16-byte loads issued together:
//   [0, nw)            tile bits (1 = DIRT), flat cell index (z*Y + y)*X + x
//   [o_over, +nw)      path-overlay bits of the last statistics update (transposed index, see m3_stats)
//   [o_col, +cw)       per-(y,x) column masks: AIR bits over z, 16 bits per column
//   [o_slots, ...)     Z-2 cached start-plane results: 4 header words + accepted cells (nw) + path tiles (nw) each
//   [o_mv, +2*n_cells) the move table, 4 x 16 bits (directions) per cell
struct M3Lay {
  int nw, n_slots, slot_words, o_over, o_col, o_slots, o_mv, rec_words;
};
__host__ __device__ inline M3Lay m3_layout(int Z, int Y, int X) {
  M3Lay L;
  const int n_cells = Z * Y * X;
  L.nw = m3_words(n_cells);
  L.n_slots = Z > 2 ? Z - 2 : 0;  // start planes z = 1 .. Z-2
  L.slot_words = M3_SLOT_HDR + 2 * L.nw;
  L.o_over = L.nw;
  L.o_col = 2 * L.nw;
  L.o_slots = L.o_col + ((((Y * X + 1) >> 1) + 1) & ~1);
  L.o_mv = L.o_slots + L.n_slots * L.slot_words;
  L.rec_words = (L.o_mv + 2 * n_cells + 3) & ~3;
  return L;
}

// ---------------------------------------------------------------------------------------------- plane masks
template <int PW>
struct PM {
  uint64_t w[PW];
};
template <int PW>
__device__ inline PM<PW> pm_zero() {
  PM<PW> r;
#pragma unroll
  for (int i = 0; i < PW; i++) r.w[i] = 0;
  return r;
}
template <int PW>
__device__ inline PM<PW> operator&(PM<PW> a, PM<PW> b) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] &= b.w[i];
  return a;
}
template <int PW>
__device__ inline PM<PW> operator|(PM<PW> a, PM<PW> b) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] |= b.w[i];
  return a;
}
template <int PW>
__device__ inline PM<PW> operator~(PM<PW> a) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] = ~a.w[i];
  return a;
}
template <int PW>
__device__ inline bool pm_any(PM<PW> a) {
  uint64_t o = 0;
#pragma unroll
  for (int i = 0; i < PW; i++) o |= a.w[i];
  return o != 0;
}
template <int PW>
__device__ inline PM<PW> pm_shl(PM<PW> a, int s) {  // 0 < s < 64
  PM<PW> r;
#pragma unroll
  for (int i = PW - 1; i >= 0; i--) r.w[i] = (a.w[i] << s) | (i > 0 ? a.w[i - 1] >> (64 - s) : 0ull);
  return r;
}
template <int PW>
__device__ inline PM<PW> pm_shr(PM<PW> a, int s) {
  PM<PW> r;
#pragma unroll
  for (int i = 0; i < PW; i++) r.w[i] = (a.w[i] >> s) | (i + 1 < PW ? a.w[i + 1] << (64 - s) : 0ull);
  return r;
}
template <int PW>
__device__ inline PM<PW> pm_lowest(PM<PW> a) {  // isolate the lowest set bit
  PM<PW> r = pm_zero<PW>();
  bool done = false;
#pragma unroll
  for (int i = 0; i < PW; i++) {
    r.w[i] = done ? 0ull : (a.w[i] & (0ull - a.w[i]));
    done = done || a.w[i] != 0;
  }
  return r;
}
template <int PW>
__device__ inline int pm_ctz(PM<PW> a) {  // index of the lowest set bit (64 * PW if none)
  int r = 64 * PW;
#pragma unroll
  for (int i = PW - 1; i >= 0; i--) r = a.w[i] ? 64 * i + __builtin_ctzll(a.w[i]) : r;
  return r;
}
template <int PW>
__device__ inline void pm_set(PM<PW> &a, int q) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] |= (q >> 6) == i ? 1ull << (q & 63) : 0ull;
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
template <int PW>
__device__ inline PM<PW> pm_up(PM<PW> a) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] = dpp64_up(a.w[i]);
  return a;
}
template <int PW>
__device__ inline PM<PW> pm_down(PM<PW> a) {
#pragma unroll
  for (int i = 0; i < PW; i++) a.w[i] = dpp64_down(a.w[i]);
  return a;
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

// ---------------------------------------------------------------------------------------------- LDS
// header of a cached start-plane result (see SLOT CACHE)
struct M3SlotHdr {
  uint16_t start;   // bit index (y*X + x) of the start cell in its plane
  uint16_t valid;
  uint16_t max_dist;
  uint16_t n_jump;
  uint32_t mk;      // z-planes marked visited by the first search (the fancy-index bug, :531)
  uint32_t far1;    // farthest cell of the first search + 1 (0: unknown): where the second search started (see SPECULATION)
};
static_assert(sizeof(M3SlotHdr) == 4 * M3_SLOT_HDR, "slot header layout");

// workspace of one search wave
template <int SC>
struct M3Work {
  uint2 ent[M3C<SC>::RING];         // queue ring: cell | parent cell<<12 (0x1FFF: none) | move code<<25 | direction<<30 ; len
  uint2 best[M3C<SC>::CELLS];       // per cell: epoch<<24 | len of the accepted path (the `paths` dict) ; trip claim
  uint32_t info[M3C<SC>::CELLS];    // per cell, of the accepted entry: parent cell | move code<<13 | direction<<18
  uint16_t order[M3C<SC>::CELLS];   // cells in first-insertion order
  uint32_t racc[M3C<SC>::NW];       // accepted cells of the pair of searches being run (bit per cell)
};
// the env: its record (M3Lay)
template <int SC>
struct M3Env {
  alignas(16) uint32_t rec[M3C<SC>::REC + 4];
};
// observe wavefronts of a pcgrl_step workgroup.  Size class 1 keeps 147 KB of LDS per env, i.e. ONE workgroup per CU, and
// its observation is 108 KB at 15^3: four waves share it, a contiguous quarter each (16.7 us per env on one wave).
template <int SC>
constexpr int m3_observers() { return SC == 0 ? 1 : 4; }
template <int SC>
struct M3ObsLds {  // each observe wave's own copy of the tile and overlay bits + the (shared) row masks of the encoder
  alignas(16) uint32_t bits[m3_observers<SC>()][2 * M3C<SC>::NW + 2];
  uint2 rows[SC == 0 ? 16 * 16 + 1 : 32 * 32 + 1];
};

struct M3Ctx {
  int lane, Z, Y, X, YX, n_cells;
#ifdef PCGRL_PHASE_TIMING
  int knob;  // development (timing builds): cfg.solver_power, see m3_update_moves
#endif
  M3Lay L;
  // views into the env's record in LDS
  uint32_t *dirt, *over, *slots;
  uint16_t *col;
  int16_t *mv;  // [cell][direction]
  __device__ inline M3SlotHdr *hdr(int s) const { return (M3SlotHdr *)(slots + s * L.slot_words); }
  __device__ inline uint32_t *racc(int s) const { return slots + s * L.slot_words + M3_SLOT_HDR; }
  __device__ inline uint32_t *spath(int s) const { return slots + s * L.slot_words + M3_SLOT_HDR + L.nw; }
};

__device__ inline bool m3_bit(const uint32_t *w, int i) { return (w[i >> 5] >> (i & 31)) & 1u; }
__device__ inline int m3_cell(const M3Ctx &c, int x, int y, int z) { return (z * c.Y + y) * c.X + x; }
// number of set bits of `mask` in lanes below this one
__device__ inline int m3_below(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
#define M3_BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)

// (Y*X)-bit AIR mask of plane z from the flat bit string.  Words past the string are whatever follows it in the record:
// the plane mask removes them.
template <int PW, bool INVERT = true>
__device__ inline PM<PW> m3_plane_air(const uint32_t *dirt, const M3Ctx &c, int z) {
  PM<PW> r;
  const int b0 = z * c.YX;
#pragma unroll
  for (int k = 0; k < PW; k++) {
    const int b = b0 + 64 * k, w = b >> 5, s = b & 31;
    const uint64_t lo = (uint64_t)dirt[w] | ((uint64_t)dirt[w + 1] << 32);
    uint64_t v = lo >> s;
    if (s) v |= (uint64_t)dirt[w + 2] << (64 - s);
    const int left = c.YX - 64 * k;  // plane bits in this word and beyond
    const uint64_t pm = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
    r.w[k] = (INVERT ? ~v : v) & pm;
  }
  return r;
}
// plane z of any per-cell bit string as it is (set bits, not AIR = clear bits)
template <int PW>
__device__ inline PM<PW> m3_plane_bits(const uint32_t *bits, const M3Ctx &c, int z) { return m3_plane_air<PW, false>(bits, c, z); }

// bits [start, start + len) of a plane mask, len < 64
template <int PW>
__device__ inline void pm_set_range(PM<PW> &a, int start, int len) {
#pragma unroll
  for (int i = 0; i < PW; i++) {
    const int lo = start - 64 * i, hi = lo + len;  // range in this word's coordinates
    const int l = lo < 0 ? 0 : lo, h = hi > 64 ? 64 : hi;
    if (l < h) a.w[i] |= (h - l >= 64 ? ~0ull : ((1ull << (h - l)) - 1ull)) << l;
  }
}
// plane bits whose x is not 0 / not X-1
template <int PW>
__device__ inline void m3_edge_masks(const Params &p, PM<PW> &notx0, PM<PW> &notxl) {
#pragma unroll
  for (int i = 0; i < PW; i++) {  // constants of the map shape, computed once by pcgrl_create (m3_edge_masks_host)
    notx0.w[i] = p.m3_notx0[i];
    notxl.w[i] = p.m3_notxl[i];
  }
}
inline void m3_edge_masks_host(int Y, int X, uint64_t notx0[4], uint64_t notxl[4]) {
  for (int i = 0; i < 4; i++) notx0[i] = notxl[i] = 0;
  for (int y = 0; y < Y; y++)
    for (int x = 0; x < X; x++) {
      const int q = y * X + x;
      if (q >= 256) continue;
      if (x != 0) notx0[q >> 6] |= 1ull << (q & 63);
      if (x != X - 1) notxl[q >> 6] |= 1ull << (q & 63);
    }
}

// the 6-neighbourhood of a set of cells (lane = plane)
template <int PW>
__device__ inline PM<PW> m3_grow(const M3Ctx &c, PM<PW> f, PM<PW> notx0, PM<PW> notxl) {
  return pm_shl(f & notxl, 1) | pm_shr(f & notx0, 1) | pm_shl(f, c.X) | pm_shr(f, c.X) | pm_up(f) | pm_down(f);
}

// helper_3D.py:396-406 calc_num_regions (6-neighbour components of AIR)
template <int PW>
__device__ inline int m3_regions(const M3Ctx &c, PM<PW> air, PM<PW> notx0, PM<PW> notxl) {
  PM<PW> remaining = c.lane < c.Z ? air : pm_zero<PW>();
  int n = 0;
  while (true) {
    uint64_t b = M3_BALLOT(pm_any(remaining));
    if (b == 0) break;
    int fl = __builtin_ctzll(b);
    PM<PW> f = c.lane == fl ? pm_lowest(remaining) : pm_zero<PW>();
    while (true) {  // two expansion rounds per trip (one ballot per two rounds)
      f = f | (m3_grow(c, f, notx0, notxl) & remaining);
      const PM<PW> nf = m3_grow(c, f, notx0, notxl) & remaining & ~f;
      if (M3_BALLOT(pm_any(nf)) == 0) break;
      f = f | nf;
    }
    remaining = remaining & ~f;
    n++;
  }
  return n;
}

// The region count after the edit of one cell e = (plane ez, bit eq), from the count before it.  A = the AIR planes
// WITHOUT e (the map before e became AIR / after it became DIRT).  The AIR face neighbours of e fall into `pieces`
// components of A: a new AIR cell joins them into one (count - pieces + 1; + 1 without neighbours), a removed one leaves
// them behind (count - 1 + pieces).  Each flood stops as soon as it holds every neighbour not yet accounted for -- around
// one cell that is after two or three rounds unless the cell really was a bridge.
template <int PW>
__device__ inline int m3_regions_update(const M3Ctx &c, PM<PW> A, PM<PW> notx0, PM<PW> notxl, int eq, int ez, bool added,
                                        int regions_old) {
  PM<PW> e = pm_zero<PW>();
  if (c.lane == ez) pm_set(e, eq);
  A = c.lane < c.Z ? A : pm_zero<PW>();
  PM<PW> remaining = m3_grow(c, e, notx0, notxl) & A;
  int pieces = 0;
  while (true) {
    const uint64_t b = M3_BALLOT(pm_any(remaining));
    if (b == 0) break;
    const int fl = __builtin_ctzll(b);
    PM<PW> f = c.lane == fl ? pm_lowest(remaining) : pm_zero<PW>();
    while (true) {
      f = f | (m3_grow(c, f, notx0, notxl) & A);
      if (M3_BALLOT(pm_any(remaining & ~f)) == 0) break;  // every neighbour left is in this component
      const PM<PW> nf = m3_grow(c, f, notx0, notxl) & A & ~f;
      if (M3_BALLOT(pm_any(nf)) == 0) break;
      f = f | nf;
    }
    remaining = remaining & ~f;
    pieces++;
  }
  if (pieces == 0) return regions_old + (added ? 1 : -1);
  return added ? regions_old - (pieces - 1) : regions_old + (pieces - 1);
}

// ---------------------------------------------------------------------------------------------- move table
// helper_3D._passable (:214-319) for one (foothold, direction), branch-free on 6-bit windows of the column masks: bit i of a
// window = AIR at height z-2+i (below the floor and above the ceiling read as not-AIR, which is what every rule's explicit
// bounds check amounts to).  cc / cn / cj: AIR bits over z of the foothold's column, the neighbour column and the column
// two steps away (n_in / j_in: inside the map).  The six rules are mutually exclusive.
// Result: path cost (1..3) | jump << 2 | (height change + 1) << 3;  0 = no move in this direction.
__device__ inline uint32_t m3_move(uint32_t cc, uint32_t cn, uint32_t cj, int z, bool n_in, bool j_in) {
  const uint32_t wn = ((cn << 2) >> z) & 0x3Fu, wj = ((cj << 2) >> z) & 0x3Fu, c4 = (cc >> (z + 2)) & 1u;
  // (bitwise on purpose: `&&` / `||` compile to exec-masked branches, these to mask arithmetic)
  const bool walk = (wn & 0x0Eu) == 0x0Cu;                                   // stands at z: !n[z-1], n[z], n[z+1]
  const bool down = (z >= 1) & ((wn & 0x0Fu) == 0x0Eu);                      // stands at z-1: !n[z-2], n[z-1], n[z], n[z+1]
  const bool up = ((wn & 0x1Cu) == 0x18u) & (c4 != 0u);                      // stands at z+1: !n[z], n[z+1], n[z+2], own z+2 free
  const bool gap = (z >= 2) & ((wn & 0x1Fu) == 0x1Fu) & (c4 != 0u) & j_in;   // n[z-2..z+2] all AIR: a gap to jump over
  const bool jflat = gap & ((wj & 0x1Eu) == 0x1Cu);                          // !j[z-1], j[z], j[z+1], j[z+2]
  const bool jup = gap & ((wj & 0x3Cu) == 0x38u);                            // !j[z], j[z+1], j[z+2], j[z+3]
  const bool jdown = gap & ((wj & 0x0Fu) == 0x0Eu);                          // !j[z-2], j[z-1], j[z], j[z+1]
  const bool jump = jflat | jup | jdown;
  const bool ok = n_in & (walk | down | up | jump);
  const uint32_t w = walk ? 1u : ((jup | jdown) ? 3u : 2u);
  const uint32_t dzp = 1u + ((up | jup) ? 1u : 0u) - ((down | jdown) ? 1u : 0u);
  return ok ? (w | (jump ? 4u : 0u) | (dzp << 3)) : 0u;
}
__device__ inline int m3_dx(int d) { return d == 0 ? 1 : (d == 2 ? -1 : 0); }  // helper_3D.py:220 direction order
__device__ inline int m3_dy(int d) { return d == 1 ? 1 : (d == 3 ? -1 : 0); }

// move-table entry of (x, y, z, d) from the column masks: code of m3_move | (target cell - cell) << 5, as int16
__device__ inline int m3_move_at(const M3Ctx &c, int x, int y, int z, int d) {
  const int dx = m3_dx(d), dy = m3_dy(d);
  const int nx = x + dx, ny = y + dy, jx = nx + dx, jy = ny + dy;
  const bool n_in = ((unsigned)nx < (unsigned)c.X) & ((unsigned)ny < (unsigned)c.Y);
  const bool j_in = ((unsigned)jx < (unsigned)c.X) & ((unsigned)jy < (unsigned)c.Y);
  const int q = y * c.X + x, dq = dy * c.X + dx;
  const uint32_t cc = c.col[q], cn = c.col[n_in ? q + dq : 0], cj = c.col[j_in ? q + 2 * dq : 0];
  const uint32_t m = m3_move(cc, cn, cj, z, n_in, j_in);
  const int delta = dq + ((m & 4u) ? dq : 0) + ((int)((m >> 3) & 3u) - 1) * c.YX;
  return m ? (int)(int16_t)((delta << 5) | (int)m) : 0;
}

// per-(y,x) column masks from the planes: lane q of pass k collects bit q of every plane
template <int PW>
__device__ inline void m3_build_cols(const M3Ctx &c, PM<PW> air) {
#pragma unroll
  for (int k = 0; k < PW; k++) {
    uint32_t m = 0;
    for (int z = 0; z < c.Z; z++) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)air.w[k], z);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(air.w[k] >> 32), z);
      const uint64_t a = (uint64_t)lo | ((uint64_t)hi << 32);
      m |= (uint32_t)((a >> c.lane) & 1ull) << z;
    }
    const int q = 64 * k + c.lane;
    if (q < c.YX) c.col[q] = (uint16_t)m;
  }
}

// the whole table from the column masks (reset, injected maps)
__device__ inline void m3_build_moves(const M3Ctx &c) {
  for (int i = c.lane; i < c.n_cells * 4; i += 64) {
    const int cell = i >> 2, d = i & 3;
    const int z = cell / c.YX, q = cell - z * c.YX, y = q / c.X, x = q - y * c.X;
    c.mv[i] = (int16_t)m3_move_at(c, x, y, z, d);
  }
}

// After the edit of cell (ex, ey, ez) (c.col already updated): re-evaluate the <= 48 (cell, direction) pairs whose rules
// read that cell, and drop the cached slots that accepted a cell whose move changed.
// Returns the mask of the slots it dropped; chg / chg_cell: this lane changed a byte of that cell's table row.
__device__ inline uint32_t m3_update_moves(const M3Ctx &c, int ex, int ey, int ez, bool &chg, int &chg_cell) {
  // lane -> (direction, source cell): 0..3 the cell two below (own column at z+2); 4..23 the four neighbours at heights
  // ez-2..ez+2 (window of the neighbour column); 24..47 the cells two steps away at heights ez-3..ez+2 (landing window)
  const int L = c.lane;
  const int d = L & 3;
  const int grp = L < 4 ? 0 : (L < 24 ? 1 : 2);
  const int k = grp == 0 ? 0 : (grp == 1 ? (L - 4) >> 2 : (L - 24) >> 2);
  const int sx = ex - grp * m3_dx(d), sy = ey - grp * m3_dy(d);
  const int sz = grp == 0 ? ez - 2 : (grp == 1 ? ez - 2 + k : ez - 3 + k);
  const bool in = (L < 48) & ((unsigned)sx < (unsigned)c.X) & ((unsigned)sy < (unsigned)c.Y) & ((unsigned)sz < (unsigned)c.Z);
  const int cell = in ? m3_cell(c, sx, sy, sz) : 0;
  const int nw_ = m3_move_at(c, in ? sx : 0, in ? sy : 0, in ? sz : 0, d);
  const int old = c.mv[cell * 4 + d];
  chg = in & (nw_ != old);
  chg_cell = cell;
  if (chg) c.mv[cell * 4 + d] = (int16_t)nw_;
  uint32_t dropped = 0;
#ifdef PCGRL_PHASE_TIMING
  if (c.knob == 3) return 0u;  // development: time without the slot-drop test (results are wrong)
#endif
  if (M3_BALLOT(chg) != 0) {
    // (all reads first: one LDS round trip for the whole loop)
    constexpr int MAXS = 14;
    uint32_t hv[MAXS], rw[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
      hv[s] = s < c.L.n_slots ? *(const uint32_t *)c.hdr(s) : 0u;  // start | valid << 16
      rw[s] = s < c.L.n_slots ? c.racc(s)[cell >> 5] : 0u;
    }
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
      if (s >= c.L.n_slots) break;
      const bool hit = chg & ((hv[s] >> 16) != 0u) & (((rw[s] >> (cell & 31)) & 1u) != 0u);
      if (M3_BALLOT(hit) != 0) {
        if (c.lane == 0) c.hdr(s)->valid = 0;
        dropped |= 1u << s;
      }
    }
  }
  return dropped;
}

// ---------------------------------------------------------------------------------------------- path search
// One search of helper_3D.run_dijkstra from `root`.  Uniform over the wave; returns the number of accepted cells
// (W.order).  `overflow`: more than RING live queue entries.
//
// The reference pops one queue entry at a time.  Here up to 16 consecutive entries are taken per trip, lane 4*i + d
// working on direction d of entry i, which is exact because:
//   * whether entry i is accepted (:437-440) depends on earlier entries only through `best` of ITS OWN cell.  Every
//     popped entry registers (trip, slot) for its cell with an LDS min issued BEFORE the reads of the trip -- later trips
//     have smaller stamps, so nothing is ever reset -- and one round trip returns `best`, the first slot of the cell and
//     the move byte together; the trip is cut before an accept candidate that is not the first entry of its cell in the
//     trip (rare), which then runs first in the next trip;
//   * the head-room test of :443-445 never fails here: every rule of _passable checks the head-room of its target, and
//     the start cells have it;
//   * a successor is queued unless it is known to be a no-op when popped: the move straight back to the cell the entry
//     came from (its `best` is at most the parent's length, which is below the entry's), and -- while the queue is long;
//     the check is one more dependent LDS round trip -- any target whose `best`, as read in this trip (`best` only
//     decreases), is not longer.  Entries that turn out to be no-ops later are rejected when popped, like in the reference;
//   * first-visit order and queue order are kept with prefix counts over the lanes (entry-major, direction-minor).
// SHORT QUEUES (corridors: where a search spends its trips) are popped one entry at a time, like the reference does: the
// entry sits in scalar registers, one LDS round trip returns `best` and the four move bytes of its cell, the accept test is
// scalar, lanes 0..3 evaluate the directions, and a single successor of a lone entry is handed to the next trip by a lane
// read instead of through the queue.  (A wave issues a DEPENDENT instruction every 8 cycles whatever its width and pipe --
// tools/ubench/latency.hip -- so a trip costs its instruction count: ~60 here, ~100 in the 16-entry trip.)
// `best` entries carry the search's epoch, so nothing is cleared between searches.
// `cancel` (helper wave, see SPECULATION): the search gives up as soon as *cancel is set; its result is then meaningless.
template <int SC, bool CANCEL = false>
__device__ inline int m3_search(M3Work<SC> &W, const M3Ctx &c, int root, uint32_t &epoch, uint32_t &trip, bool &overflow,
                                const int *cancel PHASE_ARG) {
  constexpr int RING = M3C<SC>::RING, RM = RING - 1;
  const int16_t *mv = c.mv;
  uint32_t ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)epoch) + 1u;
  if (ep > 255u) {  // wrapped: clear the table once
    for (int i = c.lane; i < c.n_cells; i += 64) W.best[i].x = 0;
    ep = 1;
  }
  ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
  epoch = ep;
  uint32_t tr = (uint32_t)__builtin_amdgcn_readfirstlane((int)trip);
  int head = 0, tail = 0, n_order = 0;
  const int slot_i = c.lane >> 2, d = c.lane & 3;
  const uint32_t ep24 = ep << 24;
#ifdef PCGRL_PHASE_TIMING
  int dbg_trips = 0, dbg_pushed = 1;
  const uint64_t dbg_t0 = __builtin_readcyclecounter();
#ifdef PCGRL_M3_TRIPS
  uint64_t dbg_tt = dbg_t0;
  int dbg_kind = -1;
#define M3_TRIP_KIND(kind)                                                      \
  do {                                                                          \
    const uint64_t t_ = __builtin_readcyclecounter();                           \
    if (dbg_kind >= 0) _ph[2 + dbg_kind] += (uint32_t)(t_ - dbg_tt);            \
    dbg_tt = t_;                                                                \
    dbg_kind = (kind);                                                          \
    _ph[dbg_kind] += 1;                                                         \
  } while (0)
#else
#define M3_TRIP_KIND(kind) dbg_trips++
#endif
#else
#define M3_TRIP_KIND(kind) \
  do {                     \
  } while (0)
#endif
  // the entry being processed one at a time, in scalar registers; the root has no parent
  // (loop-carried in vector registers, read back with v_readfirstlane at the top of a trip: that keeps the trip's
  // arithmetic and, above all, its branches on the scalar unit)
  uint32_t exv = (uint32_t)root | (0x1FFFu << 12), eyv = 1u;
  bool in_hand = true;
  constexpr int WIDE_MIN = 2;  // queue lengths from which the 16-entry trip pays (measured: 5 is slower, a scalar trip costs ~700 cycles)
  for (;;) {
    if (CANCEL) {
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0) {
        overflow = true;  // (reported to the caller as "no result")
        break;
      }
    }
    if (!in_hand) {
      head = __builtin_amdgcn_readfirstlane(head);
      tail = __builtin_amdgcn_readfirstlane(tail);
      if (head >= tail) break;
      if (tail - head < WIDE_MIN) {  // a short queue is popped like the reference pops it
        const uint2 e = W.ent[head & RM];
        exv = e.x;
        eyv = e.y;
        head++;
        in_hand = true;
      }
    }
    if (in_hand) {
      // ---- one entry: everything about it is scalar
      M3_TRIP_KIND(0);
      const uint32_t ex = (uint32_t)__builtin_amdgcn_readfirstlane((int)exv), ey = (uint32_t)__builtin_amdgcn_readfirstlane((int)eyv);
      const int cell = (int)(ex & 0xFFFu), parent = (int)((ex >> 12) & 0x1FFFu);
      const uint32_t len = ey;
      const uint32_t bxv = W.best[cell].x;
      const int mvl = mv[cell * 4 + d];  // (one LDS round trip; lanes 0..3 keep theirs)
      const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)bxv);
      const bool seen = (bx >> 24) == ep;
      in_hand = false;
      if (seen && (bx & 0xFFFFFFu) <= len) continue;  // :437-440 not shorter: dropped
      if (c.lane == 0) {
        if (!seen) W.order[n_order] = (uint16_t)cell;
        W.best[cell].x = ep24 | len;
        W.info[cell] = ex >> 12;
      }
      n_order += seen ? 0 : 1;
      const int m = c.lane < 4 ? mvl : 0;
      const int tcell = cell + (m >> 5);
      const bool ok = (m != 0) & (tcell != parent);
      const uint32_t okb = (uint32_t)M3_BALLOT(ok);
      const uint32_t cx = (uint32_t)tcell | ((uint32_t)cell << 12) | (((uint32_t)m & 31u) << 25) | ((uint32_t)d << 30);
      const uint32_t cy = len + ((uint32_t)m & 3u);
      if (okb == 0u) continue;
      head = __builtin_amdgcn_readfirstlane(head);
      tail = __builtin_amdgcn_readfirstlane(tail);
      if ((okb & (okb - 1u)) == 0u && head >= tail) {  // one successor and nothing waiting: it is the next entry
        const int l = __builtin_ctz(okb);
        exv = (uint32_t)__builtin_amdgcn_readlane((int)cx, l);
        eyv = (uint32_t)__builtin_amdgcn_readlane((int)cy, l);
        in_hand = true;
#ifdef PCGRL_PHASE_TIMING
        dbg_pushed++;
#endif
        continue;
      }
      const int npush = __popc(okb);
      if (tail + npush - head > RING) {
        overflow = true;
        break;
      }
      if (ok) W.ent[(tail + m3_below((uint64_t)okb)) & RM] = make_uint2(cx, cy);
      tail += npush;
#ifdef PCGRL_PHASE_TIMING
      dbg_pushed += npush;
#endif
      continue;
    }
    // ---- general trip: up to 16 entries
    M3_TRIP_KIND(1);
    tr++;
    const int n_q = tail - head;
    if (n_q > RING - 64) {  // (a trip pushes at most 64 entries)
      overflow = true;
      break;
    }
    // The trip's predicates are lane MASKS in scalar registers (one v_cmp each, combined with scalar ANDs, turned back
    // into predication with inverse_ballot): a ballot of a combined boolean costs a v_cndmask + v_cmp pair instead.
    const int nb = min(16, n_q);
    constexpr uint64_t D0 = 0x1111111111111111ull;  // the direction-0 lane of every entry
    const uint64_t live_m = M3_BALLOT(slot_i < nb);
    const int id = head + min(slot_i, nb - 1);  // (lanes beyond the live entries re-read the last one: harmless)
    const uint2 e = W.ent[id & RM];
    const int cell = (int)(e.x & 0xFFFu), parent = (int)((e.x >> 12) & 0x1FFFu);
    const uint32_t len = e.y;
    const uint32_t stamp = ((0x0FFFFFFFu - tr) << 4) | (uint32_t)slot_i;
    if (__builtin_amdgcn_inverse_ballot_w64(live_m & D0)) atomicMin(&W.best[cell].y, stamp);
    const uint2 b = W.best[cell];
    const int m = mv[cell * 4 + d];
    // :437-440 (an entry that is not shorter is dropped): best ^ epoch is the accepted length if the cell was seen in this
    // search and at least 2^24 otherwise
    const uint32_t keyv = b.x ^ ep24;
    const uint64_t accept_m = M3_BALLOT(keyv > len) & live_m;
    // cut the trip before an accept candidate that is not the first popped entry of its cell in this trip
    const uint64_t dup_m = M3_BALLOT(b.y != stamp) & accept_m & D0;
    const int nproc = dup_m ? (__builtin_ctzll(dup_m) >> 2) : nb;  // >= 1: slot 0 is always the first of its cell
    const uint64_t doit_m = accept_m & M3_BALLOT(slot_i < nproc);
    const uint64_t acc0_m = doit_m & D0;
    const uint64_t first_m = acc0_m & M3_BALLOT(keyv >= (1u << 24));
    if (__builtin_amdgcn_inverse_ballot_w64(first_m)) W.order[n_order + m3_below(first_m)] = (uint16_t)cell;
    n_order += __popcll(first_m);
    if (__builtin_amdgcn_inverse_ballot_w64(acc0_m)) {
      W.best[cell].x = ep24 | len;
      W.info[cell] = e.x >> 12;
    }
    // successor in direction d: one entry of the move table
    const int tcell = cell + (m >> 5);
    const uint32_t tlen = len + ((uint32_t)m & 3u);
    uint64_t ok_m = doit_m & M3_BALLOT(m != 0) & M3_BALLOT(tcell != parent);
    if (n_q > 32) {  // never queue what is known to be a no-op when popped
      const uint32_t bt = W.best[tcell].x ^ ep24;  // (tcell is a cell of the map also where there is no move: delta 0)
      ok_m &= M3_BALLOT(bt > tlen);
    }
    if (__builtin_amdgcn_inverse_ballot_w64(ok_m))
      W.ent[(tail + m3_below(ok_m)) & RM] =
          make_uint2((uint32_t)tcell | ((uint32_t)cell << 12) | (((uint32_t)m & 31u) << 25) | ((uint32_t)d << 30), tlen);
    const int npush = __popcll(ok_m);
    tail += npush;
    head += nproc;
#ifdef PCGRL_PHASE_TIMING
    dbg_pushed += npush;
#endif
  }
  trip = tr;
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_TRIPS)
  if (dbg_kind >= 0) _ph[2 + dbg_kind] += (uint32_t)(__builtin_readcyclecounter() - dbg_tt);
  (void)_t_prev;
  (void)dbg_trips;
  (void)dbg_pushed;
#elif defined(PCGRL_PHASE_TIMING) && !defined(PCGRL_M3_PHASES) && !defined(PCGRL_M3_SPEC) && !defined(PCGRL_M3_TAIL) && \
    !defined(PCGRL_M3_TAILSPLIT) && !defined(PCGRL_M3_HEADSPLIT)
  (void)_t_prev;
  _ph[0] += (uint32_t)dbg_trips;
  _ph[1] += (uint32_t)dbg_pushed;
  _ph[2] += 1;
  _ph[3] += (uint32_t)(__builtin_readcyclecounter() - dbg_t0);
#elif defined(PCGRL_PHASE_TIMING)
  (void)_t_prev;
  (void)_ph;
  (void)dbg_trips;
  (void)dbg_pushed;
  (void)dbg_t0;
#endif
#undef M3_TRIP_KIND
  return n_order;
}

// After a search: the first maximum of len(path) in first-insertion order (helper_3D.py:538-541) -> far; the accepted
// cells -> W.racc; returns the coordinate values of those cells (bit v set if some accepted cell has x, y or z == v: the
// marks of :531).
template <int SC>
__device__ inline uint32_t m3_collect(M3Work<SC> &W, const M3Ctx &c, int n_order, int &far) {
  uint32_t mkl = 0, key = 0;  // len << 16 | (0xFFFF - k): max key = longest, earliest
  for (int k = c.lane; k < n_order; k += 64) {
    const int ci = W.order[k];
    const uint32_t len = W.best[ci].x & 0xFFFFu;
    const uint32_t kk = (len << 16) | (uint32_t)(0xFFFF - k);
    key = kk > key ? kk : key;
    const int z = ci / c.YX, q = ci - z * c.YX, y = q / c.X, x = q - y * c.X;
    mkl |= (1u << x) | (1u << y) | (1u << z);
    atomicOr(&W.racc[ci >> 5], 1u << (ci & 31));
  }
  key = wave_max(key);
  far = W.order[0xFFFF - (int)(key & 0xFFFF)];
  return wave_or(mkl);
}

// The tiles of paths[(mx,my,mz)] of the search just run in W, as a bit mask into slot s; returns n_jump of that path.
// The accepted entries form a tree (an accepted entry's parent is the accepted entry of the parent cell: a strictly
// shorter path to the parent would have produced a strictly shorter, hence accepted, entry for the child).  The chain of
// parent cells is walked first -- one dependent LDS read per hop and nothing else, hop h parked in lane h -- then the
// lanes mark their hop's cell and the intermediate tiles of its move in parallel: an entry knows its move code and
// direction (helper_3D.py:214-319; +-YX = one plane up / down).  n_j of an entry = the jumps along its chain (:283-319).
template <int SC>
__device__ inline int m3_path_tiles(M3Work<SC> &W, const M3Ctx &c, int s, int far2) {
  uint32_t *sp = c.spath(s);
  for (int i = c.lane; i < c.L.nw; i += 64) sp[i] = 0;
  int cell = far2, n_jump = 0;
  bool more = true;
  while (more) {
    int mine = -1;
    int h = 0;
    for (; h < 64; h++) {  // up to 64 hops per batch
      mine = c.lane == h ? cell : mine;
      const int par = __builtin_amdgcn_readfirstlane((int)W.info[cell]) & 0x1FFF;
      if (par == 0x1FFF) break;
      cell = par;
    }
    more = h == 64;  // (the batch ended on a cell that is not the root: it continues there)
    const uint32_t inf = mine >= 0 ? W.info[mine] : 0u;
    const uint32_t m = (inf >> 13) & 31u;  // move code of the hop INTO this cell (0: the root)
    n_jump += __popcll(M3_BALLOT((m & 4u) != 0u));
    if (mine >= 0) {
      const int d = (int)((inf >> 18) & 3u);
      const int dq = m3_dy(d) * c.X + m3_dx(d), dz = (int)((m >> 3) & 3u) - 1;
      auto mark = [&](int t) { atomicOr(&sp[t >> 5], 1u << (t & 31)); };
      mark(mine);
      if (m != 0u) {
        if (m & 4u) {  // jumps: the jumped-over column at the landing's height, and at the take-off's height if they differ
          mark(mine - dq);
          if (dz != 0) mark(mine - dq - dz * c.YX);
        } else {
          if (dz < 0) mark(mine + c.YX);  // step down: the target column at the parent's height
          if (dz > 0) mark(mine - dq);    // step up: above the parent
        }
      }
    }
  }
  return n_jump;
}

// SPECULATION.  The second search of a pair starts at the farthest cell of the first, which is only known when the first
// has ended -- but after a one-cell edit it is usually the cell it was the last time this plane was searched.  The step
// kernel therefore carries a helper wavefront with a search workspace of its own: while the simulate wave runs the first
// search, the helper runs the second one from the REMEMBERED farthest cell (kept in the slot header), including its
// farthest-cell arg-max and path tiles.  If the first search ends at that cell the pair costs the longer of the two
// searches instead of their sum; otherwise the helper is cancelled and the simulate wave runs the second search itself.
// Results are those of the sequential pair either way.  Hand-over through an LDS mailbox.
struct M3Mail {
  int32_t seq;     // job number, bumped by the simulate wave once root / slot are in place
  int32_t done;    // = seq when the helper has finished (or dropped) that job
  int32_t cancel;  // the running job's result is not needed
  int32_t exit;    // the simulate wave is done with the launch
  int32_t root, slot;
  int32_t ok, far2, max_dist, n_jump;  // results
  // the region count, also on the helper wave (it does not depend on the searches): job number / completion, the edited
  // cell (plane bit, plane), kind (0 new AIR cell, 1 removed, 2 count from scratch), count before the edit -> count
  int32_t rseq, rdone, r_eq, r_ez, r_kind, r_old, r_out;
  int32_t obs_read;  // observe waves that hold their copy of the old state (the simulate wave writes back after all of them)
};
__device__ inline int m3_ld(const int32_t *x) { return __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void m3_st(int32_t *x, int v) { __hip_atomic_store(x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// second search of a pair from `root` in workspace W: accepted cells -> W.racc (OR-ed in), path tiles -> slot s.
template <int SC, bool CANCEL>
__device__ inline bool m3_second_search(M3Work<SC> &W, const M3Ctx &c, int s, int root, uint32_t &epoch, uint32_t &trip, int &far2,
                                        int &max_dist, int &n_jump, const int *cancel PHASE_ARG) {
  bool overflow = false;
  const int n_order = m3_search<SC, CANCEL>(W, c, root, epoch, trip, overflow, cancel PHASE_PASS);
  if (overflow) return false;
  (void)m3_collect(W, c, n_order, far2);
  max_dist = (int)(__builtin_amdgcn_readfirstlane((int)W.best[far2].x) & 0xFFFF);
  n_jump = m3_path_tiles(W, c, s, far2);
  return true;
}

// body of the helper wave: serve the simulate wave's jobs until it leaves
// SEARCH: the wave also runs speculative second searches in its own workspace W (size class 0); otherwise it only counts
// regions (size class 1: there is no LDS for a second workspace) and W is not touched.
template <int SC, bool SEARCH>
__device__ inline void m3_helper(const Params &p, M3Work<SC> &W, const M3Ctx &c, M3Mail &m PHASE_ARG) {
  constexpr int PW = M3C<SC>::PW;
  uint32_t epoch = 0, trip = 0;
  if constexpr (SEARCH) {
    for (int i = c.lane; i < c.n_cells; i += 64) W.best[i] = make_uint2(0u, 0xFFFFFFFFu);
  }
  PM<PW> notx0, notxl;
  m3_edge_masks<PW>(p, notx0, notxl);
  int seen = 0, rseen = 0;
  while (true) {
    int sq, rq;
    while (true) {
      rq = __builtin_amdgcn_readfirstlane(m3_ld(&m.rseq));
      sq = __builtin_amdgcn_readfirstlane(m3_ld(&m.seq));
      if (rq != rseen || sq != seen) break;
      if (__builtin_amdgcn_readfirstlane(m3_ld(&m.exit)) != 0) return;
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // (raising this wave's issue priority while it runs a job was measured: 20.3 -> 20.6 us per launch, it takes slots from
    // the simulate waves of the other workgroups on its SIMD)
    if (rq != rseen) {  // the region count of the edited map (the tile bits in LDS are already those of the new map)
      rseen = rq;
      const int eq = __builtin_amdgcn_readfirstlane(m3_ld(&m.r_eq)), ez = __builtin_amdgcn_readfirstlane(m3_ld(&m.r_ez));
      const int kind = __builtin_amdgcn_readfirstlane(m3_ld(&m.r_kind)), old = __builtin_amdgcn_readfirstlane(m3_ld(&m.r_old));
      PM<PW> air = c.lane < c.Z ? m3_plane_air<PW>(c.dirt, c, c.lane) : pm_zero<PW>();
      int out;
      if (kind == 2) {
        out = m3_regions<PW>(c, air, notx0, notxl);
      } else {
        if (c.lane == ez) {
          PM<PW> e = pm_zero<PW>();
          pm_set(e, eq);
          air = air & ~e;
        }
        out = m3_regions_update<PW>(c, air, notx0, notxl, eq, ez, kind == 0, old);
      }
      if (c.lane == 0) {
        m.r_out = out;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        m3_st(&m.rdone, rq);
      }
      continue;
    }
    seen = sq;
    if constexpr (!SEARCH) continue;  // (never posted without a workspace)
    const int root = __builtin_amdgcn_readfirstlane(m3_ld(&m.root)), s = __builtin_amdgcn_readfirstlane(m3_ld(&m.slot));
    for (int i = c.lane; i < c.L.nw; i += 64) W.racc[i] = 0;
    int far2 = 0, max_dist = 0, n_jump = 0;
    const bool ok = m3_second_search<SC, true>(W, c, s, root, epoch, trip, far2, max_dist, n_jump, &m.cancel PHASE_PASS);
    if (c.lane == 0) {
      m.ok = ok ? 1 : 0;
      m.far2 = far2;
      m.max_dist = max_dist;
      m.n_jump = n_jump;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      m3_st(&m.done, sq);
    }
  }
}

// The pair of searches of one start candidate (helper_3D.py:527-553) -> slot s of the env (result + accepted cells).
// W2 / mail: the helper wave's workspace and mailbox (null: no helper).
template <int SC>
__device__ inline void m3_fill_slot(M3Work<SC> &W, const M3Ctx &c, int s, int start_bit, int sz, uint32_t &epoch, uint32_t &trip,
                                    bool &overflow, M3Work<SC> *W2, M3Mail *mail PHASE_ARG) {
  // the farthest cell of this plane's first search the last time it ran (+1; 0: none)
  const uint32_t far_word = mail != nullptr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)c.hdr(s)->far1) : 0u;
  int guess = (int)(far_word & 0xFFFFu) - 1;
  if (guess >= c.n_cells) guess = -1;
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_SPEC)
  const int guess2 = (int)(far_word >> 16) - 1;  // (development: the far end of the last second search)
  const int old_start = (int)__builtin_amdgcn_readfirstlane((int)c.hdr(s)->start);
#endif
  int seq = 0;
  if (guess >= 0) {
    if (c.lane == 0) {
      seq = m3_ld(&mail->seq) + 1;
      mail->root = guess;
      mail->slot = s;
      m3_st(&mail->cancel, 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      m3_st(&mail->seq, seq);
    }
    seq = __builtin_amdgcn_readfirstlane(seq);
  }
  if (c.lane == 0) c.hdr(s)->valid = 0;
  for (int i = c.lane; i < c.L.nw; i += 64) W.racc[i] = 0;
  const int root = sz * c.YX + start_bit;
  int n_order = m3_search<SC>(W, c, root, epoch, trip, overflow, nullptr PHASE_PASS);
  int far1 = 0, far2 = 0, n_jump = 0, max_dist = 0;
  uint32_t mk = 0;
  if (!overflow) mk = m3_collect(W, c, n_order, far1);
  bool have2 = false;
  if (guess >= 0) {  // the helper's job: used, or called off
    const bool hit = !overflow && far1 == guess;
    if (!hit && c.lane == 0) m3_st(&mail->cancel, 1);
    while (__builtin_amdgcn_readfirstlane(m3_ld(&mail->done)) != seq) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (hit && __builtin_amdgcn_readfirstlane(m3_ld(&mail->ok)) != 0) {
      far2 = __builtin_amdgcn_readfirstlane(m3_ld(&mail->far2));
      max_dist = __builtin_amdgcn_readfirstlane(m3_ld(&mail->max_dist));
      n_jump = __builtin_amdgcn_readfirstlane(m3_ld(&mail->n_jump));
      for (int i = c.lane; i < c.L.nw; i += 64) W.racc[i] |= W2->racc[i];
      have2 = true;
    }
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_SPEC)
    _ph[1] += 1;
    _ph[2] += have2 ? 1 : 0;
    _ph[3] += (!overflow && far1 == guess2) ? 1 : 0;
    _ph[4] += (old_start != start_bit) ? 1 : 0;
#endif
  }
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_SPEC)
  _ph[0] += 1;
#endif
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_TAIL)
  _ph[1] += 1;
  _ph[5] += have2 ? 1u : 0u;
#endif
  if (overflow) return;
  if (!have2) {
    if (!m3_second_search<SC, false>(W, c, s, far1, epoch, trip, far2, max_dist, n_jump, nullptr PHASE_PASS)) {
      overflow = true;
      return;
    }
  }
  uint32_t *ra = c.racc(s);
  for (int i = c.lane; i < c.L.nw; i += 64) ra[i] = W.racc[i];
  if (c.lane == 0) {
    M3SlotHdr h;
    h.start = (uint16_t)start_bit;
    h.valid = 1;
    h.max_dist = (uint16_t)max_dist;
    h.n_jump = (uint16_t)n_jump;
    h.mk = mk & ((1u << c.Z) - 1u);
    h.far1 = (uint32_t)(far1 + 1) | ((uint32_t)(far2 + 1) << 16);
    *c.hdr(s) = h;
  }
}

// helper_3D.calc_longest_path + remove_stacked_path_tiles (path-length, n_jump and the overlay of
// minecraft_3D_maze_prob.get_stats; the caller supplies the region count).
// air: this lane's plane (lanes < Z).  Results uniform over the wave.  c.over receives the new overlay mask.
// Slots that are still valid (see SLOT CACHE) are reused; the caller invalidates them for fresh maps.
template <int SC>
__device__ inline void m3_paths(M3Env<SC> &E, M3Work<SC> &W, const M3Ctx &c, PM<M3C<SC>::PW> air, int32_t *st, uint32_t &epoch,
                                uint32_t &trip, uint32_t &filled, bool &overflow, M3Work<SC> *W2, M3Mail *mail PHASE_ARG) {
  constexpr int PW = M3C<SC>::PW;
  // start candidates per plane: AIR with head-room, standing on something, z >= 1 (:520-526)
  const PM<PW> above = pm_down(air), below = pm_up(air);
  const PM<PW> cand = (c.lane >= 1 && c.lane + 1 < c.Z) ? (air & above & ~below) : pm_zero<PW>();
  uint32_t marked = 0;  // z-planes of final_visited_map that are fully set (the fancy-index bug, :531)
  int final_value = 0, n_jump = 0, best_slot = -1;
#if defined(PCGRL_PHASE_TIMING) && defined(PCGRL_M3_TAIL)
  {  // start planes whose cached result is missing right now (most of them lie in planes the walk never looks at)
    bool need = false;
    if (c.lane >= 1 && c.lane + 1 < c.Z && pm_any(cand)) {
      const uint32_t h0 = *(const uint32_t *)c.hdr(c.lane - 1);
      need = !((h0 >> 16) != 0u && (int)(h0 & 0xFFFFu) == pm_ctz(cand));
    }
    _ph[0] += (uint32_t)__popcll(M3_BALLOT(need));
  }
#endif
  while (true) {
    const bool mine = c.lane < c.Z && pm_any(cand) && !((marked >> c.lane) & 1u);
    const uint64_t b = M3_BALLOT(mine);
    if (b == 0) break;
    const int sz = __builtin_ctzll(b);
    const int bit = __builtin_amdgcn_readlane(pm_ctz(cand), sz);
    const int s = sz - 1;
    {
      const uint32_t h0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t *)c.hdr(s));  // start | valid << 16
      if (!((h0 >> 16) != 0u && (int)(h0 & 0xFFFFu) == bit)) {
        m3_fill_slot(W, c, s, bit, sz, epoch, trip, overflow, W2, mail PHASE_PASS);
        filled |= 1u << s;
        if (overflow) break;
      }
    }
    const uint32_t h1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const uint32_t *)c.hdr(s))[1]);  // max_dist | n_jump << 16
    marked |= (uint32_t)__builtin_amdgcn_readfirstlane((int)c.hdr(s)->mk);
    n_jump = (int)(h1 >> 16);  // :553 overwritten by every processed component
    if ((int)(h1 & 0xFFFFu) > final_value) {
      final_value = (int)(h1 & 0xFFFFu);
      best_slot = s;
    }
  }
  M3_MARK(3, 6);  // path searches (incl. the search loops counted in [3])
  if (overflow) return;  // reported by the caller; the overlay and the statistics stay those of the last finished update
  // remove_stacked_path_tiles (:657-675) then the transposed overlay of process_observation (:84-93):
  // path tile (x,y,z) is drawn at array index [x][y][z]
  for (int i = c.lane; i < c.L.nw; i += 64) c.over[i] = 0;
  if (best_slot >= 0) {
    const uint32_t *pathm = c.spath(best_slot);
    bool by_rows = false;
    if constexpr (SC == 0) by_rows = c.X <= 8 && c.Z * c.Y <= 64;  // (compile-time for the 7^3 kernels)
    if (by_rows) {
      // Small maps (7^3: 49 rows of 7 cells): lane r = z * Y + y takes ROW y of plane z -- X <= 8 bits of the packed path
      // mask, and the same row one plane lower for "no path tile directly below" -- and draws it with X predicated LDS ORs: no
      // loop over tiles, no ballots (round 6; rounds 3-5 gave every plane a lane and looped over its tiles, ~1 900 cycles of the
      // critical wave's tail at 7^3).
      const int r = c.lane, z = r / c.Y, y = r - z * c.Y;
      if (r < c.Z * c.Y) {
        auto row_at = [&](int o) -> uint32_t {  // X bits from bit offset o of the mask
          const int w0 = o >> 5, w1 = w0 + 1 < c.L.nw ? w0 + 1 : w0;
          const uint64_t v = (uint64_t)pathm[w0] | ((uint64_t)pathm[w1] << 32);
          return (uint32_t)(v >> (o & 31)) & ((1u << c.X) - 1u);
        };
        const int o = r * c.X;
        uint32_t keep = row_at(o);
        if (z >= 1) keep &= ~row_at(o - c.YX);
        if (z < c.X) {
          for (int x = 0; x < c.X && x < c.Z; x++) {
            const int oi = (x * c.Y + y) * c.X + z;
            if ((keep >> x) & 1u) atomicOr(&c.over[oi >> 5], 1u << (oi & 31));
          }
        }
      }
    } else {
      // Lane z holds plane z of the path as a bit mask, so "no path tile directly below" (remove_stacked_path_tiles) is one
      // shift between lanes; what is left -- a handful of tiles per plane -- is drawn bit by bit, every lane its own plane.
      // (Round 3 walked all Z heights of every column: 60 dependent LDS reads per lane at 15^3, 4 us of a changing step.)
      const PM<PW> P = c.lane < c.Z ? m3_plane_bits<PW>(pathm, c, c.lane) : pm_zero<PW>();
      PM<PW> keep = P & ~pm_up(P);  // (lane z - 1's plane; zero into lane 0)
      const int z = c.lane;
      while (M3_BALLOT(pm_any(keep)) != 0) {
        if (pm_any(keep)) {
          const int q = pm_ctz(keep);
          keep = keep & ~pm_lowest(keep);
          const int y = q / c.X, x = q - y * c.X;
          if (x < c.Z && z < c.X) {
            const int oi = (x * c.Y + y) * c.X + z;
            atomicOr(&c.over[oi >> 5], 1u << (oi & 31));
          }
        }
      }
    }
  }
  st[1] = final_value;
  st[2] = n_jump;
  M3_MARK(4, 5);  // overlay post-processing
}

// observation: (o0, o1, o2, 4) uint8, channel 0 = out of bounds, 1 = AIR, 2 = DIRT, 3 = path overlay.
// Cell-by-cell form (any window).
__device__ inline void m3_encode_obs_cells(const uint32_t *dirt, const uint32_t *over, const M3Ctx &c, const Params &p, int env,
                                           const int *pos, bool show_path, uint8_t *obs_base) {
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

// Row form: the window is a crop of the (conceptually padded) map, so a row of it along x is a shifted copy of one map row.
// Pass 1: per window row (i, j) two o2-bit masks -- bit k = low / high bit of the channel of cell (i, j, k) -- into
// `scratch` (o0 * o1 + 1 entries).  Pass 2: every 16-byte chunk (4 cells, possibly across a row end) picks its bits from
// the masks of two consecutive rows; chunk ch = lane + 64 * t, so a store instruction covers 1 KiB of consecutive bytes.
// WIN: a cubic window with compile-time sizes (14: BASELINE's, 30: the reference's stock map's), 0: run-time sizes.
template <int WIN>
__device__ inline void m3_encode_obs(const uint32_t *dirt, const uint32_t *over, const M3Ctx &c, const Params &p, int env, const int *pos,
                                     bool show_path, uint2 *scratch, int scratch_rows, uint8_t *obs_base = nullptr, int part = 0,
                                     int nparts = 1) {
  if (p.obs == nullptr) return;
  if (obs_base == nullptr) obs_base = p.obs;
  const int o0 = WIN ? WIN : p.cfg.obs_window[0], o1 = WIN ? WIN : p.cfg.obs_window[1], o2 = WIN ? WIN : p.cfg.obs_window[2];
  const int rows = o0 * o1;
  if (!WIN && (rows + 1 > scratch_rows || o2 > 32 || o2 < 2)) {  // (o2 == 1: a chunk of 4 cells spans 4 rows, pass 2 reads 2)
    if (part == 0) m3_encode_obs_cells(dirt, over, c, p, env, pos, show_path, obs_base);
    return;
  }
  const int total = rows * o2, chunks = total >> 2;
  uint4 *dst = (uint4 *)(obs_base + (size_t)env * total * 4);
  const int t0 = pos[0] - o0 / 2, t1 = pos[1] - o1 / 2, t2 = pos[2] - o2 / 2;
  const uint32_t xmask = (1u << c.X) - 1u, omask = o2 >= 32 ? 0xFFFFFFFFu : (1u << o2) - 1u;
  const float inv1 = 1.0f / (float)o1, inv2 = 1.0f / (float)o2;
  // `part` of `nparts`: several observe waves share the observation, each a contiguous range of chunks -- and only the row
  // masks that range reads (its rows and the one after), so the waves need nothing from each other
  const int ch_lo = (int)((long long)chunks * part / nparts), ch_hi = (int)((long long)chunks * (part + 1) / nparts);
  const int r_lo = nparts > 1 ? (ch_lo * 4) / o2 : 0;
  const int r_hi = nparts > 1 ? min(rows, (ch_hi * 4 - 1) / o2 + 1) : rows;
  for (int r = r_lo + c.lane; r <= r_hi; r += 64) {
    const int i = (int)(((float)r + 0.5f) * inv1), j = r - i * o1;  // (exact in fp32 for these sizes)
    const int a = t0 + i, b = t1 + j;
    const bool inb = (r < rows) & ((unsigned)a < (unsigned)c.Z) & ((unsigned)b < (unsigned)c.Y);
    const int f = inb ? (a * c.Y + b) * c.X : 0, w = f >> 5, sh = f & 31;
    uint64_t db = (((uint64_t)dirt[w] | ((uint64_t)dirt[w + 1] << 32)) >> sh) & xmask;
    uint64_t ob = show_path ? (((uint64_t)over[w] | ((uint64_t)over[w + 1] << 32)) >> sh) & xmask : 0ull;
    uint64_t xin = xmask;
    if (t2 >= 0) {  // window cell k shows map column t2 + k
      db >>= t2;
      ob >>= t2;
      xin >>= t2;
    } else {
      db <<= -t2;
      ob <<= -t2;
      xin <<= -t2;
    }
    const uint32_t in = inb ? (uint32_t)xin & omask : 0u;
    scratch[r] = make_uint2(in & (~(uint32_t)db | (uint32_t)ob), in & ((uint32_t)db | (uint32_t)ob));
  }
  // Four chunks per trip: the store is an asm statement with a memory clobber, so the LDS reads of the next chunk do not
  // move above it by themselves.  (Row masks shared by two waves' ranges are written by both: identical values.)
  auto chunk = [&](int ch) -> uint4 {
    const int q0 = ch * 4;
    const int r = (int)(((float)q0 + 0.5f) * inv2), k0 = q0 - r * o2;
    const uint2 ra = scratch[r], rb = scratch[r + 1];
    const uint64_t m0 = ((uint64_t)ra.x | ((uint64_t)rb.x << o2)) >> k0, m1 = ((uint64_t)ra.y | ((uint64_t)rb.y << o2)) >> k0;
    uint32_t w[4];
#pragma unroll
    for (int t = 0; t < 4; t++) w[t] = 1u << (8 * (((uint32_t)(m0 >> t) & 1u) + 2u * ((uint32_t)(m1 >> t) & 1u)));
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  const int stride = 64;
  int ch = ch_lo + c.lane;
  for (; ch + 3 * stride < ch_hi; ch += 4 * stride) {
    const uint4 v0 = chunk(ch), v1 = chunk(ch + stride), v2 = chunk(ch + 2 * stride), v3 = chunk(ch + 3 * stride);
    if (p.obs16 & 2) {  // (launches that write far more than the last-level cache holds: non-temporal, see store_obs16_nt)
      store_obs16_nt(dst + ch, v0);
      store_obs16_nt(dst + ch + stride, v1);
      store_obs16_nt(dst + ch + 2 * stride, v2);
      store_obs16_nt(dst + ch + 3 * stride, v3);
    } else {
      store_obs16(dst + ch, v0);
      store_obs16(dst + ch + stride, v1);
      store_obs16(dst + ch + 2 * stride, v2);
      store_obs16(dst + ch + 3 * stride, v3);
    }
  }
  for (; ch < ch_hi; ch += stride) store_obs16(dst + ch, chunk(ch));
}

// reset from the env's RNG streams (envs/pcgrl_env.py:158-188; probabilities, then the map in (z,y,x) order) into `dirt`.
// rp / rr: the env's streams (in registers); advanced.  cpl = cells per lane (<= 64).
__device__ inline void m3_reset_rng(uint32_t *dirt, const M3Ctx &c, const Params &p, int cpl, Pcg &rp, Pcg &rr) {
  double p0 = rp.next_double(), p1 = rp.next_double();
  double total = 0.0;
  total += p0;
  total += p1;
  double c0 = p0 / total, c1 = c0 + p1 / total;
  c0 /= c1;  // cdf /= cdf[-1]
  for (int i = c.lane; i < c.L.nw; i += 64) dirt[i] = 0;
  Pcg end = rr;
  end.jump(p.jump[64]);
  rr.jump(p.jump[c.lane]);
  uint64_t bits = 0;
  const int first = c.lane * cpl;
  for (int k = 0; k < cpl; k++) {
    int ci = first + k;
    if (ci < c.n_cells) {
      double u = rr.next_double();
      int idx = (c0 <= u ? 1 : 0) + (1.0 <= u ? 1 : 0);  // searchsorted(cdf, u, 'right') with cdf[-1] == 1.0
      if (idx >= 1) bits |= 1ull << k;
    }
  }
  for (int k = 0; k < cpl; k++) {
    int ci = first + k;
    if (ci < c.n_cells && ((bits >> k) & 1ull)) atomicOr(&dirt[ci >> 5], 1u << (ci & 31));
  }
  rr = end;
}

// bytes -> bit string (caller-provided maps)
__device__ inline void m3_load_bytes(uint32_t *dirt, const M3Ctx &c, const uint8_t *src) {
  for (int i = c.lane; i < c.L.nw; i += 64) dirt[i] = 0;
  for (int ci = c.lane; ci < c.n_cells; ci += 64)
    if (src[ci]) atomicOr(&dirt[ci >> 5], 1u << (ci & 31));
}

// M3_ROLLOUT: pcgrl_rollout, p.n_steps steps per launch with the env state in LDS / registers (one wave does both roles)
enum M3Mode { M3_STEP = 0, M3_RESET = 1, M3_OBSERVE = 2, M3_STATS_FOR_GRIDS = 3, M3_GET_STATE = 4, M3_ROLLOUT = 5 };

// narrow_rep.py:89-102 with Q1 (position of the NEXT edit from the pre-increment counter)
__device__ inline void m3_advance_pos(const M3Ctx &c, int *pos, int &n_step) {
  const int idx = n_step % c.n_cells;
  pos[0] = idx / c.YX;
  pos[1] = (idx / c.X) % c.Y;
  pos[2] = idx % c.X;
  n_step++;
}

// global -> LDS copy of 16-byte words [from, to): B loads per lane are issued before the first of them is stored
template <int B>
__device__ inline void m3_copy_batched(uint4 *dst, const uint4 *src, int from, int to, int lane) {
  for (int base = from; base < to; base += 64 * B) {
    uint4 r[B];
#pragma unroll
    for (int b = 0; b < B; b++) {
      const int i = base + lane + 64 * b;
      r[b] = i < to ? src[i] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
      const int i = base + lane + 64 * b;
      if (i < to) dst[i] = r[b];
    }
  }
}

// DIM: a cubic map DIM^3 with its 2 DIM window as compile-time dimensions -- 7: BASELINE's shape, 15: the reference's stock
// map (configs/config.py:153-157); 0: run-time dimensions (the search trip
// and the observation encoder are instruction-bound: constant strides and bounds take instructions away)
// HELP (pcgrl_step): one more wavefront counts the regions while the simulate wave searches and, in size class 0 (HELP_S),
// runs second searches speculatively, see SPECULATION.  Waves of a step workgroup: 0 simulate, 1 .. NOBS observe, then helper.
template <int MODE, int SC, int DIM = 0>
// (size class 0 step kernel: 129 VGPRs would mean 3 waves per SIMD = 4 workgroups per CU where the LDS allows 5; the
// second launch bound asks for 4 waves per SIMD, i.e. <= 128 VGPRs)
__global__ __launch_bounds__(MODE == M3_STEP ? 64 * (2 + m3_observers<SC>()) : 64, (MODE == M3_STEP && SC == 0) ? 4 : 1)
void m3_kernel(Params p, int cpl) {
  constexpr int PW = M3C<SC>::PW;
  constexpr bool HELP = MODE == M3_STEP, HELP_S = HELP && SC == 0;
  if (MODE == M3_STEP) touch_kernarg(p);  // every line of the argument block in one scalar-memory round trip
  __shared__ M3Env<SC> E;
  __shared__ M3Work<SC> W;
  __shared__ M3Mail mail;
  M3Work<SC> *WH = nullptr;  // the helper wave's workspace
  if constexpr (HELP_S) {
    __shared__ M3Work<SC> wh_;
    WH = &wh_;
  }
  __shared__ M3ObsLds<SC> O;
  M3Ctx c;
  c.lane = (int)__lane_id();
  c.Z = DIM ? DIM : p.cfg.dims[0];
  c.Y = DIM ? DIM : p.cfg.dims[1];
  c.X = DIM ? DIM : p.cfg.dims[2];
  if (DIM) cpl = (DIM * DIM * DIM + 63) / 64;
  c.YX = c.Y * c.X;
  c.n_cells = c.Z * c.YX;
#ifdef PCGRL_PHASE_TIMING
  c.knob = p.cfg.solver_power;
#endif
  c.L = m3_layout(c.Z, c.Y, c.X);
  c.dirt = E.rec;
  c.over = E.rec + c.L.o_over;
  c.col = (uint16_t *)(E.rec + c.L.o_col);
  c.slots = E.rec + c.L.o_slots;
  c.mv = (int16_t *)(E.rec + c.L.o_mv);
  const int nw = c.L.nw, n_slots = c.L.n_slots;
  const int env = blockIdx.x;
  constexpr int NS = M3_NS;
  PHASE_DECL();
  TRACE_DECL();
  uint32_t *grec = (uint32_t *)p.planes + (size_t)env * c.L.rec_words;
  EnvState *S = &p.st[env];

  if constexpr (HELP) {
    if (threadIdx.x == 0) {
      mail.seq = 0;
      mail.done = 0;
      mail.rseq = 0;
      mail.rdone = 0;
      mail.cancel = 0;
      mail.exit = 0;
      mail.obs_read = 0;
    }
    __syncthreads();  // (the waves of a workgroup start together: nobody waits here)
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1 + m3_observers<SC>()) {
      // ---------------------------------------------------------------------------------------- helper wave
      m3_helper<SC, HELP_S>(p, HELP_S ? *WH : W, c, mail PHASE_PASS);  // (regions only: W is not touched)
      return;
    }
  }
  if constexpr (MODE == M3_STEP) {
    // ------------------------------------------------------------------------------------------ observe wave
    constexpr int NOBS = m3_observers<SC>();
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (readfirstlane: scalar branches)
    if (wave_id >= 1 && wave_id <= NOBS) {
      if (p.obs == nullptr) return;  // (the simulate wave skips the barrier in that case, too)
      const int part = wave_id - 1;  // this wave's share of the observation's chunks
      uint32_t *obits = O.bits[part];
      // everything this wave needs of the old state is requested at once; an auto-reset replays the env's RNG streams in
      // both waves, so it takes its copy of them, too
      for (int i = c.lane; i < 2 * nw; i += 64) obits[i] = grec[i];
      int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};
      int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
      const int action = p.actions[env];
      const bool upd_only = p.update_only != 0;
      Pcg rp, rr;
      rp.load(p.rng[env].prob);
      rr.load(p.rng[env].rep);
      // The simulate wave overwrites the env's state only after every observe wave holds its copy of the old one: each reports
      // in once its loads have returned (rounds 3-5 closed the launch with a barrier of all waves, which made the simulate
      // wave wait for the helper wave to notice the end of the launch: ~400 cycles of every launch's critical wave).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c.lane == 0) __hip_atomic_fetch_add(&mail.obs_read, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      uint32_t *odirt = obits, *oover = obits + nw;
      iteration += upd_only ? 0 : 1;
      bool change = false;
      if (action >= 0 && action < 2) {
        const int ci = m3_cell(c, pos[2], pos[1], pos[0]);  // pos = (z, y, x)
        change = m3_bit(odirt, ci) != (action != 0);
        if (change && c.lane == 0) odirt[ci >> 5] ^= 1u << (ci & 31);
        m3_advance_pos(c, pos, n_step);
      }
      changes += (change && !upd_only) ? 1 : 0;
      bool done = !upd_only && iteration > p.cfg.max_iterations;
      if (p.cfg.max_changes >= 0) done = done || (!upd_only && changes > p.cfg.max_changes);
      if (done && p.auto_reset != 0) {  // first observation of the new episode: no overlay (PcgrlEnv.reset)
        m3_reset_rng(odirt, c, p, cpl, rp, rr);
        pos[0] = pos[1] = pos[2] = 0;
        m3_encode_obs<2 * DIM>(odirt, oover, c, p, env, pos, false, O.rows, (int)(sizeof(O.rows) / sizeof(uint2)), nullptr, part, NOBS);
      } else {
        m3_encode_obs<2 * DIM>(odirt, oover, c, p, env, pos, true, O.rows, (int)(sizeof(O.rows) / sizeof(uint2)), nullptr, part, NOBS);
      }
      TRACE_PUT(4, _tr0);
      TRACE_PUT(3, TRACE_NOW());
      return;
    }
    __builtin_amdgcn_s_setprio(3);  // the simulate wave's dependent chain issues ahead of the observe wave on its SIMD
  }

  if constexpr (MODE == M3_GET_STATE) {
    if (p.out_grids)
      for (int ci = c.lane; ci < c.n_cells; ci += 64) p.out_grids[(size_t)env * c.n_cells + ci] = (grec[ci >> 5] >> (ci & 31)) & 1u;
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
  if constexpr (MODE == M3_OBSERVE) {
    // reset()/observe(): no path overlay (PcgrlEnv.reset does not call process_observation)
    for (int i = c.lane; i < 2 * nw; i += 64) E.rec[i] = grec[i];
    const int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};
    m3_encode_obs<false>(c.dirt, c.over, c, p, env, pos, false, (uint2 *)W.info, M3C<SC>::CELLS / 2);
    return;
  }

  // search tables of this wave
  uint32_t epoch = 0, trip = 0;
  uint32_t dirty_hdr = 0, dirty_full = 0;  // slots whose header / whose whole record differs from the copy in HBM
  auto init_work = [&]() {
    for (int i = c.lane; i < c.n_cells; i += 64) W.best[i] = make_uint2(0u, 0xFFFFFFFFu);
  };
  PM<PW> notx0, notxl;
  m3_edge_masks<PW>(p, notx0, notxl);
  auto plane_of = [&](const uint32_t *dirt) { return c.lane < c.Z ? m3_plane_air<PW>(dirt, c, c.lane) : pm_zero<PW>(); };
  // statistics of a map the kernel has not seen before: columns, move table, no cached slots
  auto fresh_stats = [&](int32_t *st, bool &ovf) {
    const PM<PW> air = plane_of(c.dirt);
    for (int i = c.lane; i < c.L.o_slots - c.L.o_col; i += 64) E.rec[c.L.o_col + i] = 0;
    m3_build_cols<PW>(c, air);
    m3_build_moves(c);
    if (c.lane < n_slots) *(uint4 *)c.hdr(c.lane) = make_uint4(0u, 0u, 0u, 0u);
    dirty_hdr = (1u << n_slots) - 1u;
    M3_MARK(6, 5);  // column masks + move table
    st[0] = m3_regions<PW>(c, air, notx0, notxl);
    M3_MARK(2, 4);  // regions
    m3_paths<SC>(E, W, c, air, st, epoch, trip, dirty_full, ovf, nullptr, nullptr PHASE_PASS);
  };
  auto store_record = [&]() {  // the whole record
    for (int i = c.lane; i < c.L.rec_words / 4; i += 64) ((uint4 *)grec)[i] = ((const uint4 *)E.rec)[i];
  };

  if constexpr (MODE == M3_STATS_FOR_GRIDS) {
    m3_load_bytes(c.dirt, c, p.init_grids + (size_t)env * c.n_cells);
    init_work();
    int32_t st[NS] = {0, 0, 0};
    bool ovf = false;
    fresh_stats(st, ovf);
    if (ovf && c.lane == 0) atomicOr(p.err, 4);
    if (c.lane == 0)
      for (int k = 0; k < NS; k++) p.stats_out[(size_t)env * NS + k] = ovf ? -1 : st[k];
    return;
  }

  int32_t st[NS];
  bool ovf = false;
  EnvTargets<NS> trg;
  Pcg rp, rr;

  if constexpr (MODE == M3_RESET) {
    if (p.mask != nullptr && p.mask[env] == 0) return;
    for (int i = c.lane; i < 2 * nw; i += 64) E.rec[i] = grec[i];
    init_work();
    trg.load(p, env, false);
    if (p.refresh_only) {  // statistics (and the path overlay) of the current map, nothing else
      fresh_stats(st, ovf);
      if (ovf && c.lane == 0) atomicOr(p.err, 4);
      store_record();
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
    int pos[3] = {0, 0, 0};
    if (p.init_grids) {
      m3_load_bytes(c.dirt, c, p.init_grids + (size_t)env * c.n_cells);
      if (p.init_pos)
        for (int d = 0; d < 3; d++) pos[d] = p.init_pos[(size_t)env * 3 + d];
    } else {
      rp.load(p.rng[env].prob);
      rr.load(p.rng[env].rep);
      m3_reset_rng(c.dirt, c, p, cpl, rp, rr);
      if (c.lane == 0) {
        rr.store(p.rng[env].rep);
        rp.store(p.rng[env].prob);
      }
    }
    fresh_stats(st, ovf);
    int n_step = 0, iteration = 0, changes = 0;
    double ep_return = 0.0;
    if (p.set_state) {  // pcgrl_set_state: injected map, the caller's counters / return
      if (p.in_counters) {
        iteration = p.in_counters[(size_t)env * 4 + 0];
        changes = p.in_counters[(size_t)env * 4 + 1];
        n_step = p.in_counters[(size_t)env * 4 + 2];
      }
      if (p.in_ep_return) ep_return = p.in_ep_return[env];
    }
    trg.load(p, env, true);
    const double last_loss = trg.loss(p.cfg, st);
    if (ovf && c.lane == 0) atomicOr(p.err, 4);
    store_record();
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
    return;
  }

  if constexpr (MODE == M3_STEP || MODE == M3_ROLLOUT) {
    // ---- everything of the old state is requested before anything is waited for
    constexpr int CH = SC == 0 ? (M3C<0>::REC / 4 + 63) / 64 : 1;  // 16-byte chunks of the record per lane (size class 0)
    uint4 rch[CH];
    if (SC == 0) {
#pragma unroll
      for (int k = 0; k < CH; k++) {
        const int i = c.lane + 64 * k;
        rch[k] = i < c.L.rec_words / 4 ? ((const uint4 *)grec)[i] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    int pos[3] = {S->pos[0], S->pos[1], S->pos[2]};
    int n_step = S->n_step, iteration = S->iteration, changes = S->changes, flags = S->flags;
    double last_loss = S->last_loss, ep_return = S->ep_return;
    for (int k = 0; k < NS; k++) st[k] = S->stats[k];
    int action0 = p.actions[env];
    trg.load(p, env, false);
    init_work();
    if (SC == 0) {
#pragma unroll
      for (int k = 0; k < CH; k++) {
        const int i = c.lane + 64 * k;
        if (i < c.L.rec_words / 4) ((uint4 *)E.rec)[i] = rch[k];
      }
    } else {
      // Size class 1: the record is 49 KB at 15^3, most of it the move table and the cached start planes, which only a step
      // that CHANGES the map reads (statistics are recomputed only then, pcgrl_env.py:314-323; an auto-reset rebuilds every
      // table from the new map, but writes the whole record back, so it takes the old one, too).  The tile bits, overlay and
      // column masks come first, B x 16 bytes per lane in flight per round trip; the rest follows only when it is needed.
      const int s0 = (c.L.o_slots & ~3) / 4, s1 = c.L.rec_words / 4;
      m3_copy_batched<4>((uint4 *)E.rec, (const uint4 *)grec, 0, s0, c.lane);
      bool need_rest = true;
      if constexpr (MODE == M3_STEP) {
        const bool ok0 = action0 >= 0 && action0 < 2;
        const bool ch0 = ok0 && m3_bit(c.dirt, m3_cell(c, pos[2], pos[1], pos[0])) != (action0 != 0);
        const bool reset0 = p.auto_reset != 0 && p.update_only == 0 && iteration + 1 > p.cfg.max_iterations;
        need_rest = ch0 || reset0;
      }
#ifdef PCGRL_PHASE_TIMING  // development knob (timing builds only; cfg.solver_power is unused in 3-D): 1 = no second part, 2 = 4 per trip
      if (p.cfg.solver_power == 1) need_rest = false;
      if (p.cfg.solver_power == 2 && need_rest) {
        m3_copy_batched<4>((uint4 *)E.rec, (const uint4 *)grec, s0, s1, c.lane);
        need_rest = false;
      }
#endif
      if (need_rest) m3_copy_batched<16>((uint4 *)E.rec, (const uint4 *)grec, s0, s1, c.lane);
    }
    M3_MARK(0, 5);  // loads
    const int K = MODE == M3_ROLLOUT ? p.n_steps : 1;
    const size_t N = (size_t)p.n_envs;
    bool any_reset = false, whole_record = false, edited = false, mv_chg = false, upd_exit = false, over_dirty = false;
    bool ovf_any = false;
    int mv_cell = 0, col_word = 0;
    for (int k = 0; k < K; k++) {
      const size_t o = (size_t)k * N + (size_t)env;  // index of this step's outputs
      uint8_t *obs_k = p.obs == nullptr ? nullptr
                       : (MODE == M3_ROLLOUT && !p.obs_last_only ? p.obs + (size_t)k * N * (size_t)p.obs_env_bytes : p.obs);
      const bool want_obs = MODE == M3_ROLLOUT && (!p.obs_last_only || k == K - 1);  // (M3_STEP: the observe wave)
      // ---- step (envs/pcgrl_env.py:267-342 with narrow_rep.py:89-102)
      const int action = k == 0 ? action0 : p.actions[o];
      const bool bad = action < 0 || action >= 2;
      const bool upd_only = p.update_only != 0;
      iteration += upd_only ? 0 : 1;
      bool change = false;
      int ex = 0, ey = 0, ez = 0;
      if (!bad) {
        ez = pos[0], ey = pos[1], ex = pos[2];  // pos = (z, y, x)
        const int ci = m3_cell(c, ex, ey, ez);
        const bool old = m3_bit(c.dirt, ci);
        change = old != (action != 0);
        if (change) {
          if (c.lane == 0) {  // (LDS XORs without a result: nothing to wait for, where a read-modify-write is two round trips each)
            const int q = ey * c.X + ex;
            atomicXor(&c.dirt[ci >> 5], 1u << (ci & 31));
            atomicXor((uint32_t *)c.col + (q >> 1), (1u << ez) << (16 * (q & 1)));
          }
          col_word = (ey * c.X + ex) >> 1;
          // the edit changes at most 48 bytes of the move table and drops exactly the slots that accepted one of their cells
          if (edited) whole_record = true;  // (rollout: more than one edit per launch)
          edited = true;
#ifdef PCGRL_M3_HEADSPLIT
          PHASE_MARK(1);
#endif
          dirty_hdr |= m3_update_moves(c, ex, ey, ez, mv_chg, mv_cell);
#ifdef PCGRL_M3_HEADSPLIT
          PHASE_MARK(2);
#endif
        }
#ifdef PCGRL_M3_PHASES
        M3_MARK(6, 5);  // (development: edit + move-table update, apart from the position arithmetic below)
#endif
        m3_advance_pos(c, pos, n_step);
      } else if (c.lane == 0) {
        atomicOr(p.err, 1);
      }
      if (upd_only) {  // rep.update() only: map, position (the observe wave shows the stale overlay)
        if (change) flags |= ENV_STATS_DIRTY;  // the region count is no longer that of the map
        upd_exit = true;
        break;
      }
      changes += change ? 1 : 0;
      bool done = iteration > p.cfg.max_iterations;
      if (p.cfg.max_changes >= 0) done = done || changes > p.cfg.max_changes;
      const bool do_reset = done && p.auto_reset != 0;
      // the observation is assembled BEFORE the stats refresh (pcgrl_env.py:298-299 vs :314-323): it shows the path of
      // the previous stats update on the already edited map
      M3_MARK(1, 5);  // action + move-table update
      if (!do_reset && want_obs) m3_encode_obs<2 * DIM>(c.dirt, c.over, c, p, env, pos, true, (uint2 *)W.info, M3C<SC>::CELLS / 2, obs_k);
      M3_MARK(1, 5);  // observation (rollout mode)
      if (change) {
        const PM<PW> air = plane_of(c.dirt);
        const int32_t st_old[NS] = {st[0], st[1], st[2]};
        int rjob = 0;
        if constexpr (HELP) {  // the helper wave counts the regions while this wave searches
          if (c.lane == 0) {
            rjob = m3_ld(&mail.rseq) + 1;
            mail.r_eq = ey * c.X + ex;
            mail.r_ez = ez;
            mail.r_kind = (flags & ENV_STATS_DIRTY) ? 2 : (action == 0 ? 0 : 1);
            mail.r_old = st[0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            m3_st(&mail.rseq, rjob);
          }
          rjob = __builtin_amdgcn_readfirstlane(rjob);
        } else if (flags & ENV_STATS_DIRTY) {  // after pcgrl_update: from scratch, like the reference's get_stats
          st[0] = m3_regions<PW>(c, air, notx0, notxl);
        } else {
          PM<PW> A = air;  // the planes without the edited cell
          if (c.lane == ez) {
            PM<PW> e = pm_zero<PW>();
            pm_set(e, ey * c.X + ex);
            A = A & ~e;
          }
          st[0] = m3_regions_update<PW>(c, A, notx0, notxl, ey * c.X + ex, ez, action == 0, st[0]);
        }
        flags &= ~ENV_STATS_DIRTY;
        M3_MARK(2, 4);  // regions
        m3_paths<SC>(E, W, c, air, st, epoch, trip, dirty_full, ovf, WH, HELP_S ? &mail : nullptr PHASE_PASS);
        over_dirty = true;
        if constexpr (HELP) {
          while (__builtin_amdgcn_readfirstlane(m3_ld(&mail.rdone)) != rjob) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          st[0] = __builtin_amdgcn_readfirstlane(m3_ld(&mail.r_out));
        }
#ifdef PCGRL_M3_TAILSPLIT
        PHASE_MARK(2);
#endif
        if (ovf)  // queue overflow: reported (pcgrl_poll_error), no statistics of an unfinished search are handed out
          for (int i = 0; i < NS; i++) st[i] = st_old[i];
      }
      if (ovf) {
        // The kept statistics are those of the map BEFORE this edit and the unfinished search leaves nothing usable behind:
        // every cached start plane is dropped and the env is marked stale, so the next step that changes the map recomputes
        // region count and paths from scratch instead of updating a count that no longer belongs to the map.
        if (c.lane < n_slots) *(uint4 *)c.hdr(c.lane) = make_uint4(0u, 0u, 0u, 0u);
        dirty_hdr = (1u << n_slots) - 1u;
        flags |= ENV_STATS_DIRTY;
        ovf_any = true;
        ovf = false;  // (rollout: the later steps of the launch are judged on their own)
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
        if (!any_reset) {
          rp.load(p.rng[env].prob);  // (only this wave writes them, at the end)
          rr.load(p.rng[env].rep);
        }
        if (c.lane == 0) {
          latch_episode<NS>(p, env, S, ep_return, iteration, st);
          accumulate_episode<NS>(S);
        }
        m3_reset_rng(c.dirt, c, p, cpl, rp, rr);
        any_reset = true;
        whole_record = true;
        pos[0] = pos[1] = pos[2] = 0;
        fresh_stats(st, ovf);
        flags = 0;
        n_step = iteration = changes = 0;
        ep_return = 0.0;
        trg.load(p, env, true);
        last_loss = trg.loss(p.cfg, st);
        if (c.lane == 0) trg.commit(p, env);  // (a rollout may reset the env again: the next reset draws / pops anew)
        if (want_obs) m3_encode_obs<2 * DIM>(c.dirt, c.over, c, p, env, pos, false, (uint2 *)W.info, M3C<SC>::CELLS / 2, obs_k);
      }
    }
    if ((ovf || ovf_any) && c.lane == 0) atomicOr(p.err, 4);
    // ---- write back, once the observe wave has read the old state
    if (HELP && c.lane == 0) m3_st(&mail.exit, 1);
#ifdef PCGRL_M3_TAILSPLIT
    PHASE_MARK(5);
#endif
    if (MODE == M3_STEP && p.obs != nullptr) {
      while (__builtin_amdgcn_readfirstlane(m3_ld(&mail.obs_read)) != m3_observers<SC>()) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#ifdef PCGRL_M3_TAILSPLIT
    PHASE_MARK(6);
#endif
    if (whole_record) {
      store_record();
    } else if (SC == 0) {
      // size class 0: every piece fits one pass of the wave (nw <= 16 words, a slot <= 36): all LDS reads are issued before the
      // first store instead of one LDS round trip per piece
      constexpr int NSL = M3C<0>::SLOTS;
      const int l_nw = c.lane < nw ? c.lane : 0;
      const uint32_t v_dirt = c.dirt[l_nw], v_over = c.over[l_nw], v_col = E.rec[c.L.o_col + col_word];
      const uint2 v_mv = ((const uint2 *)(E.rec + c.L.o_mv))[mv_chg ? mv_cell : 0];
      const uint32_t dm = dirty_hdr | dirty_full;
      uint32_t v_slot[NSL];
#pragma unroll
      for (int s = 0; s < NSL; s++) {
        const int n = ((dirty_full >> s) & 1u) ? c.L.slot_words : M3_SLOT_HDR;
        v_slot[s] = (s < n_slots && ((dm >> s) & 1u) && c.lane < n) ? E.rec[c.L.o_slots + s * c.L.slot_words + c.lane] : 0u;
      }
      if (edited) {
        if (c.lane < nw) grec[c.lane] = v_dirt;
        if (c.lane == 0) grec[c.L.o_col + col_word] = v_col;
        if (mv_chg) ((uint2 *)(grec + c.L.o_mv))[mv_cell] = v_mv;
      }
      if (over_dirty && c.lane < nw) grec[c.L.o_over + c.lane] = v_over;
#pragma unroll
      for (int s = 0; s < NSL; s++) {
        const int n = ((dirty_full >> s) & 1u) ? c.L.slot_words : M3_SLOT_HDR;
        if (s < n_slots && ((dm >> s) & 1u) && c.lane < n) grec[c.L.o_slots + s * c.L.slot_words + c.lane] = v_slot[s];
      }
    } else {
      if (edited) {  // the tile bits, one column mask, the changed rows of the move table
        for (int i = c.lane; i < nw; i += 64) grec[i] = c.dirt[i];
        if (c.lane == 0) grec[c.L.o_col + col_word] = E.rec[c.L.o_col + col_word];
        if (mv_chg) ((uint2 *)(grec + c.L.o_mv))[mv_cell] = ((const uint2 *)(E.rec + c.L.o_mv))[mv_cell];
      }
      if (over_dirty)  // new statistics: the overlay
        for (int i = c.lane; i < nw; i += 64) grec[c.L.o_over + i] = c.over[i];
      for (int s = 0; s < n_slots; s++) {
        if ((((dirty_hdr | dirty_full) >> s) & 1u) == 0u) continue;
        const int o0 = c.L.o_slots + s * c.L.slot_words;
        const int n = ((dirty_full >> s) & 1u) ? c.L.slot_words : M3_SLOT_HDR;
        for (int i = c.lane; i < n; i += 64) grec[o0 + i] = E.rec[o0 + i];
      }
    }
    if (any_reset && c.lane == 0) {
      rr.store(p.rng[env].rep);
      rp.store(p.rng[env].prob);
    }
    if (c.lane == 0) {
      S->pos[0] = pos[0];
      S->pos[1] = pos[1];
      S->pos[2] = pos[2];
      S->n_step = n_step;
      S->flags = flags;
      if (!upd_exit) {
        trg.write_ctrl_obs(p, env, st);
        trg.commit(p, env);
        S->iteration = iteration;
        S->changes = changes;
        S->last_loss = last_loss;
        S->ep_return = ep_return;
        for (int k = 0; k < NS; k++) S->stats[k] = st[k];
      }
    }
    M3_MARK(5, 5);
    PHASE_FLUSH();
    TRACE_PUT(0, _tr0);
    TRACE_PUT(1, TRACE_NOW());
    TRACE_DRAIN();
    TRACE_PUT(2, TRACE_NOW());
  }
}

}  // namespace pcgrl
