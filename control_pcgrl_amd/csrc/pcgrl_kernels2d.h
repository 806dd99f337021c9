// pcgrl_kernels2d.h -- gfx950 kernels for the 2-D problems (binary, zelda, sokoban).
//
// Execution model: a wavefront (64 lanes) is split into groups of LPE lanes, one group per env, one lane per
// map ROW.  A row of W<=32 tiles is held as bit masks in VGPRs (one 32-bit word per bit-plane), so
//   * horizontal neighbours are 1-bit shifts inside the lane,
//   * vertical neighbours are one DPP row_shr:1 / row_shl:1 lane shift (a DPP row is 16 lanes = one 16x16 env),
//   * "is any frontier left" is a 64-bit wave ballot, sliced per group,
//   * "first cell in row-major order" is ctz(ballot) then ctz(row word).
// Flood fill / BFS therefore run on whole rows per instruction and never touch memory.  LDS is used only to
// assemble the byte-granular one-hot observation rows before they are written with 16-byte stores.
// No MFMA: this is integer/bit work, bounded by HBM writes of the observation tensor.
//
// Reference semantics restated here (paths relative to the reference's control_pcgrl/):
//   envs/pcgrl_env.py:158-188, :267-342   reset / step orchestration
//   envs/reps/{narrow,turtle,wide}_rep.py update()
//   envs/helper.py:173-276                flood fill, BFS ("dijkstra"), longest path
//   envs/probs/binary/binary_prob.py:152-158, zelda/zelda_ctrl_prob.py:90-168, sokoban/sokoban_prob.py:160-180
//   control_wrappers.py:216-244, :318-345 reward = delta of weighted target distance
//   wrappers.py:407-437, :232-257, :140-150  cropped one-hot observation
#pragma once
#include <hip/hip_runtime.h>

#include "pcgrl_common.h"

namespace pcgrl {

// Development aid: -DPCGRL_PHASE_TIMING accumulates s_memtime deltas of the simulate wave's phases into p.err[8..]
// (read back by tools/phase_timing.py).  Not compiled into the shipped library.
#ifdef PCGRL_PHASE_TIMING
#define PHASE_DECL() uint32_t _ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t _t_prev = __builtin_readcyclecounter(); uint64_t _w0 = wall_clock64()
#define PHASE_ARG , uint32_t *_ph, uint64_t &_t_prev
#define PHASE_PASS , _ph, _t_prev
#define PHASE_MARK(i)                                  \
  do {                                                 \
    uint64_t _t = __builtin_readcyclecounter();        \
    _ph[i] += (uint32_t)(_t - _t_prev);                \
    _t_prev = _t;                                      \
  } while (0)
#define PHASE_FLUSH()                                                                              \
  do {                                                                                             \
    _ph[7] = (uint32_t)(wall_clock64() - _w0);                                                     \
    if (threadIdx.x == 0)                                                                          \
      for (int _i = 0; _i < 8; _i++) ((unsigned long long *)(p.err + 64))[blockIdx.x * 8 + _i] += _ph[_i]; \
  } while (0)
#else
#define PHASE_DECL() \
  do {               \
  } while (0)
#define PHASE_ARG
#define PHASE_PASS
#define PHASE_MARK(i) \
  do {                \
  } while (0)
#define PHASE_FLUSH() \
  do {                \
  } while (0)
#endif

// Development aid: -DPCGRL_WAVE_TRACE records, per workgroup, the 100 MHz wall-clock stamps of the simulate wave
// (start, end) and of the observe wave (start, stores issued, stores acknowledged) of the LAST launch into p.err[64..]
// (read back by tools/wave_trace.py).  Not compiled into the shipped library.
#ifdef PCGRL_WAVE_TRACE
#define TRACE_DECL() const unsigned long long _tr0 = wall_clock64()
#define TRACE_PUT(slot, v)                                                                          \
  do {                                                                                              \
    if ((threadIdx.x & 63) == 0) ((unsigned long long *)(p.err + 64))[(size_t)blockIdx.x * 8 + (slot)] = (v); \
  } while (0)
#define TRACE_NOW() wall_clock64()
#define TRACE_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define TRACE_DECL() \
  do {               \
  } while (0)
#define TRACE_PUT(slot, v) \
  do {                     \
  } while (0)
#define TRACE_NOW() 0
#define TRACE_DRAIN() \
  do {                \
  } while (0)
#endif

// ------------------------------------------------------------------------------------------------ PCG64
struct U128 {
  uint64_t hi, lo;
};
__host__ __device__ inline U128 mul128(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo * b.lo;
#ifdef __HIP_DEVICE_COMPILE__
  r.hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
#else
  r.hi = (uint64_t)(((unsigned __int128)a.lo * b.lo) >> 64) + a.hi * b.lo + a.lo * b.hi;
#endif
  return r;
}
__host__ __device__ inline U128 add128(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
}
#define PCG_MULT_HI 0x2360ED051FC65DA4ULL
#define PCG_MULT_LO 0x4385DF649FCCF645ULL

struct Pcg {
  U128 s, inc;
  __device__ inline void load(const uint64_t *p) {
    s.hi = p[0];
    s.lo = p[1];
    inc.hi = p[2];
    inc.lo = p[3];
  }
  __device__ inline void store(uint64_t *p) const {
    p[0] = s.hi;
    p[1] = s.lo;
  }
  __device__ inline uint64_t next() {
    s = add128(mul128(s, U128{PCG_MULT_HI, PCG_MULT_LO}), inc);
    uint64_t x = s.hi ^ s.lo;
    unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  __device__ inline double next_double() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  __device__ inline void jump(const JumpEntry &j) {
    s = add128(mul128(U128{j.a_hi, j.a_lo}, s), mul128(U128{j.g_hi, j.g_lo}, inc));
  }
};

// ------------------------------------------------------------------------------------------------ row-mask helpers
// A map row is a bit mask M: uint32_t for W <= 32, uint64_t for W <= 64.
__device__ inline int popc_m(uint32_t x) { return __popc(x); }
__device__ inline int popc_m(uint64_t x) { return __popcll(x); }
__device__ inline int ctz_m(uint32_t x) { return __builtin_ctz(x); }
__device__ inline int ctz_m(uint64_t x) { return __builtin_ctzll(x); }
__device__ inline uint32_t brev_m(uint32_t x) { return __brev(x); }
__device__ inline uint64_t brev_m(uint64_t x) { return __brevll(x); }

// ------------------------------------------------------------------------------------------------ lane groups
template <int LPE>
struct Grp {
  static_assert(LPE == 8 || LPE == 16 || LPE == 32 || LPE == 64, "lanes per env");
  int lane, row, gbase;
  __device__ inline void init() {
    lane = (int)__lane_id();
    row = lane & (LPE - 1);
    gbase = lane - row;
  }
  // value held by lane-1 / lane+1 of the same group (0 at the group edge)
  __device__ inline uint32_t from_above(uint32_t v) const {
    uint32_t r;
    if constexpr (LPE <= 16) {
      r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xF, 0xF, true);
      if constexpr (LPE < 16) r = row == 0 ? 0u : r;
    } else {
      // gfx9 DPP wave shift: one lane across the whole wavefront (crosses the 16-lane DPP rows)
      r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, true);
      if constexpr (LPE < 64) r = row == 0 ? 0u : r;  // (64 rows: the group is the wave, bound_ctrl gives lane 0 its zero)
    }
    return r;
  }
  __device__ inline uint32_t from_below(uint32_t v) const {
    uint32_t r;
    if constexpr (LPE <= 16) {
      r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /*row_shl:1*/, 0xF, 0xF, true);
      if constexpr (LPE < 16) r = row == LPE - 1 ? 0u : r;
    } else {
      r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xF, 0xF, true);
      if constexpr (LPE < 64) r = row == LPE - 1 ? 0u : r;
    }
    return r;
  }
  __device__ inline uint64_t from_above(uint64_t v) const {
    return (uint64_t)from_above((uint32_t)v) | ((uint64_t)from_above((uint32_t)(v >> 32)) << 32);
  }
  __device__ inline uint64_t from_below(uint64_t v) const {
    return (uint64_t)from_below((uint32_t)v) | ((uint64_t)from_below((uint32_t)(v >> 32)) << 32);
  }
  // this group's slice of a wave ballot
  __device__ inline uint64_t gballot(bool p) const {
    uint64_t b = __ballot(p);
    if constexpr (LPE == 64) return b;
    return (b >> gbase) & ((1ull << LPE) - 1ull);
  }
  __device__ inline bool gany(bool p) const { return gballot(p) != 0; }
  // sum over the group's lanes (result in every lane): DPP butterfly inside a 16-lane row
  // (quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror), bpermute only across rows
  __device__ inline uint32_t gsum(uint32_t v) const {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    if constexpr (LPE >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    if constexpr (LPE >= 32) v += (uint32_t)__shfl_xor((int)v, 16, 64);
    if constexpr (LPE >= 64) v += (uint32_t)__shfl_xor((int)v, 32, 64);
    return v;
  }
  // max over the group's lanes (result in every lane), same butterfly as gsum
  __device__ inline uint32_t gmax(uint32_t v) const {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    if constexpr (LPE >= 16) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
    if constexpr (LPE >= 32) v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64));
    if constexpr (LPE >= 64) v = max(v, (uint32_t)__shfl_xor((int)v, 32, 64));
    return v;
  }
  __device__ inline uint32_t gmin(uint32_t v) const { return ~gmax(~v); }
  // this group's slice of an already computed wave ballot
  __device__ inline uint64_t gslice(uint64_t b) const {
    if constexpr (LPE == 64) return b;
    return (b >> gbase) & ((1ull << LPE) - 1ull);
  }
  // broadcast from the group's lane `src_row`
  __device__ inline uint32_t gbcast(uint32_t v, int src_row) const { return (uint32_t)__shfl((int)v, gbase + src_row, 64); }
};

// 4-neighbour dilation of a row-mask set (without the set itself; callers AND with the passable mask,
// which also removes the bit shifted past column W-1)
template <int LPE, typename M>
__device__ inline M expand(const Grp<LPE> &g, M f) {
  return (f << 1) | (f >> 1) | g.from_above(f) | g.from_below(f);
}

// One BFS level: returns the new frontier expand(front) & free_cells and removes it from free_cells.  16-row maps with
// 32-bit masks (the headline kernels): six instructions instead of the compiler's seven -- the row shifts ride on the ORs
// (v_or_b32_dpp; the compiler re-associates the four-way OR into 2 x v_mov_b32_dpp + v_or3_b32) and both results come from
// one three-input bit operation each.  The two plain shifts come first: a DPP operand written by the preceding vector
// instruction needs two wait states.  bitop3 truth tables (index = a<<2 | b<<1 | free): 0xA8 = (a | b) & free,
// 0x02 = free & ~(a | b).
template <int LPE, typename M>
__device__ __attribute__((always_inline)) inline M bfs_level(const Grp<LPE> &g, M front, M &free_cells) {
  if constexpr (LPE == 16 && sizeof(M) == 4) {
    uint32_t nb, a, b, fr = (uint32_t)free_cells;
    asm("v_lshlrev_b32 %1, 1, %4\n\t"
        "v_lshrrev_b32 %2, 1, %4\n\t"
        "v_or_b32_dpp %1, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or_b32_dpp %2, %4, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_bitop3_b32 %0, %1, %2, %3 bitop3:0xa8\n\t"
        "v_bitop3_b32 %3, %1, %2, %3 bitop3:2"
        : "=&v"(nb), "=&v"(a), "=&v"(b), "+v"(fr)
        : "v"((uint32_t)front));
    free_cells = (M)fr;
    return (M)nb;
#ifndef PCGRL_NO_BFS32_ASM  // (development: A/B builds, tools/ab_bench.sh)
  } else if constexpr (LPE == 32 && sizeof(M) == 4) {
    // 32-row maps (two envs per wavefront): the wave shifts cross the group edge at lanes 31 / 32, so the shifted rows are
    // masked by per-lane constants inside the DPP instruction itself (v_and_b32_dpp): seven instructions instead of eleven
    const uint32_t mu = g.row == 0 ? 0u : ~0u, md = g.row == 31 ? 0u : ~0u;
    uint32_t nb, a, b, u, d, fr = (uint32_t)free_cells;
    asm("v_lshlrev_b32 %1, 1, %6\n\t"
        "v_lshrrev_b32 %2, 1, %6\n\t"
        "v_and_b32_dpp %3, %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_and_b32_dpp %4, %6, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or3_b32 %1, %1, %2, %3\n\t"
        "v_bitop3_b32 %0, %1, %4, %5 bitop3:0xa8\n\t"
        "v_bitop3_b32 %5, %1, %4, %5 bitop3:2"
        : "=&v"(nb), "=&v"(a), "=&v"(b), "=&v"(u), "=&v"(d), "+v"(fr)
        : "v"((uint32_t)front), "v"(mu), "v"(md));
    free_cells = (M)fr;
    return (M)nb;
#endif
  } else if constexpr (LPE == 64 && sizeof(M) == 4) {
    // 64-row maps, 32-bit masks: the same six instructions with the DPP wave shifts (the group is the whole wavefront, so
    // bound_ctrl supplies the zeros at rows 0 and 63)
    uint32_t nb, a, b, fr = (uint32_t)free_cells;
    asm("v_lshlrev_b32 %1, 1, %4\n\t"
        "v_lshrrev_b32 %2, 1, %4\n\t"
        "v_or_b32_dpp %1, %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or_b32_dpp %2, %4, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_bitop3_b32 %0, %1, %2, %3 bitop3:0xa8\n\t"
        "v_bitop3_b32 %3, %1, %2, %3 bitop3:2"
        : "=&v"(nb), "=&v"(a), "=&v"(b), "+v"(fr)
        : "v"((uint32_t)front));
    free_cells = (M)fr;
    return (M)nb;
  } else if constexpr (LPE == 64 && sizeof(M) == 8) {
    // 64 x 64 maps (the reference's binary_bigger / zelda_bigger): a level in twelve 32-bit instructions on the two halves of
    // the row masks -- the one-bit shifts across the halves are v_alignbit_b32, the row shifts ride on the ORs, both results
    // of a half come from one three-input bit operation each.  (The compiler's form: two 64-bit shifts, four DPP moves, and a
    // chain of 64-bit ORs / ANDs, ~22 instructions; a launch of 4096 such envs ends with the env whose path sweeps are longest.)
    uint32_t flo = (uint32_t)front, fhi = (uint32_t)((uint64_t)front >> 32);
    uint32_t rlo = (uint32_t)free_cells, rhi = (uint32_t)((uint64_t)free_cells >> 32);
    uint32_t nlo, nhi, alo, ahi, blo, bhi;
    asm("v_lshlrev_b32 %2, 1, %8\n\t"
        "v_alignbit_b32 %3, %9, %8, 31\n\t"
        "v_alignbit_b32 %4, %9, %8, 1\n\t"
        "v_lshrrev_b32 %5, 1, %9\n\t"
        "v_or_b32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or_b32_dpp %3, %9, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or_b32_dpp %4, %8, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_or_b32_dpp %5, %9, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_bitop3_b32 %0, %2, %4, %6 bitop3:0xa8\n\t"
        "v_bitop3_b32 %6, %2, %4, %6 bitop3:2\n\t"
        "v_bitop3_b32 %1, %3, %5, %7 bitop3:0xa8\n\t"
        "v_bitop3_b32 %7, %3, %5, %7 bitop3:2"
        : "=&v"(nlo), "=&v"(nhi), "=&v"(alo), "=&v"(ahi), "=&v"(blo), "=&v"(bhi), "+v"(rlo), "+v"(rhi)
        : "v"(flo), "v"(fhi));
    free_cells = (M)((uint64_t)rlo | ((uint64_t)rhi << 32));
    return (M)((uint64_t)nlo | ((uint64_t)nhi << 32));
  } else {
    const M nb = expand(g, front) & free_cells;
    free_cells ^= nb;
    return nb;
  }
}

// first set cell in row-major order: one bit in one lane of the group (0 everywhere if the set is empty)
template <int LPE, typename M>
__device__ inline M first_rowmajor(const Grp<LPE> &g, M x) {
  uint64_t gb = g.gballot(x != 0);
  int fl = gb ? __builtin_ctzll(gb) : -1;
  return g.row == fl ? (x & (M(0) - x)) : M(0);
}

// all cells of the horizontal runs of `a` that contain a bit of s (s subset of a): carry-propagation fill
template <typename M>
__device__ inline M hfill(M s, M a) {
  M up = (a & ~(a + s)) | s;
  M ar = brev_m(a), sr = brev_m(s);
  M dn = brev_m((ar & ~(ar + sr)) | sr);
  return up | dn;
}

// helper.py:200-210 calc_num_regions: number of 4-connected components of `avail`.
// Components are peeled off in row-major order of their first cell; each fill alternates an O(1) horizontal
// run fill with a one-row vertical step until it stops growing.
template <int LPE, typename M>
__device__ inline int count_regions(const Grp<LPE> &g, M avail) {
  M remaining = avail;
  int n = 0;
  while (true) {
    uint64_t gb = g.gballot(remaining != 0);
    if (__ballot(gb != 0) == 0) break;
    int fl = gb ? __builtin_ctzll(gb) : -1;
    M f = g.row == fl ? (remaining & (M(0) - remaining)) : M(0);
    f = hfill(f, remaining);
    while (true) {  // two rounds per trip: one ballot + branch per two vertical steps
      M v = (g.from_above(f) | g.from_below(f)) & remaining & ~f;
      f = hfill(f | v, remaining);
      v = (g.from_above(f) | g.from_below(f)) & remaining & ~f;
      if (__ballot(v != 0) == 0) break;
      f = hfill(f | v, remaining);
    }
    remaining &= ~f;
    n += gb != 0;
  }
  return n;
}

// helper.py:255-276 calc_longest_path + :200-210 calc_num_regions over the same passable set, split into pieces:
//
//   component_fars(comps)      for every component of `comps` (row-major order of first cell): level-synchronous BFS
//                              from its first cell; the last non-empty frontier holds the farthest cells and np.argmax
//                              picks its first cell in row-major order (:265).  Isolated cells are their own "far".
//   eccentricity(src, pass)    multi-source BFS from a set of far cells (components are disjoint): number of levels
//                              until every frontier is empty = max over those components of the second sweep's maximum
//                              (:266-268), plus the last non-empty frontier (marks the component(s) attaining it).
//   flood(seed, avail)         all cells connected to `seed`.
//
// regions = number of far cells (one per component); path-length = eccentricity(all fars).
//
// INCREMENTAL UPDATE.  A step edits ONE cell, and a component's far cell depends only on that component's cells, so
// the far cells of every component that does not touch the edited cell are unchanged.  The engine keeps two extra
// row masks per env -- `fars` and `best` (last frontier of the sweep that produced the current path-length) -- and on a
// change re-runs the first sweep only inside the affected component(s); the second sweep runs from the new far cells
// only, unless a component that attained the old maximum was touched (then from all far cells).
template <int LPE, typename M>
__device__ inline M flood(const Grp<LPE> &g, M seed, M avail) {
  if constexpr (LPE <= 16) {
    // small maps: plain frontier expansion, 8 levels per trip (7 instructions per level, no run fill)
    M front = seed & avail, free_cells = avail & ~front;
    while (true) {
#pragma unroll
      for (int u = 0; u < 8; u++) front = bfs_level(g, front, free_cells);
      if (__ballot(front != 0) == 0) break;
    }
    return avail & ~free_cells;
  } else {
    // larger maps: O(1) horizontal run fill + one vertical step per round (corridors cost one round per row, not one
    // level per cell), four rounds per trip
    M f = hfill(seed & avail, avail);
    while (true) {
      M v;
#pragma unroll
      for (int u = 0; u < 3; u++) {
        v = (g.from_above(f) | g.from_below(f)) & avail & ~f;
        f = hfill(f | v, avail);
      }
      v = (g.from_above(f) | g.from_below(f)) & avail & ~f;
      if (__ballot(v != 0) == 0) break;
      f = hfill(f | v, avail);
    }
    return f;
  }
}

// Level-synchronous BFS of every group from `src` inside `avail`.  Per group: the number of levels (eccentricity of the
// source set) and the last non-empty frontier.  The loop does SWEEP_UNROLL levels per trip and carries no cross-lane
// state: every lane only remembers the deepest level at which IT received new cells and those cells; the group's depth
// is a DPP max-reduction after the loop.  One ballot + branch per trip instead of per level shortens the dependent
// chain that bounds the launch at small batches (a frontier that died stays empty, so testing the trip's last level
// is enough).
constexpr int SWEEP_UNROLL = 6;
#ifndef SWEEP_UNTRACKED_UNROLL
#define SWEEP_UNTRACKED_UNROLL 6
#endif
#ifndef SWEEP_TRACKED_TRIPS
#define SWEEP_TRACKED_TRIPS 1
#endif
// Level-synchronous BFS from `src` inside `avail`: len = number of levels, last = the cells of the final level,
// visited = everything reached.  SWEEP_UNROLL levels per loop trip (one ballot per trip).  The per-lane bookkeeping of
// "my last non-empty level and its cells" costs 4-5 of the 12 instructions of a level, and it only matters for the
// final level, so after the first SWEEP_TRACKED_TRIPS trips (which cover the many short sweeps) the loop runs without
// it, remembering per group the state at the end of its last trip with a live frontier; that one trip is then replayed
// with the bookkeeping.  The launch time is set by the env with the longest path, i.e. by exactly these long sweeps.
template <int LPE, typename M>
__device__ inline void sweep(const Grp<LPE> &g, M src, M avail, int &len, M &last, M &visited) {
  M front = src & avail;
  M free_cells = avail & ~front;
  M mynb = M(0);
  int mylev = 0, lev = 0;
  bool more = true;
  for (int trip = 0; trip < SWEEP_TRACKED_TRIPS && more; trip++) {
#pragma unroll
    for (int u = 0; u < SWEEP_UNROLL; u++) {
      const M nb = bfs_level(g, front, free_cells);
      lev++;
      mylev = nb ? lev : mylev;
      mynb = nb ? nb : mynb;
      front = nb;
    }
    more = __ballot(front != 0) != 0;
  }
  if (more) {
    M s_front = front, s_free = free_cells;
    int s_lev = lev;
    while (true) {
#pragma unroll
      for (int u = 0; u < SWEEP_UNTRACKED_UNROLL; u++) front = bfs_level(g, front, free_cells);
      lev += SWEEP_UNTRACKED_UNROLL;
      const uint64_t bal = __ballot(front != 0);
      const bool alive = g.gslice(bal) != 0;
      s_front = alive ? front : s_front;
      s_free = alive ? free_cells : s_free;
      s_lev = alive ? lev : s_lev;
      if (bal == 0) break;
    }
    // replay each group's last live trip with the bookkeeping (s_front are the cells of level s_lev)
    M f = s_front, fr = s_free;
    int l = s_lev;
    mylev = f ? l : mylev;
    mynb = f ? f : mynb;
#pragma unroll
    for (int u = 0; u < SWEEP_UNTRACKED_UNROLL; u++) {
      const M nb = bfs_level(g, f, fr);
      l++;
      mylev = nb ? l : mylev;
      mynb = nb ? nb : mynb;
      f = nb;
    }
  }
  len = (int)g.gmax((uint32_t)mylev);
  last = (len > 0 && mylev == len) ? mynb : M(0);
  visited = avail & ~free_cells;
}

template <int LPE, typename M>
__device__ inline M component_fars(const Grp<LPE> &g, M comps) {
  const M iso = comps & ~expand(g, comps);
  M remaining = comps & ~iso, fars = iso;
  while (true) {
    uint64_t gb = g.gballot(remaining != 0);
    if (__ballot(gb != 0) == 0) break;
    int fl = gb ? __builtin_ctzll(gb) : -1;
    M seed = g.row == fl ? (remaining & (M(0) - remaining)) : M(0);
    int depth;
    M last, vis;
    sweep(g, seed, remaining, depth, last, vis);
    // non-isolated components always reach level 1, so `last` is empty only for groups without a seed
    fars |= first_rowmajor(g, last);
    remaining &= ~vis;
  }
  return fars;
}

template <int LPE, typename M>
__device__ inline void eccentricity(const Grp<LPE> &g, M src, M pass, int &len, M &last) {
  M vis;
  sweep(g, src, pass, len, last, vis);
}

// from scratch (reset / stats_for_grids)
template <int LPE, typename M>
__device__ inline void binary_stats_full(const Grp<LPE> &g, M pass, int &regions, int &path_len, M &fars,
                                         M &best) {
  fars = component_fars(g, pass);
  regions = (int)g.gsum((uint32_t)popc_m(fars));
  eccentricity(g, fars, pass, path_len, best);
}

// after editing the single cell `x` (row mask, 0 for groups without a change): old passable set p_old, new p_new
template <int LPE, typename M>
__device__ inline void binary_stats_update(const Grp<LPE> &g, M x, M p_old, M p_new, int &regions,
                                           int &path_len, M &fars, M &best PHASE_ARG, bool have_pre = false,
                                           M pre = M(0)) {
  const bool became_pass = g.gany((x & p_new) != 0);
  // cells of the affected components in the NEW map: the merged component of x, or the old component of x minus x
  // (have_pre: the component of x with x passable was flooded ahead of time, see PREFLOOD)
  const bool edited = g.gany(x != 0);
  M K = edited ? pre : M(0);
  if (__ballot(edited && !have_pre) != 0) {
    const M Kf = flood(g, x, became_pass ? p_new : p_old);
    K = have_pre ? K : Kf;
  }
  K = became_pass ? K : (K & ~x);
  const M touched = K | x;
  const bool hit = g.gany((best & touched) != 0);
  PHASE_MARK(3);  // flood
  const M newfars = component_fars(g, K);
  PHASE_MARK(4);  // first sweeps
  fars = (fars & ~touched) | newfars;
  const bool changed = g.gany(x != 0);
  int l;
  M b;
  eccentricity(g, hit ? fars : newfars, p_new, l, b);
  PHASE_MARK(5);  // second sweep
  if (changed) {
    regions = (int)g.gsum((uint32_t)popc_m(fars));
    if (hit || l > path_len) {
      path_len = l;
      best = b;
    }
  } else {
    (void)g.gsum(0u);  // keep the cross-lane reduction in uniform control flow
  }
}

// The same after editing SEVERAL cells X at once (action patches, the static-tile revert after a reset).  A component
// that neither contains nor touches a cell of X is the same set of cells before and after, so its far cell stands:
// drop the far cells of the old components that meet N[X] = X + neighbours, re-sweep the new components that meet N[X].
template <int LPE, typename M>
__device__ inline void binary_stats_update_multi(const Grp<LPE> &g, M X, M p_old, M p_new, int &regions, int &path_len,
                                                 M &fars, M &best) {
  const M nx = X | expand(g, X);
  const M k_new = flood(g, nx & p_new, p_new);
  // The far cells to drop are those of the OLD components that meet N[X].  No second flood is needed for them: a far cell f
  // (passable before) belongs to such a component iff f is in X or in k_new.  (=>) walk the old path from f to N[X] up to
  // its first cell in X or N[X]: everything before it is unchanged, and it -- or, if it is a cell of X that became solid,
  // its predecessor, a neighbour of X -- is an unchanged passable cell of N[X], so f's new component meets N[X].  (<=) walk
  // the new path from f to N[X] the same way: the first cell that is new (in X) or in N[X] has an unchanged predecessor in
  // N[X] that the old map connects to f.  (Round 3 flooded the old map, too: 1 200 of the 7 500 cycles of a 3 x 3 patch.)
  const M t_old = k_new | X;
  const bool hit = g.gany((best & t_old) != 0);
  const M newfars = component_fars(g, k_new);
  fars = (fars & ~t_old) | newfars;
  const bool changed = g.gany(X != 0);
  int l;
  M b;
  eccentricity(g, hit ? fars : newfars, p_new, l, b);
  if (changed) {
    regions = (int)g.gsum((uint32_t)popc_m(fars));
    if (hit || l > path_len) {
      path_len = l;
      best = b;
    }
  } else {
    (void)g.gsum(0u);
  }
}

// helper.py:225-240 run_dijkstra from a single source, reduced to "distance to the first target cell":
// level k >= 1 at which the frontier first meets targetA / targetB, or -1 if the frontier dies first.
template <int LPE, typename M>
__device__ inline void bfs_first_hit(const Grp<LPE> &g, M src, M avail, M targetA, M targetB,
                                     int &dA, int &dB) {
  constexpr uint32_t NONE = 0xFFFFu;
  M front = src & avail;
  M free_cells = avail & ~front;
  const bool needA = g.gany(targetA != 0), needB = g.gany(targetB != 0);
  uint32_t myA = NONE, myB = NONE;  // first level at which THIS lane's new cells meet the target
  uint32_t lev = 0;
  while (true) {
#pragma unroll
    for (int u = 0; u < SWEEP_UNROLL; u++) {
      const M nb = bfs_level(g, front, free_cells);
      lev++;
      myA = ((nb & targetA) != 0 && myA == NONE) ? lev : myA;
      myB = ((nb & targetB) != 0 && myB == NONE) ? lev : myB;
      front = nb;
    }
    // a group is finished when its frontier died or every target it looks for has been met
    const bool alive = g.gany(front != 0);
    const bool foundA = g.gany(myA != NONE), foundB = g.gany(myB != NONE);
    const bool pending = alive && ((needA && !foundA) || (needB && !foundB));
    if (__ballot(pending) == 0) break;
  }
  const uint32_t a = g.gmin(myA), b = g.gmin(myB);
  dA = (needA && a != NONE) ? (int)a : -1;
  dB = (needB && b != NONE) ? (int)b : -1;
}

// ------------------------------------------------------------------------------------------------ per-problem stats
template <int PROB>
struct ProbTraits;
template <>
struct ProbTraits<PCGRL_PROB_BINARY> {
  static constexpr int NT = 2, NB = 1, NS = 2;
  static constexpr int NAUX = 2;  // fars, best (incremental path-length state)
};
template <>
struct ProbTraits<PCGRL_PROB_ZELDA> {
  static constexpr int NT = 8, NB = 3, NS = 7;
  static constexpr int NAUX = 0;
};
template <>
struct ProbTraits<PCGRL_PROB_SOKOBAN> {
  static constexpr int NT = 5, NB = 3, NS = 7;
  static constexpr int NAUX = 0;
};

template <>
struct ProbTraits<PCGRL_PROB_MC3DMAZE> {
  static constexpr int NT = 2, NB = 1, NS = 3;
  static constexpr int NAUX = 0;
};

// The kernel argument block (Params, ~750 bytes = 12 cache lines) is new for every launch, so the first read of each of its
// lines misses the scalar cache, and the compiler reads fields where they are first used: the step kernel's prologue paid
// five to six DEPENDENT scalar-memory round trips before its first useful instruction.  One dword of every line, all
// requested at once at kernel entry (a single asm statement: its operands must be in registers together), turns that into
// one round trip; the later reads hit the scalar cache.
__device__ __attribute__((always_inline)) inline void touch_kernarg(const Params &p) {
  const int32_t *kw = (const int32_t *)&p;
  constexpr int L = (int)((sizeof(Params) + 63) / 64);
  static_assert(L <= 14, "touch_kernarg: one operand per 64-byte line");
  asm volatile("" ::"s"(kw[0]), "s"(kw[16 < sizeof(Params) / 4 ? 16 : 0]), "s"(kw[32 < sizeof(Params) / 4 ? 32 : 0]),
               "s"(kw[48 < sizeof(Params) / 4 ? 48 : 0]), "s"(kw[64 < sizeof(Params) / 4 ? 64 : 0]),
               "s"(kw[80 < sizeof(Params) / 4 ? 80 : 0]), "s"(kw[96 < sizeof(Params) / 4 ? 96 : 0]),
               "s"(kw[112 < sizeof(Params) / 4 ? 112 : 0]), "s"(kw[128 < sizeof(Params) / 4 ? 128 : 0]),
               "s"(kw[144 < sizeof(Params) / 4 ? 144 : 0]), "s"(kw[160 < sizeof(Params) / 4 ? 160 : 0]),
               "s"(kw[176 < sizeof(Params) / 4 ? 176 : 0]), "s"(kw[192 < sizeof(Params) / 4 ? 192 : 0]),
               "s"(kw[208 < sizeof(Params) / 4 ? 208 : 0]));
}

template <int LPE, typename M, bool HUGE>
__device__ void sokoban_solve(const Grp<LPE> &g, const Params &p, int env, bool need, M solid, M player, M crate, M target,
                              int &dist_win, int &sol_len);
// asynchronous stepping: the solver to a per-launch budget, on the env's own workspace (pcgrl_sokoban.h "resumable solver")
template <int LPE, typename M>
__device__ bool sokoban_solve_async(const Grp<LPE> &g, const Params &p, int env, bool need, M solid, M player, M crate, M target,
                                    const M *planes, int &dist_win, int &sol_len);
// helper wavefronts of the solver (pcgrl_sokoban.h): kernels launched with Params::sk_helpers carry three of them per
// workgroup, behind the simulate / observe waves
__device__ inline void sokoban_helper(const Params &p, int k, uint32_t *lds_heap);
// ... each with an expander wave behind it (A* stage k = heap wave k + expander wave k, see sk_stage_heap): kernels
// launched with sk_helpers = 3 carry 2 * 3 helper wavefronts, the heap waves first
__device__ inline void sokoban_expander(const Params &p, int k);
constexpr int SK_HELPER_LDS = 32 * 1024;  // dynamic LDS per helper wave (the top of its A* heap), behind the kernel's own
__device__ inline void sokoban_helpers_init();
__device__ inline void sokoban_helpers_release();
struct SokoHelpersGuard {  // the simulate wave lets its helpers go when it leaves the kernel, whichever way
  bool on;
  __device__ __attribute__((always_inline)) ~SokoHelpersGuard() {  // (out of line it costs every lane a 64-byte stack frame)
    if (on) sokoban_helpers_release();
  }
};

// cells that count for calc_num_regions: zelda everything but solid and door (zelda_ctrl_prob.py:104), sokoban
// everything but solid (sokoban_prob.py:166)
template <int PROB, typename M>
__device__ inline M region_cells(const M *b, M cm) {
  const M solid = b[0] & ~b[1] & ~b[2];
  if constexpr (PROB == PCGRL_PROB_ZELDA) return cm & ~(solid | (~b[0] & ~b[1] & b[2]));
  return cm & ~solid;
}

// Region count after editing the single cell `x` (one bit in one lane; 0 for groups that keep their count).  If the
// cell's membership flips, let U be the component of x in the map where x counts: on the other side of the edit U - x
// falls into `pieces` components (0..4), so the count moves by +-(pieces - 1).  Every cell of U - x is connected, not
// through x, to one of x's (up to four) member neighbours, so `pieces` is the number of classes of those neighbours:
//   * no member neighbour: pieces = 0;
//   * all of them connected to each other inside the eight cells around x (a walk of at most seven steps along that
//     ring): pieces = 1, WITHOUT looking at the rest of the map -- the common case;
//   * otherwise one flood from the first neighbour over the members other than x; neighbours it does not reach start
//     further floods (at most three more, rare).
// Round 1 / first pass of round 2: one flood for U plus one fill per piece, for every flipping edit.
template <int LPE, typename M>
__device__ inline int regions_update(const Grp<LPE> &g, M x, M w_old, M w_new, int regions_old) {
  const bool flip = g.gany((x & (w_old ^ w_new)) != 0);
  const bool became = g.gany((x & w_new & ~w_old) != 0);
  const M xs = flip ? x : M(0);
  const M W = (w_old | w_new) & ~xs;  // the members other than x (the two maps differ in x only)
  const M xh = xs | (xs << 1) | (xs >> 1);
  const M ring = (xh | g.from_above(xh) | g.from_below(xh)) & W;  // members among the eight cells around x
  const M nbr = expand(g, xs) & W;                                // members among the four neighbours
  // the ring class of the first neighbour (ring cells that follow each other on the ring are 4-adjacent)
  M loc = first_rowmajor(g, nbr);
#pragma unroll
  for (int u = 0; u < 7; u++) loc |= expand(g, loc) & ring;
  M rest = nbr & ~loc;
  int pieces = g.gany(nbr != 0) ? 1 : 0;
  if (__ballot(rest != 0) != 0) {  // (some group's neighbours are not connected around x: ask the map)
    const bool open = g.gany(rest != 0);
    M seen = flood(g, open ? loc : M(0), W);
    rest &= ~seen;
    while (__ballot(rest != 0) != 0) {
      pieces += g.gany(rest != 0) ? 1 : 0;
      const M f = flood(g, first_rowmajor(g, rest), W & ~seen);
      seen |= f;
      rest &= ~f;
    }
  }
  return flip ? (became ? regions_old - pieces + 1 : regions_old - 1 + pieces) : regions_old;
}

// regions_known >= 0: the caller already has the region count (incremental update in the step kernel)
// SK_HUGE (sokoban): also the search for levels with more than 128 crate / target pairs (false in the 16x16 kernels)
// SKA (sokoban, asynchronous stepping): the solver works to Params::sk_budget and may leave the search parked: *unfinished
// (uniform over a group) then tells the caller that dist-win / sol-length are not in yet and the statistics must not be used
template <int PROB, int LPE, typename M, bool SK_HUGE = true, bool SKA = false>
__device__ inline void compute_stats(const Grp<LPE> &g, const Params &p, int env, bool active, M *b, M colmask, int32_t *st,
                                     int regions_known = -1, bool *unfinished = nullptr) {
  if constexpr (PROB == PCGRL_PROB_BINARY) {
    // binary_prob.py:152-158: regions and path-length over "empty" (tile 0); b[1], b[2] receive fars / best
    M pass = active ? (~b[0] & colmask) : M(0);
    int reg, len;
    M fars, best;
    binary_stats_full(g, pass, reg, len, fars, best);
    st[0] = reg;
    st[1] = len;
    if (active) {
      b[1] = fars;
      b[2] = best;
    }
  } else if constexpr (PROB == PCGRL_PROB_ZELDA) {
    // zelda_ctrl_prob.py:90-168.  ids: 0 empty 1 solid 2 player 3 key 4 door 5 bat 6 scorpion 7 spider
    M cm = active ? colmask : M(0);
    M solid = b[0] & ~b[1] & ~b[2] & cm, door = ~b[0] & ~b[1] & b[2] & cm;
    M player = ~b[0] & b[1] & ~b[2] & cm, key = b[0] & b[1] & ~b[2] & cm, enemy = b[2] & (b[0] | b[1]) & cm;
    M walk = cm & ~(solid | door), walkd = cm & ~solid;
    uint32_t c01 = g.gsum((uint32_t)popc_m(player) | ((uint32_t)popc_m(key) << 16));
    uint32_t c23 = g.gsum((uint32_t)popc_m(door) | ((uint32_t)popc_m(enemy) << 16));
    int n_player = c01 & 0xFFFF, n_key = c01 >> 16, n_door = c23 & 0xFFFF, n_enemy = c23 >> 16;
    st[0] = n_player;
    st[1] = n_key;
    st[2] = n_door;
    st[3] = n_enemy;
    {
      const bool need_cr = active && regions_known < 0;
      int r = regions_known;
      if (__ballot(need_cr) != 0) {
        const int cr = count_regions(g, need_cr ? walk : M(0));
        r = need_cr ? cr : r;
      }
      st[4] = r;
    }
    int nearest = 0, plen = 0;
    bool one_player = n_player == 1;
    bool want_enemy = one_player && n_enemy > 0;
    bool want_path = one_player && n_key == 1 && n_door == 1;
    if (__ballot(want_enemy || want_path) != 0) {
      int dE, dK, dD, dummy;
      bfs_first_hit(g, (want_enemy || want_path) ? player : M(0), walk, want_enemy ? enemy : M(0), want_path ? key : M(0), dE, dK);
      if (want_enemy) nearest = dE > 0 ? dE : 0;
      if (__ballot(want_path) != 0) {
        bfs_first_hit(g, want_path ? key : M(0), walkd, want_path ? door : M(0), M(0), dD, dummy);
        if (want_path) plen = dK + dD;  // each term is -1 when unreachable
      }
    }
    st[5] = nearest;
    st[6] = plen;
  } else {
    // sokoban_prob.py:160-180 + sokoban_ctrl_prob.py:58-65.  ids: 0 empty 1 solid 2 player 3 crate 4 target
    M cm = active ? colmask : M(0);
    M solid = b[0] & ~b[1] & ~b[2] & cm, player = ~b[0] & b[1] & ~b[2] & cm;
    M crate = b[0] & b[1] & ~b[2] & cm, target = ~b[0] & ~b[1] & b[2] & cm;
    uint32_t c01 = g.gsum((uint32_t)popc_m(player) | ((uint32_t)popc_m(crate) << 16));
    int n_player = c01 & 0xFFFF, n_crate = c01 >> 16, n_target = (int)g.gsum((uint32_t)popc_m(target));
    int regions = regions_known;
    {
      const bool need_cr = active && regions_known < 0;
      if (__ballot(need_cr) != 0) {
        const int cr = count_regions(g, need_cr ? (cm & ~solid) : M(0));
        regions = need_cr ? cr : regions;
      }
    }
    int dist_win = p.cfg.dims[0] * p.cfg.dims[1] * (p.cfg.dims[0] + p.cfg.dims[1]);
    int sol_len = 0;
    bool need = active && n_player == 1 && n_crate == n_target && n_crate > 0 && regions == 1;
    if (__ballot(need) != 0) {
      if constexpr (SKA) {
        const M lvl[3] = {b[0] & cm, b[1] & cm, b[2] & cm};
        const bool u = sokoban_solve_async<LPE, M>(g, p, env, need, solid, player, crate, target, lvl, dist_win, sol_len);
        if (unfinished != nullptr) *unfinished = u;
      } else {
        sokoban_solve<LPE, M, SK_HUGE>(g, p, env, need, solid, player, crate, target, dist_win, sol_len);
      }
    }
    st[0] = n_player;
    st[1] = n_crate;
    st[2] = n_target;
    st[3] = regions;
    st[4] = dist_win;
    st[5] = sol_len;
    st[6] = n_crate > n_target ? n_crate - n_target : n_target - n_crate;
  }
}

// control_wrappers.py:318-345 get_loss against the config's static targets (plain mode: all operands are kernel arguments)
// Every problem's static targets are integers or infinite (Problem.static_trgs): the engine then hands the bounds over as
// int32 (Params::trg_lo_i / trg_hi_i, int_targets) and the distance to the target interval is integer arithmetic, one
// conversion and one multiplication per statistic instead of a chain of float64 compares and selects (sokoban, 7
// statistics: 2 400 -> ~900 cycles of every simulate wave).  Same values bit for bit: the float64 form computes the
// same small integer.
template <int NS>
__device__ inline double get_loss(const Params &p, const int32_t *st) {
  const pcgrl_config &c = p.cfg;
  double loss = 0.0;
  if (p.int_targets) {
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int32_t v = st[k];
      const int32_t d = max(max(p.trg_lo_i[k] - v, v - p.trg_hi_i[k]), 0);
      loss += c.has_trg[k] ? (-(double)d) * c.weights[k] : 0.0;
    }
    return loss;
  }
#pragma unroll
  for (int k = 0; k < NS; k++) {
    double v = (double)st[k];
    double d = v < c.trg_lo[k] ? c.trg_lo[k] - v : (v > c.trg_hi[k] ? v - c.trg_hi[k] : 0.0);
    loss += c.has_trg[k] ? (-d) * c.weights[k] : 0.0;
  }
  return loss;
}

// The c-th resampled target of env `env`'s control j under `seed` (pcgrl_set_target_resampling): a pure function of its
// arguments -- counter-based, so a captured launch re-targets at every replay with no host call -- uniform in [lo, hi):
// u * (hi - lo) + lo with u a 53-bit double, exactly numpy's  random() * (ub - lb) + lb  (control_wrappers.py:456-459).
__host__ __device__ inline uint64_t trg_mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__host__ __device__ inline double trg_resampled(uint64_t seed, int env, uint32_t c, int j, double lo, double hi) {
  const uint64_t r = trg_mix64(trg_mix64(seed + (uint64_t)c * 0x9e3779b97f4a7c15ull) ^
                               ((uint64_t)(uint32_t)env * 0xd1b54a32d192ed03ull + (uint64_t)(j + 1) * 0x8cb92ba72f3d8dd7ull));
  const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
  return u * (hi - lo) + lo;
}

// Target intervals of one env.  Plain mode: the config's static targets.  Controllable mode (control_wrappers.py:27-121):
// per-env targets in HBM; targets queued by pcgrl_queue_targets replace the control metrics' targets at the env's next
// reset (:174-178), which is when `take_pending` is set.  With target resampling switched on (TrgResample) every reset
// draws new control targets instead -- the reference's UniformNoiseyTargets.reset overwrites the queue with its draw
// (:453-471) -- from the env's own counter-based stream; the env's draw counter lives in bits 1.. of its trg_flag word.
template <int NS>
struct EnvTargets {
  double lo[NS], hi[NS];
  bool took_pending;
  int32_t new_flag;
  __device__ inline void load(const Params &p, int env, bool at_reset) {
    took_pending = false;
    new_flag = 0;
    if (p.trg == nullptr) {
#pragma unroll
      for (int k = 0; k < NS; k++) {
        lo[k] = p.cfg.trg_lo[k];
        hi[k] = p.cfg.trg_hi[k];
      }
      return;
    }
    const double *a = p.trg + (size_t)env * PCGRL_MAX_STATS * 2;
    const double *q = p.trg_pending + (size_t)env * PCGRL_MAX_STATS * 2;
    // (at a reset the flag word and the active targets may have been written by this very launch -- an earlier reset of the
    // same env inside a rollout: read past the vector L1)
    auto ld = [&](const double *x) -> double {
      if (!at_reset) return *x;
      const uint64_t v = __hip_atomic_load((const uint64_t *)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return __longlong_as_double((long long)v);
    };
    const int32_t flag = at_reset ? __hip_atomic_load(&p.trg_flag[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const TrgResample *rs = (const TrgResample *)p.trg - 1;
    const bool resample = at_reset && rs->enable != 0;
    took_pending = at_reset && ((flag & 1) != 0 || resample);
    new_flag = resample ? (int32_t)(((uint32_t)flag >> 1) + 1u) << 1 : (flag & ~1);
    uint32_t ctrl_mask = 0;
    for (int j = 0; j < p.cfg.n_ctrl; j++) ctrl_mask |= 1u << p.cfg.ctrl_idx[j];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const bool pend = took_pending && !resample && ((ctrl_mask >> k) & 1u);
      lo[k] = pend ? q[2 * k] : ld(a + 2 * k);
      hi[k] = pend ? q[2 * k + 1] : ld(a + 2 * k + 1);
    }
    if (resample) {
      const uint32_t c = (uint32_t)flag >> 1;
      for (int j = 0; j < p.cfg.n_ctrl; j++) {
        const double t = trg_resampled(rs->seed, env, c, j, rs->lo[j], rs->hi[j]);
#pragma unroll
        for (int k = 0; k < NS; k++)
          if (k == p.cfg.ctrl_idx[j]) lo[k] = hi[k] = t;
      }
    }
  }
  // one lane per env makes the queued (or resampled) targets the active ones (once per load: a second call is a no-op)
  __device__ inline void commit(const Params &p, int env) {
    if (p.trg == nullptr || !took_pending) return;
    double *a = p.trg + (size_t)env * PCGRL_MAX_STATS * 2;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      a[2 * k] = lo[k];
      a[2 * k + 1] = hi[k];
    }
    __hip_atomic_store(&p.trg_flag[env], new_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    took_pending = false;
  }
  // control_wrappers.py:318-345 get_loss
  __device__ inline double loss(const pcgrl_config &c, const int32_t *st) const {
    double l = 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      double v = (double)st[k];
      double d = v < lo[k] ? lo[k] - v : (v > hi[k] ? v - hi[k] : 0.0);
      l += c.has_trg[k] ? (-d) * c.weights[k] : 0.0;
    }
    return l;
  }
  // control_wrappers.py:189-214 observe_metric_trgs: (target / range, metric / range) per control metric
  __device__ inline void write_ctrl_obs(const Params &p, int env, const int32_t *st) const {
    if (p.ctrl_obs == nullptr) return;
    for (int j = 0; j < p.cfg.n_ctrl; j++) {
      const int s = p.cfg.ctrl_idx[j];
      double t = 0.0, v = 0.0;
#pragma unroll
      for (int k = 0; k < NS; k++)
        if (k == s) {
          t = (lo[k] + hi[k]) / 2;  // tuple target -> midpoint (:203-204)
          v = (double)st[k];
        }
      p.ctrl_obs[(size_t)env * 2 * p.cfg.n_ctrl + 2 * j] = (float)(t / p.cfg.ctrl_range[j]);
      p.ctrl_obs[(size_t)env * 2 * p.cfg.n_ctrl + 2 * j + 1] = (float)(v / p.cfg.ctrl_range[j]);
    }
  }
};

// ------------------------------------------------------------------------------------------------ tile <-> planes
template <int NB, typename M>
__device__ inline int tile_at(const M *b, int x) {
  int t = 0;
#pragma unroll
  for (int k = 0; k < NB; k++) t |= (int)((b[k] >> x) & M(1)) << k;
  return t;
}
template <int NB, typename M>
__device__ inline void set_tile(M *b, int x, int t) {
#pragma unroll
  for (int k = 0; k < NB; k++) b[k] = (b[k] & ~(M(1) << x)) | ((M)((t >> k) & 1) << x);
}

// ------------------------------------------------------------------------------------------------ ext: rep wrappers
// StaticTileRepresentation / MultiActionRepresentation (envs/reps/wrappers.py:234-376, :397-545) are a run-time
// option (p.ext) of the general (non-FAST) kernels.  Per-row extra state of one lane:
template <int NB, typename M>
struct ExtRow {
  M prot;       // static_tiles[row+1][1..W]: the cells of this map row the agent cannot change
  M stale[NB];  // tile planes of _bordered_map's interior while it lags behind _map (flags bit 0, see rep_update_ext)
  uint32_t flags, has32, val32;  // has32/val32: spare half of the representation RNG's last 64-bit draw
  __device__ inline void load(const Params &p, int env, int row, bool ok) {
    const M *xp = (const M *)p.xplanes + (size_t)env * (1 + NB) * p.cfg.dims[0] + row;
    const uint32_t *xs = p.xstate + (size_t)env * 4;
    flags = xs[0];
    has32 = xs[1];
    val32 = xs[2];
    prot = ok ? xp[0] : M(0);
    // (the lagging planes are read whether or not they are in use -- flags bit 0 -- and NOT masked here: a test of the flags
    // word right behind the loads would make everything the kernel loads after them wait for it, i.e. a second memory
    // round trip.  Every reader tests the flag itself: rep_update_ext, store.)
#pragma unroll
    for (int k = 0; k < NB; k++) stale[k] = ok ? xp[(size_t)(1 + k) * p.cfg.dims[0]] : M(0);
  }
  __device__ inline void store(const Params &p, int env, int row, bool ok, bool lead, bool planes) const {
    if (planes && ok) {
      M *xp = (M *)p.xplanes + (size_t)env * (1 + NB) * p.cfg.dims[0] + row;
      xp[0] = prot;
      if (flags & 1u) {
#pragma unroll
        for (int k = 0; k < NB; k++) xp[(size_t)(1 + k) * p.cfg.dims[0]] = stale[k];
      }
    }
    if (lead) {
      uint32_t *xs = p.xstate + (size_t)env * 4;
      xs[0] = flags;
      xs[1] = has32;
      xs[2] = val32;
    }
  }
};

// numpy/random/src/pcg64/pcg64.h pcg64_next32: one 64-bit draw serves two 32-bit requests, low half first
__device__ inline uint32_t pcg_next32(Pcg &r, uint32_t &has32, uint32_t &val32) {
  if (has32) {
    has32 = 0;
    return val32;
  }
  const uint64_t n = r.next();
  has32 = 1;
  val32 = (uint32_t)(n >> 32);
  return (uint32_t)n;
}
// Generator.integers(low, high), scalar int64 request with a range below 2^32: numpy/random/src/distributions/
// distributions.c random_bounded_uint64 -> buffered_bounded_lemire_uint32 (Lemire rejection on 32-bit draws)
__device__ inline int pcg_integers(Pcg &r, uint32_t &has32, uint32_t &val32, int low, int high) {
  const uint32_t rng = (uint32_t)(high - 1 - low);
  if (rng == 0) return low;
  const uint32_t ex = rng + 1;
  uint64_t m = (uint64_t)pcg_next32(r, has32, val32) * ex;
  uint32_t leftover = (uint32_t)m;
  if (leftover < ex) {
    const uint32_t threshold = (0u - ex) % ex;
    while (leftover < threshold) {
      m = (uint64_t)pcg_next32(r, has32, val32) * ex;
      leftover = (uint32_t)m;
    }
  }
  return low + (int)(m >> 32);
}
// bits [lo, hi) of a row mask, clipped to [0, W)  (numpy slices clip the same way)
template <typename M>
__device__ inline M bit_range(int lo, int hi, int W) {
  lo = max(lo, 0);
  hi = min(hi, W);
  if (hi <= lo) return M(0);
  const M upto_hi = hi >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << hi) - M(1));
  return upto_hi & ~((M(1) << lo) - M(1));
}

// StaticTileRepresentation.reset (reps/wrappers.py:265-319) after the wrapped representation's reset; `rr` is the
// representation RNG after the map draws.  Every lane replays the scalar draws; the (H+2)x(W+2) block of uniform
// draws is split by bordered row with the LCG skip-ahead table p.jump_b.
template <int NB, typename M>
__device__ inline void static_reset(const Params &p, int row, M *b, Pcg &rr, ExtRow<NB, M> &X) {
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  X.prot = M(0);
  X.flags = 0;
  if (p.cfg.static_prob > 0.0) {
    // :269-278: prob_static ~ U[0, static_prob) unless evaluating; static_tiles = random(bordered shape) < prob_static
    const double ps = p.cfg.static_eval ? p.cfg.static_prob : rr.next_double() * p.cfg.static_prob;
    Pcg blk = rr;
    rr.jump(p.jump_b[H + 2]);
    if (row < H) {
      blk.jump(p.jump_b[row + 1]);
      (void)blk.next();  // bordered column 0
      for (int x = 0; x < W; x++) X.prot |= (M)(blk.next_double() < ps ? 1 : 0) << x;
    }
  }
  for (int n = 0; n < p.cfg.n_static_walls; n++) {  // :280-299
    const int dim = pcg_integers(rr, X.has32, X.val32, 0, 2);
    const int size = dim == 0 ? H : W;  // :292 draws the other axis from shape[dim] as well
    const int wall_len = pcg_integers(rr, X.has32, X.val32, 1, size - 1);
    const int other = pcg_integers(rr, X.has32, X.val32, 0, size);
    const int along = pcg_integers(rr, X.has32, X.val32, 0, size - wall_len);
    // :295 the +1 border shift is applied to the slices of BOTH the bordered static mask and the unbordered map
    const int r0 = (dim == 0 ? along : other) + 1, c0 = (dim == 0 ? other : along) + 1;
    const int nr = dim == 0 ? wall_len : 1, nc = dim == 0 ? 1 : wall_len;
    if (row >= r0 && row < r0 + nr && row < H) {  // _map[wall_slices] = _wall_tile (tile 1)
      const M cm = bit_range<M>(c0, c0 + nc, W);
      b[0] |= cm;
#pragma unroll
      for (int k = 1; k < NB; k++) b[k] &= ~cm;
    }
    if (row + 1 >= r0 && row + 1 < r0 + nr && row < H) X.prot |= bit_range<M>(c0 - 1, c0 - 1 + nc, W);
    X.flags = 1;  // _bordered_map was synchronised before the walls were written: it lags until the first update
  }
}

// ------------------------------------------------------------------------------------------------ reset (RNG)
// envs/pcgrl_env.py:158-188 + reps/representation.py:65-76 + helper.py:491-494, :527-536.
// Every lane of the group replays the env's problem-RNG draws; the map draws of the representation RNG are
// split by row with an LCG skip-ahead so the 16 lanes generate their rows concurrently.
template <int PROB, int LPE, typename M>
__device__ inline void reset_from_rng(const Grp<LPE> &g, const Params &p, int env, bool active, M *b, int *pos,
                                      bool commit = true, ExtRow<ProbTraits<PROB>::NB, M> *X = nullptr,
                                      Pcg *reg_prob = nullptr, Pcg *reg_rep = nullptr) {
  // (the nullable X keeps the ExtRow of the general kernels in 32 bytes of scratch; passing it by reference with a flag
  // removes the scratch and makes the static-tile step 30 % SLOWER -- 9.7 -> 12.6 us at 4096 envs, measured -- so it stays)
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  if (!active) return;
  Pcg rp, rr;
  if (reg_prob != nullptr) {  // rollout kernel: the streams live in registers across steps
    rp = *reg_prob;
    rr = *reg_rep;
  } else {
    rp.load(p.rng[env].prob);
    rr.load(p.rng[env].rep);
  }
  double cdf[NT], total = 0.0;
#pragma unroll
  for (int t = 0; t < NT; t++) cdf[t] = rp.next_double();
#pragma unroll
  for (int t = 0; t < NT; t++) total += cdf[t];
  double acc = 0.0;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    acc += cdf[t] / total;
    cdf[t] = acc;
  }
#pragma unroll
  for (int t = 0; t < NT; t++) cdf[t] /= acc;
  if (PROB == PCGRL_PROB_BINARY) (void)rp.next();  // binary_prob.py:139-143 draws one more double
  pos[0] = pos[1] = 0;
  if (p.cfg.representation == PCGRL_REP_TURTLE) {  // turtle_rep.py:31-44, before the map
    pos[0] = (int)(rr.next_double() * (double)H);
    pos[1] = (int)(rr.next_double() * (double)W);
  }
  Pcg end = rr;
  end.jump(p.jump[H]);
  if (g.row < H) {
    rr.jump(p.jump[g.row]);
#pragma unroll
    for (int k = 0; k < NB; k++) b[k] = 0;
    for (int x = 0; x < W; x++) {
      double u = rr.next_double();
      int idx = 0;
#pragma unroll
      for (int t = 0; t < NT; t++) idx += cdf[t] <= u ? 1 : 0;  // searchsorted(cdf, u, side='right')
#pragma unroll
      for (int k = 0; k < NB; k++) b[k] |= (M)((idx >> k) & 1) << x;
    }
  }
  if (X != nullptr) {  // representation wrappers (p.ext)
    if (p.cfg.act_window[0] > 0) {  // narrow_rep.py:41-51 with MultiActionRepresentation.get_act_coords: first centre
      pos[0] = (p.cfg.act_window[0] - 1) / 2;
      pos[1] = (p.cfg.act_window[1] - 1) / 2;
    }
    if (p.cfg.static_tiles) {
#pragma unroll
      for (int k = 0; k < NB; k++) X->stale[k] = g.row < H ? b[k] : M(0);  // representation.py:65-76, before the walls
      static_reset<NB, M>(p, g.row, b, end, *X);
    }
  }
  if (reg_prob != nullptr) {
    *reg_prob = rp;
    *reg_rep = end;
  } else if (commit && g.row == 0) {
    end.store(p.rng[env].rep);
    rp.store(p.rng[env].prob);
  }
}

// ------------------------------------------------------------------------------------------------ observation
// One lane assembles one observation row (OW*C bytes) in its own LDS row (row stride = bytes + 16, which keeps
// 16-byte alignment and makes the 16-byte fills / read-backs bank-conflict free), scatters the one-hot bytes of
// its map row into it, and streams it out with 16-byte stores.
// (an ODD number of 16-byte chunks: with an even one -- sokoban-wide's 80-byte rows padded to 96 -- lanes r and r + 8
// of a group start in the same LDS bank)
__device__ inline int lds_row_stride(int row_bytes) {
  const int s = ((row_bytes + 15) & ~15) + 16;
  return (s & 16) ? s : s + 16;
}

// 16-byte observation store, write-through (sc1): the line leaves the XCD's L2 while the kernel is still running
// instead of being written back at the kernel boundary (MI355X_MICROARCH.md "boundary" / "publish-large" rows: a
// launch that leaves ~13 MB dirty pays ~3 us at its end).  Nothing in the kernel reads these bytes again.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ inline void store_obs16(void *dst, uint4 v) {
  u32x4_t w = {v.x, v.y, v.z, v.w};
  // the trailing s_nop covers the gfx9 hazard "VALU overwrites the data VGPRs of a >64-bit VMEM store" (2 wait states),
  // which the compiler's hazard recognizer cannot see through an asm statement
#ifndef PCGRL_OBS_STORE_SEL
#define PCGRL_OBS_STORE_SEL 0  // (development: A/B builds of other cache-policy bits, profiles/r05_dev_traces.md)
#endif
#if PCGRL_OBS_STORE_SEL == 1
#define PCGRL_OBS_STORE_MODS "sc1 nt"
#elif PCGRL_OBS_STORE_SEL == 2
#define PCGRL_OBS_STORE_MODS "nt"
#elif PCGRL_OBS_STORE_SEL == 3
#define PCGRL_OBS_STORE_MODS ""
#else
#define PCGRL_OBS_STORE_MODS "sc1"
#endif
  asm volatile("global_store_dwordx4 %0, %1, off " PCGRL_OBS_STORE_MODS "\n\ts_nop 1" : : "v"(dst), "v"(w) : "memory");
}
// ... write-through AND non-temporal (sc1 nt): for launches whose observations are far larger than the 256 MB last-level cache
// (Params::obs16 bit 1, set by pcgrl_create from the batch size).  Measured with A/B builds (profiles/r05_dev_traces.md): the
// 16x16 kernels gain 1.13 x at 604 MB per launch (zelda-turtle, 65 536 envs) and 1.46 x at 439 / 878 MB (binary-narrow, 131 072 /
// 262 144 envs: 0.51 -> 0.75 of the HBM peak), and LOSE 7 - 18 % below ~320 MB, where lines written through stay useful to the
// cache: hence the switch.
__device__ inline void store_obs16_nt(void *dst, uint4 v) {
  u32x4_t w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(dst), "v"(w) : "memory");
}

// The env's observation as `total` consecutive 16-byte chunks, lane r of the group taking chunks r, r + LPE, ...: every store
// instruction covers LPE * 16 contiguous bytes.  Chunk k = chunk k % CH of observation row k / CH, read from the LDS row
// row_ptr(k / CH).  Four chunks per trip: their LDS reads are issued together (the store is an asm statement with a memory
// clobber, so nothing moves across it by itself) and the row index comes from one fp32 multiply instead of an integer
// division (floor((k + 0.5) / CH) is exact in fp32 for k < 2^20).
template <int LPE, typename RowFn>
__device__ inline void stream_obs_chunks(const Grp<LPE> &g, uint8_t *base, int total, int CH, RowFn row_ptr) {
  const float inv_ch = 1.0f / (float)CH;
  int k = g.row;
  for (; k + 3 * LPE < total; k += 4 * LPE) {
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int kk = k + j * LPE;
      const int i = (int)(((float)kk + 0.5f) * inv_ch), q = kk - i * CH;
      v[j] = *(const uint4 *)(row_ptr(i) + q * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) store_obs16(base + (size_t)(k + j * LPE) * 16, v[j]);
  }
  for (; k < total; k += LPE) {
    const int i = (int)(((float)k + 0.5f) * inv_ch), q = k - i * CH;
    store_obs16(base + (size_t)k * 16, *(const uint4 *)(row_ptr(i) + q * 16));
  }
}

// Observation rows that are not a multiple of 16 bytes (e.g. the reference's zelda_small task: 22 pixels x 9 channels =
// 198 bytes): the env's observation is streamed as one byte string of n_rows * RB bytes, byte k taken from the LDS row
// row_ptr(k / RB) at offset k % RB -- dwords when the env size allows aligned ones, bytes otherwise.  Plain stores.
template <int LPE, typename RowFn>
__device__ inline void stream_obs_bytes(const Grp<LPE> &g, uint8_t *base, int n_rows, int RB, RowFn row_ptr) {
  const int T = n_rows * RB;
  const float inv = 1.0f / (float)RB;  // floor((k + 0.5) / RB) is exact in fp32 for k < 2^20
  if ((T & 3) == 0) {
    for (int k4 = g.row; k4 < (T >> 2); k4 += LPE) {
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = 4 * k4 + j;
        const int i = (int)(((float)k + 0.5f) * inv);
        v |= (uint32_t)row_ptr(i)[k - i * RB] << (8 * j);
      }
      *(uint32_t *)(base + 4 * (size_t)k4) = v;
    }
  } else {
    for (int k = g.row; k < T; k += LPE) {
      const int i = (int)(((float)k + 0.5f) * inv);
      base[k] = row_ptr(i)[k - i * RB];
    }
  }
}

// 16-byte chunk q of the all-out-of-bounds row pattern (byte k is 1 iff k % C == 0)
template <int C>
__device__ inline uint4 oob_chunk(int q) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int k = q * 16 + i * 4 + j;
      v |= (uint32_t)((k % C) == 0) << (8 * j);
    }
    w[i] = v;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int C>
__device__ inline uint4 oob_chunk_rt(int q) {
  switch (q % C) {
    case 0: return oob_chunk<C>(0);
    case 1: return oob_chunk<C>(1);
    case 2: return oob_chunk<C>(2);
    case 3: return oob_chunk<C>(3);
    case 4: return oob_chunk<C>(4);
    case 5: return oob_chunk<C>(5);
    case 6: return oob_chunk<C>(6);
    case 7: return oob_chunk<C>(7);
    default: return oob_chunk<C>(8);
  }
}

// ---- observation chunks from tile codes (general kernels, cropped window; Params::obs_codes) ------------------------------
// The one-hot rows of encode_obs cost OW * C bytes of LDS per map row (38 KB per workgroup at 32 x 32 zelda, 26 KB at
// 64 x 64 binary), which leaves a CU with 4-6 workgroups and a launch with two or three rounds of them.  Here a map row is
// kept as ONE BYTE PER CELL (Cropped's integer map: 1 + tile, 0 = outside the map; wrappers.py:407-437) with 8 zero bytes
// in front and 12 behind, and every 16-byte chunk of the observation (wrappers.py:232-257: byte = [code == channel]) is computed by
// the lane that stores it: the 8 codes its bytes can come from (three aligned dwords + two v_alignbyte), then per output
// dword one v_perm_b32 (which code each byte looks at), one xor with the channel each byte stands for, and a zero-byte
// test (x + 0x7f7f7f7f has bit 7 of a byte clear iff the byte was 0; codes and channels are < 16).  Which code / channel
// belongs to a byte depends only on the chunk's phase (16 q) mod C: a 48-byte entry per phase in LDS (selectors,
// channels, the all-out-of-bounds chunk).  LDS per env: H * (W + 20) bytes -- 2 KB instead of 19 at 32 x 32 zelda.
__device__ inline uint32_t onehot_bytes(uint32_t x) {  // per byte: 1 if the byte of x is 0, else 0 (bytes < 0x80)
  return (~(x + 0x7F7F7F7Fu) >> 7) & 0x01010101u;
}
// bits x0 .. x0+3 of a row mask (x0 may be negative or beyond the mask: those bits are 0)
template <typename M>
__device__ inline uint32_t nibble_at(M v, int x0) {
  constexpr int BITS = (int)(8 * sizeof(M));
  if (x0 >= BITS || x0 <= -4) return 0u;
  return (uint32_t)(x0 >= 0 ? (v >> x0) : (v << (-x0))) & 0xFu;
}
__device__ inline uint32_t spread4(uint32_t n) {  // bit i of a nibble -> bit 0 of byte i
  return (n * 0x204081u) & 0x01010101u;  // (n < 16: the compiler picks v_mul_u32_u24)
}

template <int PROB, int LPE, typename M>
__device__ inline void encode_obs_codes(const Grp<LPE> &g, const Params &p, int env, bool active, const M *b, const int *pos,
                                        uint8_t *lds, uint8_t *obs_base) {
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB, C = NT + 1, EPW = 64 / LPE;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1], OH = p.cfg.obs_window[0], OW = p.cfg.obs_window[1];
  const int CH = p.obs_chunks, RS = p.obs_codes;
  constexpr int PAD = 8;  // zero bytes in front of a code row (a chunk reads 8 codes from window column j0 on)
  uint8_t *rows = lds + (size_t)(g.lane / LPE) * H * RS;  // this env's code rows
  uint8_t *zero_row = lds + (size_t)EPW * H * RS;         // what a window row above / below the map reads
  uint32_t *table = (uint32_t *)(zero_row + RS);          // [C][12]: byte selectors, channels, out-of-bounds chunk
  // ---- this lane's map row as codes: [8 zeros][1 + tile per cell][>= 12 zeros]
  if (g.row < H) {
    const M inmap = W >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << W) - M(1));
    uint8_t *row = rows + g.row * RS;
    for (int o = 0; o < RS; o += 16) {
      uint32_t w[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int x0 = o + 4 * t - PAD;  // map column of the dword's first byte
        uint32_t v = 0;
        if (x0 > -4 && x0 < W) {
          v = spread4(nibble_at<M>(inmap, x0));
#pragma unroll
          for (int k = 0; k < NB; k++) v += spread4(nibble_at<M>(b[k] & inmap, x0)) << k;
        }
        w[t] = v;
      }
      *(uint4 *)(row + o) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  for (int o = g.lane * 16; o < RS; o += 64 * 16) *(uint4 *)(zero_row + o) = make_uint4(0, 0, 0, 0);
  if (g.lane < C) {  // phase r0 = (first byte of the chunk) mod C
    const int r0 = g.lane;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      uint32_t sel = 0, ch = 0, oob = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int k = r0 + 4 * w + t, d = k / C, c = k - d * C;  // byte 4w+t: code d of the eight, channel c
        sel |= (uint32_t)d << (8 * t);
        ch |= (uint32_t)c << (8 * t);
        oob |= (uint32_t)(c == 0) << (8 * t);
      }
      table[r0 * 12 + w] = sel;
      table[r0 * 12 + 4 + w] = ch;
      table[r0 * 12 + 8 + w] = oob;
    }
  }
  // ---- the env's observation as consecutive 16-byte chunks, lane r of the group taking chunks r, r + LPE, ...
  const int top = pos[0] - OH / 2, left = pos[1] - OW / 2;
  const int total = active ? OH * CH : 0;
  uint8_t *base = obs_base + (size_t)env * OH * OW * C;
  const float inv_ch = 1.0f / (float)CH;  // floor((k + 0.5) / CH) is exact in fp32 for k < 2^20
  auto chunk = [&](int k) -> uint4 {
    const int i = (int)(((float)k + 0.5f) * inv_ch), q = k - i * CH;
    const int j0 = (q * 16) / C, r0 = q * 16 - j0 * C;  // first window column of the chunk, phase
    const int m = i + top;
    const bool valid = (unsigned)m < (unsigned)H;
    const uint32_t *t = table + r0 * 12;
    if (__ballot(valid) == 0) return *(const uint4 *)(t + 8);  // every lane is above / below the map
    // the chunk's 8 codes are map columns x0 .. x0+7 of row m: all zero (the zero row) when the row is above / below the
    // map or the columns lie wholly left / right of it
    const int x0 = j0 + left, xb = x0 + PAD;
    const bool inside = valid && (unsigned)(x0 + 7) < (unsigned)(W + 7);
    const uint32_t *src = (const uint32_t *)(inside ? rows + m * RS + (xb & ~3) : zero_row);
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)(xb & 3));
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)(xb & 3));
    const uint4 sel = *(const uint4 *)t, ch = *(const uint4 *)(t + 4);
    return make_uint4(onehot_bytes(__builtin_amdgcn_perm(hi, lo, sel.x) ^ ch.x), onehot_bytes(__builtin_amdgcn_perm(hi, lo, sel.y) ^ ch.y),
                      onehot_bytes(__builtin_amdgcn_perm(hi, lo, sel.z) ^ ch.z), onehot_bytes(__builtin_amdgcn_perm(hi, lo, sel.w) ^ ch.w));
  };
  int k = g.row;
  for (; k + LPE < total; k += 2 * LPE) {  // two chunks per trip: their LDS reads are in flight together
    const uint4 v0 = chunk(k), v1 = chunk(k + LPE);
    store_obs16(base + (size_t)k * 16, v0);
    store_obs16(base + (size_t)(k + LPE) * 16, v1);
  }
  if (k < total) store_obs16(base + (size_t)k * 16, chunk(k));
}

#ifndef PCGRL_OBS_ROT
#define PCGRL_OBS_ROT 1  // (development: 0 = the byte scatter of rounds 1-5, every env of a wave at the same cell: A/B builds)
#endif
// FAST: map 16x16 with a 32x32 window (the reference's default obs_window = 2 * map_shape): every map row is
// visible, every map cell lands inside the window, each lane writes exactly one map row and one all-OOB row,
// and every loop bound is a compile-time constant.
template <int PROB, int LPE, bool FAST, typename M, bool ROT = true>
__device__ inline void encode_obs(const Grp<LPE> &g, const Params &p, int env, bool active, const M *b, const int *pos,
                                  uint8_t *lds, uint8_t *obs_base = nullptr) {
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  if (p.obs == nullptr) return;
  if (obs_base == nullptr) obs_base = p.obs;
  if (p.cfg.representation == PCGRL_REP_WIDE) {
    // wrappers.py:502-526: plain one-hot of the map, (H, W, NT), no out-of-bounds channel
    const int row_bytes = W * NT, chunks = (row_bytes + 15) >> 4;
    uint8_t *row = lds + g.lane * lds_row_stride(row_bytes);
    for (int q = 0; q < chunks; q++) *(uint4 *)(row + q * 16) = make_uint4(0, 0, 0, 0);
    if (active && g.row < H) {
      if constexpr (FAST && ROT && PCGRL_OBS_ROT && PROB == PCGRL_PROB_BINARY) {  // (see the cropped encoder below; no problem with a wide FAST kernel qualifies today)
        const int rot = 4 * (g.lane >> 4);
#pragma unroll
        for (int x = 0; x < 16; x++) {
          const int xx = (x + rot) & 15;
          row[xx * NT + tile_at<NB, M>(b, xx)] = 1;
        }
      } else {
        for (int x = 0; x < W; x++) row[x * NT + tile_at<NB, M>(b, x)] = 1;
      }
    }
    if (row_bytes & 15) {  // rows of odd size: the group streams the env's observation as one byte string
      const int gb = g.gbase, stride = lds_row_stride(row_bytes);
      if (active)
        stream_obs_bytes(g, obs_base + (size_t)env * H * row_bytes, H, row_bytes,
                         [&](int i) -> const uint8_t * { return lds + (gb + i) * stride; });
      return;
    }
    if (active) {
      // the env's observation as consecutive 16-byte chunks, lane r of the group taking chunks r, r + LPE, ...: every
      // store instruction covers LPE * 16 contiguous bytes (whole cache lines; lane-per-row stores left every line to be
      // written piecewise by several instructions, which write-through stores turn into 2 x the HBM write traffic)
      uint8_t *base = obs_base + (size_t)env * H * row_bytes;
      const int stride = lds_row_stride(row_bytes), gb = g.gbase;
      stream_obs_chunks(g, base, H * chunks, chunks, [&](int i) -> const uint8_t * { return lds + (gb + i) * stride; });
    }
    return;
  }
  // wrappers.py:407-437 Cropped (map+1, zero pad, window of obs_window around pos) -> :232-257 one-hot with
  // C = NT+1 channels, channel 0 = out of bounds -> :140-150 channel-last image.
  //
  // Each lane builds the observation row of ITS map row in LDS (OOB pattern, then one 0/1 byte pair per map cell);
  // LDS row 64 holds the pure out-of-bounds row.  The group then streams its env's observation out as consecutive
  // 16-byte chunks (lane r takes chunks r, r+LPE, ...), i.e. 256 contiguous bytes per group per store instruction,
  // picking each chunk from the LDS row of the lane that owns that map row or from the OOB row.
  constexpr int C = NT + 1;
  if constexpr (!FAST) {
    if (p.obs_codes > 0) {  // big maps: chunks computed from tile codes, a fraction of the LDS (see encode_obs_codes)
      encode_obs_codes<PROB, LPE, M>(g, p, env, active, b, pos, lds, obs_base);
      return;
    }
  }
  constexpr int FW = 16, FOW = 32, FOH = 32;
  const int OH = FAST ? FOH : p.cfg.obs_window[0], OW = FAST ? FOW : p.cfg.obs_window[1];
  const int CW = FAST ? FW : W;
  const int CH = FAST ? FOW * C / 16 : p.obs_chunks;  // 16-byte chunks per observation row
  const int RB = OW * C, STRIDE = CH * 16 + 16;
  uint8_t *row = lds + g.lane * STRIDE;
  uint8_t *oob_row = lds + 64 * STRIDE;
  const int top = pos[0] - OH / 2;   // map row shown by obs row 0
  const int left = pos[1] - OW / 2;  // map column shown by obs column 0
  if constexpr (FAST) {
#pragma unroll
    for (int q = 0; q < FOW * C / 16; q++) {
      *(uint4 *)(row + q * 16) = oob_chunk<C>(q % C);
      if (g.lane < 1) *(uint4 *)(oob_row + q * 16) = oob_chunk<C>(q % C);
    }
    // Byte scatter, one cell per trip.  The LDS rows of the wave's 64 lanes lie STRIDE = 16 * odd bytes apart: the 16 lanes of
    // an env hit 16 different banks (every fourth), but lanes l, l + 16, l + 32, l + 48 -- the same row of the wave's four envs --
    // hit the SAME bank whenever the envs write the same cell (narrow: always; the scan position is shared): a 4-way conflict
    // on every byte (bank-conflict rate 0.48 in rounds 1-5).  So env g of the wave starts at cell 4 g: four cells are 4 C bytes
    // = C dwords on, and C is odd for every problem here (3, 9; wide: NT = 5), so the four envs land in the four different
    // residues mod 4 and the 64 lanes in 64 different banks.
    // (ROT = false: the rollout kernel -- there the sixteen lane-dependent offsets become loop invariants of the step loop and
    // cost it its second wave per SIMD: 226 -> 257 VGPRs, 3.19 -> 5.76 us per step, measured)
    // (binary only: zelda's envs rarely share a position -- turtle -- and its 80-VGPR kernel paid 16 B of scratch for nothing:
    // conflict rate 0.328 either way; sokoban-wide 0.684 -> 0.659, no time)
    const int rot = (ROT && PCGRL_OBS_ROT && PROB == PCGRL_PROB_BINARY) ? 4 * (g.lane >> 4) : 0;
#pragma unroll
    for (int x = 0; x < FW; x++) {
      const int xx = (x + rot) & (FW - 1);
      int o = (xx - left) * C;  // always inside the window when OW = 2 * W
      row[o] = 0;
      row[o + 1 + tile_at<NB, M>(b, xx)] = 1;
    }
  } else {
    for (int q = 0; q < CH; q++) {
      *(uint4 *)(row + q * 16) = oob_chunk_rt<C>(q);
      if (g.lane < 1) *(uint4 *)(oob_row + q * 16) = oob_chunk_rt<C>(q);
    }
    if (g.row < H)
      for (int x = 0; x < CW; x++) {
        int j = x - left;
        if (j >= 0 && j < OW) {
          row[j * C] = 0;
          row[j * C + 1 + tile_at<NB, M>(b, x)] = 1;
        }
      }
  }
  if (active) {
    uint8_t *base = obs_base + (size_t)env * OH * RB;
    const int total = OH * CH;
    if constexpr (FAST) {
      constexpr int FCH = FOW * C / 16, ITERS = FOH * FCH / LPE, BATCH = 4;
      // (BATCH LDS reads are issued before the first of their stores: the store is an asm statement with a memory clobber,
      // so without this every chunk pays one dependent LDS round trip)
#pragma unroll
      for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
        uint4 v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
          if (it0 + j < ITERS) {
            const int k = (it0 + j) * LPE + g.row;
            const int i = k / FCH, q = k - i * FCH;
            const int m = i + top;
            v[j] = *(const uint4 *)(((unsigned)m < (unsigned)H ? lds + (g.gbase + m) * STRIDE : oob_row) + q * 16);
          }
        }
        if (p.obs16 & 2) {  // (wave-uniform: a scalar branch)
#pragma unroll
          for (int j = 0; j < BATCH; j++)
            if (it0 + j < ITERS) store_obs16_nt(base + (size_t)((it0 + j) * LPE + g.row) * 16, v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < BATCH; j++)
            if (it0 + j < ITERS) store_obs16(base + (size_t)((it0 + j) * LPE + g.row) * 16, v[j]);
        }
      }
    } else if (RB & 15) {
      const int gb = g.gbase;
      stream_obs_bytes(g, base, OH, RB, [&](int i) -> const uint8_t * {
        const int m = i + top;
        return (unsigned)m < (unsigned)H ? lds + (gb + m) * STRIDE : oob_row;
      });
    } else {
      const int gb = g.gbase;
      stream_obs_chunks(g, base, total, CH, [&](int i) -> const uint8_t * {
        const int m = i + top;
        return (unsigned)m < (unsigned)H ? lds + (gb + m) * STRIDE : oob_row;
      });
    }
  }
}

// The general kernels on a 16x16 map with a 32x32 window (after pcgrl_update, or with an action patch) still take the
// compile-time encoder: same LDS rows, constant loop bounds (observe wave 4.7 -> 3.6 us at 4096 envs).
template <int PROB, int LPE, bool FAST, typename M, bool ROT = true>
__device__ inline void encode_obs_any(const Grp<LPE> &g, const Params &p, int env, bool active, const M *b, const int *pos,
                                      uint8_t *lds, uint8_t *obs_base = nullptr) {
  if constexpr (!FAST && LPE == 16 && sizeof(M) == 4) {
    if (p.obs16 & 1) {
      encode_obs<PROB, LPE, true, M, ROT>(g, p, env, active, b, pos, lds, obs_base);
      return;
    }
  }
  encode_obs<PROB, LPE, FAST, M, ROT>(g, p, env, active, b, pos, lds, obs_base);
}

// ------------------------------------------------------------------------------------------------ kernels
// words per map row in HBM: tile bit-planes (1 or 3) [+ fars, best for binary] + the pre-flood plane (see PREFLOOD)
constexpr int ROW_WORDS = 4;
// PREFLOOD (binary, narrow / turtle, 16x16 kernels).  The narrow representation visits the cells in a fixed order and the
// turtle only edits the cell it stands on, so the cell of the NEXT edit is known one step ahead.  The component that cell belongs to once it is passable -- the only thing the
// incremental update needs a flood fill for -- is computed by the observe wave of the PREVIOUS launch, after it has issued
// its observation stores and while the simulate wave is still searching, and handed over in plane 3 of the state (bit 31
// of every row word = valid).  This takes the flood fill of the slowest env off the critical path of the next launch.
// Every path that changes map or position without refreshing the plane clears it (then the flood runs in the kernel).
constexpr int PRE_PLANE = 3;
constexpr uint32_t PRE_VALID = 1u << 31;
template <int N, typename M, bool MAP16 = false>
__device__ inline void load_planes(const Params &p, int env, int row, bool ok, M *b) {
  const M *pl = (const M *)p.planes;
  const int H = MAP16 ? 16 : p.cfg.dims[0];
#pragma unroll
  for (int k = 0; k < N; k++) b[k] = ok ? pl[((size_t)env * ROW_WORDS + k) * H + row] : M(0);
}
template <int N, typename M, bool MAP16 = false>
__device__ inline void store_planes(const Params &p, int env, int row, bool ok, const M *b) {
  M *pl = (M *)p.planes;
  const int H = MAP16 ? 16 : p.cfg.dims[0];
  if (ok) {
#pragma unroll
    for (int k = 0; k < N; k++) pl[((size_t)env * ROW_WORDS + k) * H + row] = b[k];
  }
}

// reps/*.update(): returns change flag (uniform over the group); edits the owning lane's planes, updates pos/n_step
template <int PROB, int LPE, typename M, bool MAP16 = false>
__device__ inline bool rep_update(const Grp<LPE> &g, const Params &p, bool active, int action, M *b, int *pos,
                                  int &n_step, bool &bad_action) {
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB;
  const int H = MAP16 ? 16 : p.cfg.dims[0], W = MAP16 ? 16 : p.cfg.dims[1];  // MAP16: the 16x16 (FAST) kernels
  int r = pos[0], c = pos[1], tile = -1;
  switch (p.cfg.representation) {
    case PCGRL_REP_NARROW:  // narrow_rep.py:89-102
      bad_action = action < 0 || action >= NT;
      tile = action;
      break;
    case PCGRL_REP_TURTLE:  // turtle_rep.py:87-107
      bad_action = action < 0 || action >= NT + 4;
      if (action >= 4) tile = action - 4;
      break;
    default: {  // wrappers.py:304-323 ActionMap + wide_rep.py:40-45: row = x = (a / NT) % W, col = y = a / (W*NT)
      bad_action = action < 0 || action >= H * W * NT;
      tile = action % NT;  // NT is a compile-time constant
      const int cell = action / NT;
      c = (int)(((float)cell + 0.5f) * (1.0f / (float)W));
      r = cell - c * W;
      break;
    }
  }
  if (bad_action || !active) tile = -1;
  bool mine = tile >= 0 && g.row == r;
  bool ch = false;
  if (mine) {
    ch = tile_at<NB, M>(b, c) != tile;
    set_tile<NB, M>(b, c, tile);
  }
  bool change = g.gany(ch);
  if (active && !bad_action) {
    if (p.cfg.representation == PCGRL_REP_NARROW) {  // Q1: position advances with the pre-increment index
      // floor((x + 0.5) / d) in fp32 is exact for x < 2^20: avoids two ~40-instruction integer divisions
      const int q = (int)(((float)n_step + 0.5f) * (1.0f / (float)(H * W)));
      const int idx = n_step - q * (H * W);
      pos[0] = (int)(((float)idx + 0.5f) * (1.0f / (float)W));
      pos[1] = idx - pos[0] * W;
      n_step++;
    } else if (p.cfg.representation == PCGRL_REP_TURTLE) {
      if (action < 4) {  // _dirs = [(-1,0),(1,0),(0,-1),(0,1)] on (row, col), clamped
        int dr = action == 0 ? -1 : (action == 1 ? 1 : 0), dc = action == 2 ? -1 : (action == 3 ? 1 : 0);
        pos[0] = min(max(pos[0] + dr, 0), H - 1);
        pos[1] = min(max(pos[1] + dc, 0), W - 1);
      }
    } else {
      pos[0] = r;
      pos[1] = c;
    }
  }
  return change;
}

// rep.update() through wrap_rep's stack StaticTile(MultiAction(rep)) (reps/wrappers.py:720-727).
template <int PROB, int LPE, typename M>
__device__ inline bool rep_update_ext(const Grp<LPE> &g, const Params &p, int env, bool active, int action, M *b, int *pos,
                                      int &n_step, bool &bad_action, ExtRow<ProbTraits<PROB>::NB, M> &X,
                                      bool &map_changed, bool &multi, const int32_t *act_base = nullptr) {
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  M pre[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) pre[k] = b[k];
  const bool was_stale = (X.flags & 1u) != 0;
  // StaticTileRepresentation.update :349-366: old_state = _bordered_map before the inner update.  Its interior equals
  // _map except right after a reset with static walls (flags bit 0), when it still holds the map without the walls.
  M old[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) old[k] = (X.flags & 1u) ? X.stale[k] : b[k];
  bool change;
  if (p.cfg.act_window[0] > 0) {
    // MultiActionRepresentation.update :466-524: the action is a row-major act_window patch of tile ids centred on _pos
    // (inner pads floor / ceil((k-1)/2), :404-410); then _pos walks the row-major list of centres that keep the patch
    // inside the map (get_act_coords :441-463) with the pre-increment n_step (narrow_rep.py:137-138)
    const int ah = p.cfg.act_window[0], aw = p.cfg.act_window[1];
    const int l0 = (ah - 1) / 2, l1 = (aw - 1) / 2;
    const int a_row = g.row - (pos[0] - l0), left = pos[1] - l1;
    const bool mine = active && a_row >= 0 && a_row < ah;
    const int32_t *src = (act_base ? act_base : p.actions) + (size_t)env * p.n_act + (mine ? a_row * aw : 0);
    bool badl = false;
    if (mine)
      for (int j = 0; j < aw; j++) badl = badl || src[j] < 0 || src[j] >= NT;
    bad_action = g.gany(badl);
    bool ch = false;
    if (mine && !bad_action)
      for (int j = 0; j < aw; j++) {
        const int t = src[j];
        ch = ch || tile_at<NB, M>(b, left + j) != t;
        set_tile<NB, M>(b, left + j, t);
      }
    change = g.gany(ch);
    if (active && !bad_action) {
      const int nx = W - aw + 1, n = (H - ah + 1) * nx;
      const int k = n_step % n;
      pos[0] = l0 + k / nx;
      pos[1] = l1 + k % nx;
      n_step++;
    }
  } else {
    change = rep_update<PROB, LPE, M>(g, p, active, action, b, pos, n_step, bad_action);
  }
  if (p.cfg.static_tiles && active && !bad_action) {
    if (change) {  // :357-362 undo the builds on static cells; `change` stays true even if everything was undone
#pragma unroll
      for (int k = 0; k < NB; k++) b[k] = (b[k] & ~X.prot) | (old[k] & X.prot);
    }
    X.flags &= ~1u;  // every representation's update ends with _update_bordered_map()
  }
  // for the statistics: did the map really change (an undone build leaves it as it was), and in more than one cell?
  M diff = M(0);
#pragma unroll
  for (int k = 0; k < NB; k++) diff |= pre[k] ^ b[k];
  map_changed = g.gany(diff != 0);
  multi = p.cfg.act_window[0] > 0 || was_stale;
  return change;
}

// Observation with the static_builds plane (wrappers.py:452-461): C = NT + 2 channels.  The plane goes through the same
// Cropped wrapper as the map but has the BORDERED shape while pos and pad are the map's, so observation pixel (i, j)
// shows map[r][q] and static_tiles[r][q] with the same (r, q) = (pos - window/2 + (i, j)): the plane is shifted by one
// cell against the map channels, and rows H, H+1 / columns W, W+1 carry static bits where the map is out of bounds.
template <int NB, typename M>
__device__ inline void build_obs_row_static(uint8_t *row, const Params &p, int C, int r, int left, const M *b, M prot_above) {
  const int H = p.cfg.dims[0], W = p.cfg.dims[1], OW = p.cfg.obs_window[1];
  for (int q = 0; q < p.obs_chunks; q++) *(uint4 *)(row + q * 16) = make_uint4(0, 0, 0, 0);
  for (int j = 0; j < OW; j++) {
    const int q = left + j;
    uint8_t *px = row + j * C;
    if (r < H && q >= 0 && q < W)
      px[1 + tile_at<NB, M>(b, q)] = 1;
    else
      px[0] = 1;
    if (r <= H + 1 && q >= 0 && q <= W + 1) {
      const bool ring = r == 0 || r == H + 1 || q == 0 || q == W + 1;
      px[C - 1] = ring ? 1 : (uint8_t)((prot_above >> (ring ? 0 : q - 1)) & M(1));
    }
  }
}

// the same for two tile types: C = 4, so a pixel is one dword (byte 0 out of bounds, byte 1 + tile, byte 3 static) and
// the row is assembled in registers, four pixels per 16-byte LDS store
// pixels j0 .. j0+3 of bordered row r (one 16-byte chunk)
template <typename M>
__device__ inline uint4 obs_chunk_static4(const Params &p, int r, int left, int j0, M tiles, M prot_above) {
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  const bool maprow = r < H, srow = r <= H + 1, ring_row = r == 0 || r == H + 1;
  uint32_t w[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int q = left + j0 + t;
    const bool inmap = maprow && (unsigned)q < (unsigned)W;
    const uint32_t tile = (uint32_t)((tiles >> (inmap ? q : 0)) & M(1));
    uint32_t v = inmap ? (0x100u << (8 * tile)) : 1u;
    if (srow && (unsigned)q <= (unsigned)(W + 1)) {
      const bool ring = ring_row || q == 0 || q == W + 1;
      v |= (ring ? 1u : (uint32_t)((prot_above >> (ring ? 0 : q - 1)) & M(1))) << 24;
    }
    w[t] = v;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename M>
__device__ inline void build_obs_row_static4(uint8_t *row, const Params &p, int r, int left, M tiles, M prot_above) {
  const int OW = p.cfg.obs_window[1];
  for (int j0 = 0; j0 < OW; j0 += 4) *(uint4 *)(row + j0 * 4) = obs_chunk_static4<M>(p, r, left, j0, tiles, prot_above);
}

template <int PROB, int LPE, typename M>
__device__ inline void encode_obs_static(const Grp<LPE> &g, const Params &p, int env, bool active, const M *b, const int *pos,
                                         M prot, uint8_t *lds, uint8_t *obs_base = nullptr) {
  constexpr int NT = ProbTraits<PROB>::NT, NB = ProbTraits<PROB>::NB;
  if (p.obs == nullptr) return;
  const int H = p.cfg.dims[0], OH = p.cfg.obs_window[0], OW = p.cfg.obs_window[1];
  const int C = NT + 2, CH = p.obs_chunks, RB = OW * C, STRIDE = CH * 16 + 16;
  const int top = pos[0] - OH / 2, left = pos[1] - OW / 2;
  // LDS rows: 0..63 one per lane (bordered row index = map row index), 64 the all-out-of-bounds row,
  // 65 + 2*group + {0, 1}: bordered rows H and H+1 of the group's env
  uint8_t *oob_row = lds + 64 * STRIDE;
  uint8_t *xrows = lds + (65 + 2 * (g.lane / LPE)) * STRIDE;
  M above = g.from_above(prot);  // static_tiles row r interior = protection of map row r-1
  M last = prot;
  if constexpr (sizeof(M) == 8)
    last = (M)g.gbcast((uint32_t)prot, H - 1) | ((M)g.gbcast((uint32_t)((uint64_t)prot >> 32), H - 1) << 32);
  else
    last = (M)g.gbcast((uint32_t)prot, H - 1);
  M none[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) none[k] = M(0);
  if constexpr (NT == 2) {
    if (g.row < H) build_obs_row_static4<M>(lds + g.lane * STRIDE, p, g.row, left, b[0], above);
    // the two bordered rows below the map, chunk by chunk over the lanes of the group (2 * CH chunks: one or two per lane
    // instead of a whole row on two lanes), and the all-out-of-bounds row (every pixel = byte 0 set) over the wave's lanes
    for (int k = g.row; k < 2 * CH; k += LPE) {
      const int xr = k >= CH ? 1 : 0, j0 = (k - xr * CH) * 4;
      *(uint4 *)(xrows + xr * STRIDE + j0 * 4) = obs_chunk_static4<M>(p, H + xr, left, j0, M(0), last);
    }
    for (int k = g.lane; k < CH; k += 64) *(uint4 *)(oob_row + k * 16) = make_uint4(1u, 1u, 1u, 1u);
  } else {
    if (g.row < H) build_obs_row_static<NB, M>(lds + g.lane * STRIDE, p, C, g.row, left, b, above);
    if (g.row < 2) build_obs_row_static<NB, M>(xrows + g.row * STRIDE, p, C, H + g.row, left, none, last);
    if (g.lane == 0) build_obs_row_static<NB, M>(oob_row, p, C, H + 2, 0, none, M(0));
  }
  if (active) {
    uint8_t *base = (obs_base ? obs_base : p.obs) + (size_t)env * OH * RB;
    const int total = OH * CH;
    if (RB & 15) {
      const int gb = g.gbase;
      stream_obs_bytes(g, base, OH, RB, [&](int i) -> const uint8_t * {
        const int m = i + top;
        return (unsigned)m < (unsigned)H ? lds + (gb + m) * STRIDE : ((m == H || m == H + 1) ? xrows + (m - H) * STRIDE : oob_row);
      });
      return;
    }
    const int gb = g.gbase;
    stream_obs_chunks(g, base, total, CH, [&](int i) -> const uint8_t * {
      const int m = i + top;
      return (unsigned)m < (unsigned)H ? lds + (gb + m) * STRIDE : ((m == H || m == H + 1) ? xrows + (m - H) * STRIDE : oob_row);
    });
  }
}

// Statistics after a representation update (pcgrl_env.py:314-323), shared by step_kernel and rollout_kernel.
//   restat  the map of this lane's env really changed (group-uniform)
//   multi   several cells may have changed at once (representation wrappers); never set in the FAST kernels
// binary: incremental around the edited cell(s), b[1] / b[2] (fars / best) are maintained;
// zelda / sokoban: full refresh with the region count updated around the edited cell when it is a one-cell edit.
template <int PROB, int LPE, typename M, bool FAST, bool SKA = false>
__device__ inline void refresh_stats(const Grp<LPE> &g, const Params &p, int e, bool restat, bool multi, M tile0_old,
                                     const M *pre, M *b, M colmask, int32_t *st PHASE_ARG, bool have_pre = false,
                                     M preflood = M(0), bool *unfinished = nullptr) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS;
  const bool full = PROB != PCGRL_PROB_BINARY && restat;  // binary: always incremental around the edited cell(s)
  if (__ballot(full) != 0) {
    int32_t ns[NS];
    int regions_known = -1;
    if constexpr (PROB != PCGRL_PROB_BINARY) {
      // one-cell edit: the region count moves only around that cell
      constexpr int RI = PROB == PCGRL_PROB_ZELDA ? 4 : 3;
      const bool one = full && !multi;
      if (__ballot(one) != 0) {
        M x = M(0);
#pragma unroll
        for (int k = 0; k < NB; k++) x |= pre[k] ^ b[k];
        x = one ? (x & colmask) : M(0);
        const int r = regions_update(g, x, region_cells<PROB, M>(pre, colmask), region_cells<PROB, M>(b, colmask), st[RI]);
        regions_known = one ? r : -1;
      }
    }
    compute_stats<PROB, LPE, M, !FAST, SKA>(g, p, e, full, b, colmask, ns, regions_known, unfinished);
    if (full) {
#pragma unroll
      for (int k = 0; k < NS; k++) st[k] = ns[k];
    }
  }
  if constexpr (PROB == PCGRL_PROB_BINARY) {
    const bool inc = restat && !multi;
    if (__ballot(inc) != 0) {
      // incremental: only the component(s) touching the edited cell are re-swept
      const M x = inc ? (tile0_old ^ b[0]) & colmask : M(0);
      int reg = st[0], len = st[1];
      M fars = b[1], best = b[2];
      binary_stats_update(g, x, ~tile0_old & colmask, ~b[0] & colmask, reg, len, fars, best PHASE_PASS, have_pre, preflood);
      if (inc) {
        st[0] = reg;
        st[1] = len;
        b[1] = fars;
        b[2] = best;
      }
    }
    if constexpr (!FAST) {  // several cells at once: only with the representation wrappers
      const bool many = restat && multi;
      if (__ballot(many) != 0) {
        const M X = many ? (tile0_old ^ b[0]) & colmask : M(0);
        int reg = st[0], len = st[1];
        M fars = b[1], best = b[2];
        binary_stats_update_multi(g, X, ~tile0_old & colmask, ~b[0] & colmask, reg, len, fars, best);
        if (many) {
          st[0] = reg;
          st[1] = len;
          b[1] = fars;
          b[2] = best;
        }
      }
    }
  }
}

// Episode end (auto-reset): what RLlib's callbacks read at that point (rl/callbacks.py:91-117) is latched in the env
// record (pcgrl_get_last_episode).  One lane per env.
template <int NS>
__device__ inline void latch_episode(const Params &p, int e, EnvState *S, double ep_return, int ep_len, const int32_t *st) {
  (void)p;
  (void)e;
  S->last_ep_return = ep_return;
  S->last_ep_len = ep_len;
  S->n_episodes += 1;
#pragma unroll
  for (int k = 0; k < NS; k++) S->final_stats[k] = st[k];
}
// ... and added to the env's running totals (pcgrl_reduce_episodes).  Called at the END of the kernel by the lane that
// latched, from the latched values: at that point the searches are over and the adds cost no registers where it matters
// (the step kernel's occupancy at large batches hangs on a handful of VGPRs).
template <int NS>
__device__ inline void accumulate_episode(EnvState *S) {
  EpAcc *A = &S->acc;
  A->sum_return += S->last_ep_return;
  A->sum_len += S->last_ep_len;
  A->n += 1;
#pragma unroll
  for (int k = 0; k < NS; k++) A->sum_stats[k] += S->final_stats[k];
}

// One workgroup = two specialised wavefronts over the same 64/LPE envs:
//   wave 0 "simulate": action -> stats -> reward/done -> auto-reset -> state write-back
//   wave 1 "observe" : replays the (cheap) action / reset on its own registers and encodes the observation
// The two never exchange data: the observation depends only on the post-action grid and position, not on the
// statistics, so the BFS latency chain and the LDS/HBM-store chain overlap instead of adding up.
// PAIRS (simulate, observe) wave pairs share a workgroup: 2 for the binary 16x16 kernel (512 instead of 1 024
// workgroups to dispatch at 4096 envs: 6.65 -> 6.57 us per launch; 4 pairs are slower, 7.7 us; zelda, whose launch is
// bound by its observation stores, is faster with 1).
// SKA (sokoban): asynchronous stepping (pcgrl_step_ready) -- one env per workgroup (Params::spread), the solver to a budget
// on the env's own workspace, no helper waves; an env whose search is parked leaves the launch without committing anything
// but its pending flag (EnvState::flags) and reports itself busy in Params::ready.
template <int PROB, int LPE, typename M, bool FAST, bool CTRL, int PAIRS = 1, bool SKA = false>
#ifndef PCGRL_B64_WAVES
#define PCGRL_B64_WAVES 6  // waves per SIMD the binary 64-bit-mask step kernel is compiled for (development: A/B builds)
#endif
#ifndef PCGRL_SKA_WAVES
#define PCGRL_SKA_WAVES 2  // min waves per SIMD of the asynchronous sokoban 16x16 step kernel (169 VGPRs, no scratch; 3 and 4 -- 128 VGPRs + 188 B of scratch -- measure the same: 1.13-1.14 x 10^7 env-steps/s at budget 16)
#endif
#ifndef PCGRL_STEP_WAVES
#define PCGRL_STEP_WAVES 1  // minimum waves per SIMD the binary 16x16 step kernel is compiled for (register budget)
#endif
__global__ __launch_bounds__((PROB == PCGRL_PROB_SOKOBAN && !SKA) ? 512 : 128 * PAIRS,
                             (FAST && PROB == PCGRL_PROB_BINARY) ? PCGRL_STEP_WAVES : ((!FAST && PROB == PCGRL_PROB_BINARY && sizeof(M) == 8) ? PCGRL_B64_WAVES : ((FAST && PROB == PCGRL_PROB_ZELDA && !CTRL) ? 6 : ((FAST && PROB == PCGRL_PROB_SOKOBAN && !CTRL) ? (SKA ? PCGRL_SKA_WAVES : 4) : 1))))
void step_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS, EPW = 64 / LPE;
  constexpr int NW = NB + ProbTraits<PROB>::NAUX;  // tile planes + incremental-stats masks
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
  touch_kernarg(p);
  Grp<LPE> g;
  g.init();
  // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is wave-uniform; with it the role branches are scalar.
  // Not for binary: no gain there and a register allocation of 86 instead of 80 VGPRs, i.e. 5 instead of 6 waves per SIMD)
  const int wave = PROB == PCGRL_PROB_BINARY ? (int)(threadIdx.x >> 6) : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pair = wave >> 1;
  const bool observer = (wave & 1) != 0;  // wave-uniform
  uint8_t *lds = lds_all + (size_t)pair * p.lds_pair_bytes;
  // sokoban, launched with p.sk_helpers (while its solver is busy): waves 2.. are the solver's helpers
  const bool helped = PROB == PCGRL_PROB_SOKOBAN && !SKA && p.sk_helpers != 0;
  if constexpr (PROB == PCGRL_PROB_SOKOBAN && !SKA) {
    if (wave >= 2 * PAIRS) {
      if (wave == 2 * PAIRS) sokoban_helpers_init();
      __syncthreads();
      const int j = wave - 2 * PAIRS;
      if (j < p.sk_helpers)
        sokoban_helper(p, j + 1, (uint32_t *)(lds_all + (size_t)PAIRS * p.lds_pair_bytes + (size_t)j * SK_HELPER_LDS));
      else
        sokoban_expander(p, j - p.sk_helpers + 1);
      return;
    }
  }
  if (observer && p.obs == nullptr) {  // (no barrier below is reached by wave 0 in that case either)
    if (helped) __syncthreads();
    return;
  }
  SokoHelpersGuard helpers_guard{helped && !observer};
  // Small batches are bound by the simulate wave's dependent chain: it issues ahead of the observe wave that shares its
  // SIMD (binary 4096 envs 5.88 -> 5.73 us, sokoban-wide 6.14 -> 5.90).  From 16 384 envs on the launch is bound by
  // the observation stores and the same priority costs 3-4 %, so it depends on the batch.
  if (PROB != PCGRL_PROB_ZELDA && !observer && p.n_envs <= 8192) __builtin_amdgcn_s_setprio(3);  // (zelda: store-bound at any size)
  PHASE_DECL();
  TRACE_DECL();
  const int H = FAST ? 16 : p.cfg.dims[0], W = FAST ? 16 : p.cfg.dims[1];
  // p.spread (sokoban while its solver is busy): one env per wave pair instead of 64 / LPE, so that every env's search
  // has a wavefront of its own (a wave runs one search at a time)
  const int env = p.spread ? (int)(blockIdx.x * PAIRS + pair) : (int)((blockIdx.x * PAIRS + pair) * EPW + (g.lane / LPE));
  const bool active = env < p.n_envs && (!p.spread || g.lane < LPE);
  const bool rowok = active && g.row < H;
  const M colmask = rowok ? (W >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << W) - M(1))) : M(0);
  const int e = active ? env : 0;

  // (both roles issue the same loads -- the observe wave just never looks at the statistics masks: with role-dependent
  // loads the compiler waits for them at the join, before the state and action loads are even issued)
  M b[NW];
  load_planes<NW, M, FAST>(p, e, g.row, rowok, b);
  // representation wrappers (static tiles / action patch): run-time option of the general kernels only
  ExtRow<NB, M> X;
  bool ext = false;
  if constexpr (!FAST) {
    ext = p.ext != 0;
    if (ext) X.load(p, e, g.row, rowok);
  }
  // PREFLOOD hand-over plane (binary 16x16 kernels): written by the observe wave of the previous launch
  constexpr bool PRE = FAST && PROB == PCGRL_PROB_BINARY;
  M *pre_word = (M *)p.planes + ((size_t)e * ROW_WORDS + PRE_PLANE) * (FAST ? 16 : p.cfg.dims[0]) + g.row;
  M pre3 = M(0);
  if constexpr (PRE) {
    if (!observer && rowok) pre3 = *pre_word;
  }
  EnvState *S = &p.st[e];
  int pos[2] = {S->pos[0], S->pos[1]};
  int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
  int action = active ? p.actions[e] : 0;
  // asynchronous stepping: a pending step is re-played with the action it took; an env whose reset still waits for its
  // statistics plays no step at all
  bool pend_step = false, pend_stats = false;
  if constexpr (SKA) {
    const int fl = active ? S->flags : 0;
    pend_step = (fl & ENV_PENDING_STEP) != 0;
    pend_stats = (fl & ENV_PENDING_STATS) != 0;
    if (pend_step) action = S->pend_action;
  }
  const bool stepping = active && !pend_stats;
  // An auto-reset of this step replays the env's RNG streams in BOTH waves; the simulate wave stores the advanced
  // streams at the end of the launch, so the observe wave takes its copy before the barrier (like every other piece
  // of old state).  The copy waits in LDS (behind the pair's observation rows) so that it costs no registers meanwhile.
  uint64_t *rng_stash = (uint64_t *)(lds + p.lds_pair_bytes - 64 * EPW) + 8 * (g.lane / LPE);
  if (observer && p.auto_reset != 0 && (iteration + 1 > p.cfg.max_iterations || p.cfg.max_changes >= 0) && g.row < 8)
    rng_stash[g.row] = ((const uint64_t *)&p.rng[e])[g.row];  // rep[4], prob[4]
  // both waves have read the old state before wave 0 may overwrite it
  if (p.obs != nullptr || helped) __syncthreads();
  PHASE_MARK(0);  // loads + barrier
  const M tile0_old = b[0];
  M pre[NB];  // pre-update tile planes (zelda / sokoban: incremental region count)
#pragma unroll
  for (int k = 0; k < NB; k++) pre[k] = b[k];

  // envs/pcgrl_env.py:267-342
  bool bad = false;
  const bool upd_only = p.update_only != 0;  // evolution-driver pattern: rep.update() without PcgrlEnv.step()
  iteration += (upd_only || pend_stats) ? 0 : 1;
  bool change, map_changed, multi = false;
  if (ext) {
    change = rep_update_ext<PROB, LPE, M>(g, p, e, active, action, b, pos, n_step, bad, X, map_changed, multi);
  } else {
    change = rep_update<PROB, LPE, M, FAST>(g, p, SKA ? stepping : active, action, b, pos, n_step, bad);
    map_changed = change;
  }
  changes += (change && !upd_only) ? 1 : 0;
  bool done = !upd_only && !pend_stats && iteration > p.cfg.max_iterations;
  if (p.cfg.max_changes >= 0) done = done || (!upd_only && !pend_stats && changes > p.cfg.max_changes);
  const bool do_reset = active && done && p.auto_reset != 0;

  if (observer) {
    if (__ballot(do_reset) != 0) {
      Pcg obs_rp, obs_rr;
      obs_rr.load(rng_stash);
      obs_rp.load(rng_stash + 4);
      reset_from_rng<PROB, LPE, M>(g, p, e, do_reset, b, pos, /*commit=*/false, ext ? &X : nullptr, &obs_rp, &obs_rr);
    }
    if (ext && p.cfg.static_tiles)
      encode_obs_static<PROB, LPE, M>(g, p, e, active, b, pos, X.prot, lds);
    else
      encode_obs_any<PROB, LPE, FAST, M>(g, p, e, active, b, pos, lds);
    if constexpr (PRE) {
      // narrow: the next edit goes to `pos`; turtle: an edit can only happen at `pos` (moves edit nothing).  Flood the
      // component of that cell ahead of time.
      if (p.cfg.representation != PCGRL_REP_WIDE) {
        const M xbit = (rowok && g.row == pos[0]) ? (M(1) << pos[1]) : M(0);
        const M comp = flood(g, xbit, (~b[0] & colmask) | xbit);
        if (rowok) *pre_word = comp | (M)PRE_VALID;
      }
    }
    TRACE_PUT(2, _tr0);
    TRACE_PUT(3, TRACE_NOW());
    TRACE_DRAIN();
    TRACE_PUT(4, TRACE_NOW());
    return;
  }

  int flags = S->flags;
  double last_loss = S->last_loss, ep_return = S->ep_return;
  int32_t st[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) st[k] = S->stats[k];
  if (bad && g.row == 0 && (SKA ? stepping : active)) atomicOr(p.err, 1);
  PHASE_MARK(1);  // action + second state loads
  if constexpr (SKA) {
    // a reset (explicit or automatic) whose statistics wait for a parked search: the search goes on; the launch in which it
    // ends gives the env its statistics and loss base, and it takes the NEXT launch's action
    if (__ballot(pend_stats) != 0) {  // (one env per wave: wave-uniform)
      int32_t ns[NS];
      bool unfin = false;
      compute_stats<PROB, LPE, M, !FAST, true>(g, p, e, pend_stats, b, colmask, ns, -1, &unfin);
      if (pend_stats && g.row == 0) {
        if (!unfin) {
#pragma unroll
          for (int k = 0; k < NS; k++) S->stats[k] = ns[k];
          S->last_loss = get_loss<NS>(p, ns);
          S->flags = flags & ~ENV_PENDING_STATS;
        }
        if (p.ready) p.ready[e] = unfin ? (uint8_t)PCGRL_ENV_BUSY : (uint8_t)0;
      }
      return;
    }
  }
  if (upd_only) {  // grid / position only; the stats (and the binary fars / best masks) go stale: ENV_STATS_DIRTY
    if (change) store_planes<NB, M, FAST>(p, e, g.row, rowok, b);
    if (change && map_changed && active && g.row == 0) S->flags = flags | ENV_STATS_DIRTY;
    if constexpr (PRE) {
      if (p.obs == nullptr && rowok && pre3 != M(0)) *pre_word = M(0);  // no observe wave: the plane goes stale
    }
    if (ext) X.store(p, e, g.row, rowok, active && g.row == 0, false);
    if (active && g.row == 0) {
      S->pos[0] = pos[0];
      S->pos[1] = pos[1];
      S->n_step = n_step;
    }
    return;
  }
  // the statistics can only move if the map did (with static tiles a build may have been undone: change without edit)
  // After pcgrl_update the statistics are stale (ENV_STATS_DIRTY): from scratch.  The compile-time 16x16 kernels carry
  // no code for it (it costs them 4 % at large batches): while stale envs may exist the host launches the general kernel
  // (Params::no_fast), which also drops the PREFLOOD plane it does not maintain.
  // (stale: whenever the representation reports a change -- also a build that static tiles undid: the reference then calls
  // get_stats on the current map, which the pcgrl_update calls before have edited)
  const bool stale = !FAST && !SKA && (flags & ENV_STATS_DIRTY) != 0 && change;  // (pcgrl_update is refused in asynchronous mode)
  if constexpr (FAST) {
    // Which kernel runs is a HOST decision (Params::no_fast) that a captured HIP graph freezes: a graph captured before
    // pcgrl_update and replayed after it lands here with stale statistics.  Never silently: the launch raises error
    // bit 8 (pcgrl_poll_error -> PCGRL_ESTALE) instead of updating incrementally from stale masks.
    if ((flags & ENV_STATS_DIRTY) != 0 && change && map_changed && active && g.row == 0) atomicOr(p.err, 8);
  }
  if constexpr (PROB == PCGRL_PROB_BINARY && !FAST) {
    if (p.no_fast && rowok) *pre_word = M(0);
  }
  if (!FAST && __ballot(stale) != 0) {
    int32_t ns[NS];
    compute_stats<PROB, LPE, M>(g, p, e, stale && active, b, colmask, ns);
    if (stale) {
#pragma unroll
      for (int k = 0; k < NS; k++) st[k] = ns[k];
      flags &= ~ENV_STATS_DIRTY;
    }
  }
  bool unfin = false;
  refresh_stats<PROB, LPE, M, FAST, SKA>(g, p, e, change && map_changed && !stale, multi, tile0_old, pre, b, colmask, st PHASE_PASS,
                                         PRE && g.gany((pre3 & (M)PRE_VALID) != 0), pre3 & colmask, SKA ? &unfin : nullptr);
  if constexpr (SKA) {
    // the search of this step's level did not end within the launch's budget: it is parked (sokoban_solve_async) and NOTHING
    // of the step is committed but the action it took -- the next launch re-plays the step from the same state up to here
    if (__ballot(unfin) != 0) {  // (one env per wave: wave-uniform)
      if (active && g.row == 0) {
        if (!pend_step) {
          S->flags = flags | ENV_PENDING_STEP;
          S->pend_action = action;
        }
        if (p.ready) p.ready[e] = (uint8_t)PCGRL_ENV_BUSY;
      }
      return;
    }
    flags &= ~ENV_PENDING_STEP;
  }
  PHASE_MARK(2);  // whole stats refresh
  // control_wrappers.py:216-244
  // CTRL (controllable mode) is a compile-time variant so that the plain kernel carries none of its code
  EnvTargets<CTRL ? NS : 1> trg;
  double loss;
  if constexpr (!CTRL) {
    loss = get_loss<NS>(p, st);
  } else {
    trg.load(p, e, false);
    loss = trg.loss(p.cfg, st);
  }
  double rew = loss - last_loss;
  last_loss = loss;
  ep_return += rew;
  if (active && g.row == 0) {
    if (p.reward) p.reward[e] = (float)rew;
    if constexpr (CTRL) {
      if (p.reward64) p.reward64[e] = rew;
    }
    if (p.done) p.done[e] = done ? 1 : 0;
    if (p.stats_out) {
#pragma unroll
      for (int k = 0; k < NS; k++) p.stats_out[(size_t)e * NS + k] = st[k];
    }
  }
  if constexpr (PROB != PCGRL_PROB_BINARY) PHASE_MARK(3);  // (timing builds, non-binary: loss + outputs)
  if (__ballot(do_reset) != 0) {
    if (do_reset && g.row == 0) latch_episode<NS>(p, e, S, ep_return, iteration, st);
    reset_from_rng<PROB, LPE, M>(g, p, e, do_reset, b, pos, /*commit=*/true, ext ? &X : nullptr);
    int32_t ns[NS];
    bool unfin2 = false;
    compute_stats<PROB, LPE, M, !FAST, SKA>(g, p, e, do_reset, b, colmask, ns, -1, SKA ? &unfin2 : nullptr);
    if (do_reset) {
#pragma unroll
      for (int k = 0; k < NS; k++) st[k] = ns[k];
      iteration = 0;
      changes = 0;
      n_step = 0;
      // (asynchronous stepping: the new map's search is parked -- the step itself is complete and emitted, the env stays
      // busy until a later launch has the new episode's statistics and loss base)
      flags = (SKA && unfin2) ? ENV_PENDING_STATS : 0;
      ep_return = 0.0;
      if constexpr (!CTRL) {
        last_loss = get_loss<NS>(p, st);
      } else {
        trg.load(p, e, true);  // queued control targets take effect with the new episode
        last_loss = trg.loss(p.cfg, st);
      }
    }
  }
  if constexpr (PROB != PCGRL_PROB_BINARY) PHASE_MARK(4);  // (timing builds, non-binary: auto-reset block)
  // write back state
  if (change || do_reset) store_planes<NW, M, FAST>(p, e, g.row, rowok, b);  // (stale implies change)
  if constexpr (PRE) {
    if (p.obs == nullptr && rowok && pre3 != M(0)) *pre_word = M(0);  // no observe wave: the plane goes stale
  }
  if (ext) X.store(p, e, g.row, rowok, active && g.row == 0, do_reset);
  if (active && g.row == 0) {
    if constexpr (CTRL) {
      trg.write_ctrl_obs(p, e, st);
      trg.commit(p, e);
    }
    S->pos[0] = pos[0];
    S->pos[1] = pos[1];
    S->n_step = n_step;
    S->iteration = iteration;
    S->changes = changes;
    S->flags = flags;
    S->last_loss = last_loss;
    S->ep_return = ep_return;
#pragma unroll
    for (int k = 0; k < NS; k++) S->stats[k] = st[k];
    if (do_reset) accumulate_episode<NS>(S);
    if constexpr (SKA) {
      if (p.ready) p.ready[e] = (uint8_t)(PCGRL_ENV_EMITTED | ((flags & ENV_PENDING_STATS) ? PCGRL_ENV_BUSY : 0));
    }
  }
  PHASE_MARK(6);  // loss, outputs, write-back
  PHASE_FLUSH();
  TRACE_PUT(0, _tr0);
  TRACE_PUT(1, TRACE_NOW());
}

// Open-loop rollout: p.n_steps consecutive steps of every env in ONE launch (pcgrl_rollout).  Same two specialised waves
// as step_kernel; the whole env state, including the two RNG streams, stays in registers between steps, so there is no
// per-step kernel boundary, no per-step state traffic, and waves advance independently (a launch no longer waits for
// its slowest env at every step).  CTRL: controllable mode (per-env targets, float64 rewards); the general (non-FAST)
// kernels also run the representation wrappers (static tiles / action patches, Params::ext).
// ROLE (round 6): 0 = both roles in one workgroup (simulate wave + observe wave), as in rounds 2-5;
//   1 = simulate only, 2 = observe only -- the two roles of ONE pcgrl_rollout call as two kernels on two streams.  They never
//   exchange anything (the observe role replays actions and resets on its own copy of the RNG streams), so nothing ties them to
//   one workgroup; apart they get a register budget each (together: the maximum, 224 VGPRs = two waves per SIMD) and the
//   simulate role can give every env a wavefront of its own (Params::spread): a wave then advances at the pace of ITS env
//   instead of the slowest of four at every step.  The observe kernel reads the state as it was BEFORE the call from a
//   snapshot (the simulate kernel overwrites the live state when it ends): Params::planes / st / rng point there.
template <int PROB, int LPE, typename M, bool FAST, bool CTRL = false, int ROLE = 0>
__global__ __launch_bounds__(ROLE == 0 ? 128 : 64) void rollout_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS, EPW = 64 / LPE;
  constexpr int NW = NB + ProbTraits<PROB>::NAUX;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  Grp<LPE> g;
  g.init();
  const bool observer = ROLE == 2 ? true : (ROLE == 1 ? false : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0);  // wave-uniform
  if (observer && p.obs == nullptr) return;
  PHASE_DECL();  // (development builds: the shared helpers take the phase counters; nothing is flushed here)
  const int H = FAST ? 16 : p.cfg.dims[0], W = FAST ? 16 : p.cfg.dims[1];
  // p.spread (1 .. 64 / LPE): envs per wavefront.  Fewer than a wave holds leaves lanes idle but gives every env's chain of
  // searches a wave (nearly) of its own: a wave advances at the pace of the slowest of ITS envs at every step, and at the
  // BASELINE batch sizes the machine has wave slots to spare (4096 envs / 4 = 1024 wave pairs on 1024 SIMDs)
  const int epw = p.spread > 0 ? p.spread : EPW;
  const int env = blockIdx.x * epw + (g.lane / LPE);
  const bool active = env < p.n_envs && (g.lane / LPE) < epw;
  const bool rowok = active && g.row < H;
  const M colmask = rowok ? (W >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << W) - M(1))) : M(0);
  const int e = active ? env : 0;
  const size_t N = (size_t)p.n_envs;
  const int K = p.n_steps;

  // (both roles issue the same loads -- the observe wave just never looks at the statistics masks: with role-dependent
  // loads the compiler waits for them at the join, before the state and action loads are even issued)
  M b[NW];
  load_planes<NW, M, FAST>(p, e, g.row, rowok, b);
  ExtRow<NB, M> X;
  bool ext = false;
  if constexpr (!FAST) {
    ext = p.ext != 0;
    if (ext) X.load(p, e, g.row, rowok);
  }
  EnvState *S = &p.st[e];
  int pos[2] = {S->pos[0], S->pos[1]};
  int n_step = S->n_step, iteration = S->iteration, changes = S->changes;
  int flags = S->flags;
  double last_loss = S->last_loss, ep_return = S->ep_return;
  int32_t st[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) st[k] = S->stats[k];
  EnvTargets<CTRL ? NS : 1> trg;
  if constexpr (CTRL) {
    if (!observer) trg.load(p, e, false);
  }
  Pcg rp, rr;
  rp.load(p.rng[e].prob);
  rr.load(p.rng[e].rep);
  const size_t astride = N * (size_t)p.n_act;  // action entries per step
  int action = (active && p.n_act == 1) ? p.actions[e] : 0;
  if (ROLE == 0 && p.obs != nullptr) __syncthreads();  // both waves hold the old state before wave 0 may overwrite it
  if (PROB != PCGRL_PROB_ZELDA && !observer && p.n_envs <= 8192) __builtin_amdgcn_s_setprio(3);  // (as in step_kernel)
  bool any_change = false, any_reset = false, bad_any = false;

  for (int k = 0; k < K; k++) {
    // the next step's action is requested now and consumed one iteration later (action patches are read in place)
    const int next_action = (k + 1 < K && active && p.n_act == 1) ? p.actions[(size_t)(k + 1) * N + e] : 0;
    const M tile0_old = b[0];
    M pre[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) pre[i] = b[i];
    bool bad = false;
    iteration++;
    bool change, map_changed, multi = false;
    if (ext) {
      change = rep_update_ext<PROB, LPE, M>(g, p, e, active, action, b, pos, n_step, bad, X, map_changed, multi,
                                            p.actions + (size_t)k * astride);
    } else {
      change = rep_update<PROB, LPE, M, FAST>(g, p, active, action, b, pos, n_step, bad);
      map_changed = change;
    }
    changes += change ? 1 : 0;
    bool done = iteration > p.cfg.max_iterations;
    if (p.cfg.max_changes >= 0) done = done || changes > p.cfg.max_changes;
    const bool do_reset = active && done && p.auto_reset != 0;
    bad_any = bad_any || bad;

    if (observer) {
      if (__ballot(do_reset) != 0) reset_from_rng<PROB, LPE, M>(g, p, e, do_reset, b, pos, false, ext ? &X : nullptr, &rp, &rr);
      if (!p.obs_last_only || k == K - 1) {
        uint8_t *obs_k = p.obs + (p.obs_last_only ? (size_t)0 : (size_t)k * N * (size_t)p.obs_env_bytes);
        if (ext && p.cfg.static_tiles)
          encode_obs_static<PROB, LPE, M>(g, p, e, active, b, pos, X.prot, lds, obs_k);
        else
          encode_obs_any<PROB, LPE, FAST, M, false>(g, p, e, active, b, pos, lds, obs_k);
      }
    } else {
      const bool restat = change && map_changed;
      const bool stale = !FAST && (flags & ENV_STATS_DIRTY) != 0 && change;  // first changing step after pcgrl_update
      if constexpr (FAST) {  // (see step_kernel: a captured launch replayed after pcgrl_update)
        if ((flags & ENV_STATS_DIRTY) != 0 && restat && active && g.row == 0) atomicOr(p.err, 8);
      }
      if (!FAST && __ballot(stale) != 0) {
        int32_t ns[NS];
        compute_stats<PROB, LPE, M>(g, p, e, stale && active, b, colmask, ns);
        if (stale) {
#pragma unroll
          for (int i = 0; i < NS; i++) st[i] = ns[i];
          flags &= ~ENV_STATS_DIRTY;
        }
      }
      refresh_stats<PROB, LPE, M, FAST>(g, p, e, restat && !stale, multi, tile0_old, pre, b, colmask, st PHASE_PASS);
      double loss;
      if constexpr (!CTRL) loss = get_loss<NS>(p, st);
      else loss = trg.loss(p.cfg, st);
      const double rew = loss - last_loss;
      last_loss = loss;
      ep_return += rew;
      if (active && g.row == 0) {
        const size_t o = (size_t)k * N + (size_t)e;
        if (p.reward) p.reward[o] = (float)rew;
        if constexpr (CTRL) {
          if (p.reward64) p.reward64[o] = rew;
        }
        if (p.done) p.done[o] = done ? 1 : 0;
        if (p.stats_out) {
#pragma unroll
          for (int i = 0; i < NS; i++) p.stats_out[o * NS + i] = st[i];
        }
      }
      if (__ballot(do_reset) != 0) {
        if (do_reset && g.row == 0) {
          latch_episode<NS>(p, e, S, ep_return, iteration, st);
          accumulate_episode<NS>(S);
        }
        reset_from_rng<PROB, LPE, M>(g, p, e, do_reset, b, pos, true, ext ? &X : nullptr, &rp, &rr);
        int32_t ns[NS];
        compute_stats<PROB, LPE, M, !FAST>(g, p, e, do_reset, b, colmask, ns);
        if (do_reset) {
#pragma unroll
          for (int i = 0; i < NS; i++) st[i] = ns[i];
          flags = 0;
          ep_return = 0.0;
          if constexpr (!CTRL) {
            last_loss = get_loss<NS>(p, st);
          } else {
            trg.load(p, e, true);  // queued control targets take effect with the new episode ...
            last_loss = trg.loss(p.cfg, st);
            if (g.row == 0) trg.commit(p, e);  // ... and are the active ones from here on
          }
        }
      }
      any_change = any_change || change;
      any_reset = any_reset || do_reset;
    }
    if (do_reset) {  // both waves restart their replay of the representation from the new episode
      iteration = 0;
      changes = 0;
      n_step = 0;
    }
    action = next_action;
  }
  if (observer) return;
  if (bad_any && g.row == 0 && active) atomicOr(p.err, 1);
  if (any_change || any_reset) store_planes<NW, M, FAST>(p, e, g.row, rowok, b);
  if constexpr (PROB == PCGRL_PROB_BINARY) {  // PREFLOOD plane: stale after a rollout
    if (rowok) ((M *)p.planes)[((size_t)e * ROW_WORDS + PRE_PLANE) * H + g.row] = M(0);
  }
  if (ext) X.store(p, e, g.row, rowok, active && g.row == 0, any_reset);
  if (active && g.row == 0) {
    if constexpr (CTRL) trg.write_ctrl_obs(p, e, st);  // control observation after the last step
    S->pos[0] = pos[0];
    S->pos[1] = pos[1];
    S->n_step = n_step;
    S->iteration = iteration;
    S->changes = changes;
    S->flags = flags;
    S->last_loss = last_loss;
    S->ep_return = ep_return;
#pragma unroll
    for (int i = 0; i < NS; i++) S->stats[i] = st[i];
    if (any_reset) {
      rr.store(p.rng[e].rep);
      rp.store(p.rng[e].prob);
    }
  }
}

// SKA (sokoban, asynchronous stepping): a level whose search does not end within the budget leaves the env ENV_PENDING_STATS
template <int PROB, int LPE, typename M, bool SKA = false>
__global__ __launch_bounds__(64) void reset_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS, EPW = 64 / LPE;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  Grp<LPE> g;
  g.init();
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  // (SKA: one env per wavefront -- a wave serves the searches of its envs one after the other, each to the full budget, so with
  // 64 / LPE envs per wave a reset launch of playable levels takes 64 / LPE budgets)
  const int env = SKA ? (int)blockIdx.x : blockIdx.x * EPW + (g.lane / LPE);
  const bool inb = env < p.n_envs && (!SKA || g.lane < LPE);
  const int e = inb ? env : 0;
  const bool active = inb && (p.mask == nullptr || p.mask[e] != 0);
  const bool rowok = active && g.row < H;
  const M colmask = rowok ? (W >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << W) - M(1))) : M(0);
  constexpr int NW = NB + ProbTraits<PROB>::NAUX;
  EnvState *S = &p.st[e];
  M b[NW];
  int pos[2] = {0, 0};
#pragma unroll
  for (int k = 0; k < NW; k++) b[k] = 0;
  ExtRow<NB, M> X;
  const bool ext = p.ext != 0;
  if (ext) X.load(p, e, g.row, rowok);
  if (p.refresh_only) {  // keep map, position and counters: only the statistics are recomputed from scratch
    load_planes<NB, M>(p, e, g.row, rowok, b);
  } else if (p.init_grids) {  // inject: bytes -> planes (envs/pcgrl_ctrl_env.py:12-14 set_map)
    if (rowok) {
      const uint8_t *src = p.init_grids + ((size_t)e * H + g.row) * W;
      for (int x = 0; x < W; x++) {
        int t = src[x];
#pragma unroll
        for (int k = 0; k < NB; k++) b[k] |= (M)((t >> k) & 1) << x;
      }
    }
    if (ext) {  // injected maps carry no static tiles (border ring only) and draw nothing
      X.prot = M(0);
      X.flags = 0;
      pos[0] = p.cfg.act_window[0] > 0 ? (p.cfg.act_window[0] - 1) / 2 : 0;
      pos[1] = p.cfg.act_window[0] > 0 ? (p.cfg.act_window[1] - 1) / 2 : 0;
    }
    // (with an action patch the position is a function of the step counter alone: init_pos is ignored)
    if (p.init_pos && p.cfg.representation != PCGRL_REP_WIDE && p.cfg.act_window[0] <= 0) {
      pos[0] = p.init_pos[(size_t)e * 3 + 0];
      pos[1] = p.init_pos[(size_t)e * 3 + 1];
    }
  } else {
    reset_from_rng<PROB, LPE, M>(g, p, e, active, b, pos, /*commit=*/true, ext ? &X : nullptr);
  }
  int32_t st[NS];
  bool unfin = false;
  compute_stats<PROB, LPE, M, true, SKA>(g, p, e, active, b, colmask, st, -1, SKA ? &unfin : nullptr);
  store_planes<NW, M>(p, e, g.row, rowok, b);
  if constexpr (PROB == PCGRL_PROB_BINARY) {  // PREFLOOD plane: stale after a reset (map and position changed)
    if (rowok && !p.refresh_only) ((M *)p.planes)[((size_t)e * ROW_WORDS + PRE_PLANE) * H + g.row] = M(0);
  }
  if (ext && !p.refresh_only) X.store(p, e, g.row, rowok, active && g.row == 0, true);
  if (p.refresh_only) {
    if (active && g.row == 0) {
      if (SKA && unfin) {  // the statistics arrive with a later pcgrl_step_ready launch
        S->flags = (S->flags & ~ENV_STATS_DIRTY) | ENV_PENDING_STATS;
        return;
      }
      EnvTargets<NS> trg;
      trg.load(p, e, false);
      S->last_loss = trg.loss(p.cfg, st);
      S->flags &= ~(ENV_STATS_DIRTY | ENV_PENDING_STATS);
#pragma unroll
      for (int k = 0; k < NS; k++) {
        S->stats[k] = st[k];
        if (p.stats_out) p.stats_out[(size_t)e * NS + k] = st[k];
      }
    }
    return;
  }
  if (active && g.row == 0) {
    S->pos[0] = pos[0];
    S->pos[1] = pos[1];
    S->pos[2] = 0;
    S->n_step = 0;
    S->iteration = 0;
    S->changes = 0;
    S->flags = (SKA && unfin) ? ENV_PENDING_STATS : 0;  // (a step that was pending is abandoned with the old map)
    S->ep_return = 0.0;
    if (p.set_state) {  // pcgrl_set_state: the map was injected above, the counters / return come from the caller
      if (p.in_counters) {
        S->iteration = p.in_counters[(size_t)e * 4 + 0];
        S->changes = p.in_counters[(size_t)e * 4 + 1];
        S->n_step = p.in_counters[(size_t)e * 4 + 2];
      }
      if (p.in_ep_return) S->ep_return = p.in_ep_return[e];
    }
    EnvTargets<NS> trg;
    trg.load(p, e, true);
    S->last_loss = trg.loss(p.cfg, st);
    trg.write_ctrl_obs(p, e, st);
    trg.commit(p, e);
#pragma unroll
    for (int k = 0; k < NS; k++) S->stats[k] = st[k];
  }
  (void)lds;
}

// control observation of the current state (pcgrl_ctrl_observe) and target queueing (pcgrl_queue_targets)
template <int NS>
__global__ __launch_bounds__(64) void ctrl_observe_kernel(Params p) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= p.n_envs) return;
  EnvTargets<NS> trg;
  trg.load(p, env, false);
  int32_t st[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) st[k] = p.st[env].stats[k];
  trg.write_ctrl_obs(p, env, st);
}

template <int PROB, int LPE, typename M, bool FAST>
__global__ __launch_bounds__(64) void observe_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, EPW = 64 / LPE;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  Grp<LPE> g;
  g.init();
  const int env = blockIdx.x * EPW + (g.lane / LPE);
  const bool active = env < p.n_envs;
  const int e = active ? env : 0;
  const bool rowok = active && g.row < p.cfg.dims[0];
  M b[NB];
  load_planes<NB, M>(p, e, g.row, rowok, b);
  int pos[2] = {p.st[e].pos[0], p.st[e].pos[1]};
  if constexpr (!FAST) {
    if (p.ext && p.cfg.static_tiles) {
      ExtRow<NB, M> X;
      X.load(p, e, g.row, rowok);
      encode_obs_static<PROB, LPE, M>(g, p, e, active, b, pos, X.prot, lds);
      return;
    }
  }
  encode_obs_any<PROB, LPE, FAST, M>(g, p, e, active, b, pos, lds);
}

// pcgrl_get_static: static mask in the reference's bordered layout, uint8 [N][(H+2)*(W+2)]
template <typename M>
__global__ __launch_bounds__(64) void get_static_kernel(Params p, int nb) {
  const int H = p.cfg.dims[0], W = p.cfg.dims[1], BW = W + 2;
  const int env = blockIdx.x;
  const M *xp = (const M *)p.xplanes + (size_t)env * (1 + nb) * H;
  uint8_t *dst = p.out_static + (size_t)env * (H + 2) * BW;
  for (int i = threadIdx.x; i < (H + 2) * BW; i += 64) {
    const int r = i / BW, c = i - r * BW;
    const bool ring = r == 0 || r == H + 1 || c == 0 || c == W + 1;
    dst[i] = ring ? 1 : (uint8_t)((xp[r - 1] >> (c - 1)) & M(1));
  }
}

template <int PROB, int LPE, typename M>
__global__ __launch_bounds__(64) void get_state_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS, EPW = 64 / LPE;
  Grp<LPE> g;
  g.init();
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  const int env = blockIdx.x * EPW + (g.lane / LPE);
  if (env >= p.n_envs) return;
  const EnvState *S = &p.st[env];
  if (p.out_grids && g.row < H) {
    M b[NB];
    load_planes<NB, M>(p, env, g.row, true, b);
    uint8_t *dst = p.out_grids + ((size_t)env * H + g.row) * W;
    for (int x = 0; x < W; x++) dst[x] = (uint8_t)tile_at<NB, M>(b, x);
  }
  if (g.row == 0) {
    if (p.out_pos) {
      p.out_pos[(size_t)env * 3 + 0] = S->pos[0];
      p.out_pos[(size_t)env * 3 + 1] = S->pos[1];
      p.out_pos[(size_t)env * 3 + 2] = 0;
    }
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
}

template <int PROB, int LPE>
__global__ __launch_bounds__(64) void last_episode_kernel(Params p) {
  constexpr int NS = ProbTraits<PROB>::NS;
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= p.n_envs) return;
  const EnvState *S = &p.st[env];
  if (p.out_ep_return) p.out_ep_return[env] = S->last_ep_return;
  if (p.out_ep_len) p.out_ep_len[env] = S->last_ep_len;
  if (p.out_n_episodes) p.out_n_episodes[env] = S->n_episodes;
  if (p.stats_out)
    for (int k = 0; k < NS; k++) p.stats_out[(size_t)env * NS + k] = S->final_stats[k];
}

// Problem.get_stats on caller-provided byte grids (no engine state)
template <int PROB, int LPE, typename M>
__global__ __launch_bounds__(PROB == PCGRL_PROB_SOKOBAN ? 448 : 64) void stats_for_grids_kernel(Params p) {
  constexpr int NB = ProbTraits<PROB>::NB, NS = ProbTraits<PROB>::NS, EPW = 64 / LPE;
  Grp<LPE> g;
  g.init();
  const bool helped = PROB == PCGRL_PROB_SOKOBAN && p.sk_helpers != 0;  // waves 1..3: the solver's heap waves, 4..6: their expanders
  if constexpr (PROB == PCGRL_PROB_SOKOBAN) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= 1) {
      if (wave == 1) sokoban_helpers_init();
      __syncthreads();
      extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
      if (wave - 1 < p.sk_helpers) sokoban_helper(p, wave, (uint32_t *)(lds_all + (size_t)(wave - 1) * SK_HELPER_LDS));
      else sokoban_expander(p, wave - p.sk_helpers);
      return;
    }
    if (helped) __syncthreads();
  }
  SokoHelpersGuard helpers_guard{helped};
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  // sokoban: one map per wavefront (its first lane group) -- a wave runs the solver for one of its maps at a time, so
  // spreading the maps over waves lets all of them search concurrently
  constexpr bool ONE = PROB == PCGRL_PROB_SOKOBAN;
  const int env = ONE ? (int)blockIdx.x : blockIdx.x * EPW + (g.lane / LPE);
  const bool active = env < p.n_envs && (!ONE || g.lane < LPE);
  const int e = active ? env : 0;
  const bool rowok = active && g.row < H;
  const M colmask = rowok ? (W >= (int)(8 * sizeof(M)) ? ~M(0) : ((M(1) << W) - M(1))) : M(0);
  M b[NB + ProbTraits<PROB>::NAUX];
#pragma unroll
  for (int k = 0; k < NB + ProbTraits<PROB>::NAUX; k++) b[k] = 0;
  if (rowok) {
    const uint8_t *src = p.init_grids + ((size_t)e * H + g.row) * W;
    for (int x = 0; x < W; x++) {
      int t = src[x];
#pragma unroll
      for (int k = 0; k < NB; k++) b[k] |= (M)((t >> k) & 1) << x;
    }
  }
  int32_t st[NS];
  compute_stats<PROB, LPE, M>(g, p, e, active, b, colmask, st);
  if (active && g.row == 0)
    for (int k = 0; k < NS; k++) p.stats_out[(size_t)e * NS + k] = st[k];
}

}  // namespace pcgrl
