// pcgrl_sokoban.h -- device-side Sokoban solver cascade (sokoban's get_stats when a level is "playable").
//
// Reference: envs/probs/sokoban/sokoban_prob.py:99-148 (_run_game: BFS, then A* with balance 1, 0.5, 0, each
// limited to `solver_power` iterations) and envs/probs/sokoban/sokoban/engine.py (Node :4-50, BFSAgent :56-74,
// AStarAgent :96-119, State :121-363).  It only runs when a map has exactly one player, as many crates as targets
// (> 0) and a single region (sokoban_prob.py:172-177) -- never under random actions, often under a trained policy.
//
// The search itself is sequential and order dependent (FIFO queue; CPython heapq sift order with
// Node.__lt__ = h + balance*depth; visited keyed on player + ORDERED crate list), so nodes are expanded one at a
// time -- but every per-node operation is spread over the wavefront that owns the env:
//   lane k holds crate k (and k + 64) of the node being expanded: "which crate stands at (x, y)", the win test,
//   the dead-cell count, the visited-set hash and key comparison, and the copy into a child node are one or two
//   ballots / coalesced accesses each instead of loops over the crate list;
//   lane t holds target t (and t + 64) for the greedy heuristic (engine.py:282-296): per crate one broadcast, one
//   distance per lane and one wave arg-min, in list order (the order decides the matching);
//   lane r holds level row r (then column r) while the dead-cell table is built (engine.py:203-246).
// The open list stores (2h + 2*balance*depth) << 16 | node, so CPython's sift comparisons read no node records.
// Node pool / visited table / open list live in an HBM workspace slot taken from a lock-protected pool sized to the
// batch; a wavefront holds at most one slot at a time (no lock cycles however many envs need solving at once).
//
// The four stages of the cascade are independent searches from the same root: only which results are USED depends on
// the order.  Kernels launched with Params::sk_helpers = 3 carry helper wavefronts that run the three A* stages
// speculatively (each in its own quarter of the slot) while the simulate wave runs the BFS stage; a stage is cancelled as
// soon as an earlier one has won (or the BFS stage has expanded the whole state space), so the launch waits for the
// longest single stage instead of the sum of four.  Each helped A* stage is a pair of wavefronts -- one keeps the open
// list, the other expands nodes one iteration ahead (sk_stage_heap / sokoban_expander).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "pcgrl_common.h"
#include "pcgrl_kernels2d.h"

namespace pcgrl {

// Development aid: -DPCGRL_SK_TIMING adds up the shader-clock cycles of the phases of a search iteration (pop, record
// loads + win test, visited set, children) per stage kind into p.err[64..] (tools/solver_phase.py).  Not shipped.
#ifdef PCGRL_SK_TIMING
#define SK_T_DECL() uint64_t _skt[5] = {0, 0, 0, 0, 0}; uint64_t _sk_prev = __builtin_readcyclecounter()
#define SK_T_MARK(i)                                 \
  do {                                               \
    const uint64_t _t = __builtin_readcyclecounter(); \
    _skt[i] += _t - _sk_prev;                        \
    _sk_prev = _t;                                   \
  } while (0)
#define SK_T_FLUSH(kind, iters)                                                                         \
  do {                                                                                                  \
    if (c.lane == 0 && c.dbg != nullptr) {                                                              \
      for (int _i = 0; _i < 5; _i++) atomicAdd(c.dbg + (kind) * 8 + _i, (unsigned long long)_skt[_i]);   \
      atomicAdd(c.dbg + (kind) * 8 + 5, (unsigned long long)(iters));                                   \
      atomicAdd(c.dbg + (kind) * 8 + 6, 1ull);                                                          \
    }                                                                                                   \
  } while (0)
#else
#define SK_T_DECL() \
  do {              \
  } while (0)
#define SK_T_MARK(i) \
  do {               \
  } while (0)
#define SK_T_FLUSH(kind, iters) \
  do {                          \
  } while (0)
#endif

constexpr int SK_MAXC = 128;     // crates (= targets) of a level the wave-cooperative solver keeps in two registers per lane, and
                                 // the crate-list area of a node in a stage workspace (a 16x16 map holds at most 127 pairs + player)
constexpr int SK_NH_HUGE = 8;    // levels with more pairs (maps of >= 258 cells can have them): 8 registers per lane, up to
constexpr int SK_MAXC_HUGE = 64 * SK_NH_HUGE;  // 512 pairs; their node crate lists spread over the crate areas of the slot's
                                 // four stage workspaces (sk_crate_ptr), the stages run one after the other on the simulate wave
constexpr int SK_MAXDIM = 64;    // bordered level side (H+2, W+2 <= 64: one lane / one mask bit per row and per column)
constexpr int SK_VCAP = 1 << 15; // visited table entries (>= 2 x iterations per stage), 16 bytes each, in groups of 8
constexpr int SK_VGROUPS = SK_VCAP / 8;  // a group = one 128-byte line: a probe reads it with one load (lane l < 8: entry l)
constexpr size_t SK_VIS_BYTES = 16 * (size_t)SK_VCAP;
constexpr int SK_MAX_POWER = 16000;  // cfg.solver_power accepted by pcgrl_create: node ids (<= 4 * power + 8) fit 16 bits
constexpr uint32_t SK_NOCRATE = 0xFFFFu;

struct SokoLevel {  // LDS, one per workgroup (one solve at a time per simulate wave)
  int32_t w, h, ncr, ntg;
  uint64_t solid[SK_MAXDIM + 2], dead[SK_MAXDIM + 2], tgt[SK_MAXDIM + 2];
  uint16_t root[SK_MAXC_HUGE];     // crates of the level in row-major order: x | y << 8 (engine.py:170-188)
  uint16_t target[SK_MAXC_HUGE];   // targets, same encoding
};
// The workspace pointers carry their address space (global / LDS): pointers read from memory are generic otherwise and
// every access becomes a flat_load that counts on both the vector-memory and the LDS counter.
#define SK_GLOBAL __attribute__((address_space(1)))
#define SK_LDS __attribute__((address_space(3)))
typedef uint32_t sk_u32x4 __attribute__((ext_vector_type(4)));
// node record in memory, 16 bytes: parent | depth (low 16) h (high 16) | px (bits 0-7) py (bits 8-15) | unused
struct SokoNode {
  int32_t parent, depth, h, px, py;
  __device__ inline sk_u32x4 pack() const {
    sk_u32x4 r;
    r.x = (uint32_t)parent;
    r.y = ((uint32_t)depth & 0xFFFFu) | ((uint32_t)h << 16);
    r.z = (uint32_t)px | ((uint32_t)py << 8);
    r.w = 0;
    return r;
  }
  __device__ static inline SokoNode unpack(sk_u32x4 r) {
    SokoNode n;
    n.parent = (int32_t)r.x;
    n.depth = (int32_t)(r.y & 0xFFFFu);
    n.h = (int32_t)(r.y >> 16);
    n.px = (int32_t)(r.z & 255u);
    n.py = (int32_t)((r.z >> 8) & 255u);
    return n;
  }
};
constexpr size_t SK_NODE_BYTES = 16;
// wave-uniform values the compiler cannot prove uniform (loop-carried counters, words read back from memory): one
// v_readfirstlane keeps the search's control flow on the scalar unit instead of exec-masked branches
__device__ inline int sk_u(int v) { return __builtin_amdgcn_readfirstlane(v); }

constexpr int SK_STAGES = 4;  // BFS, A* balance 1, 0.5, 0 (sokoban_prob.py:131-145)
struct SokoPool {  // lives in Params-reachable global memory
  int32_t n_slots, max_nodes;
  size_t stage_bytes;  // one stage's workspace; a slot holds SK_STAGES of them
  uint8_t *base;
  int32_t *locks;   // [n_slots] 0 = free
  uint32_t *epochs; // [n_slots][SK_STAGES]
  // asynchronous stepping (pcgrl_set_solver_budget): one stage workspace and one park record (SkPark) per env, else null
  uint8_t *async_ws;
  void *async_park;
};
// hand-over between the simulate wave and the helper waves of its workgroup (LDS)
struct SokoMail {
  int32_t seq;           // job number, bumped by the simulate wave once the level and the fields below are in place
  int32_t exit;          // the simulate wave is done with the launch
  int32_t slot, px, py, power;
  int32_t cancel_after;  // stages with a larger index stop
  int32_t done[SK_STAGES], won[SK_STAGES], h[SK_STAGES], depth[SK_STAGES];
};
// A* stage k of a helped cascade runs on TWO helper wavefronts (see sk_stage_heap): the heap wave posts the node it is
// about to pop, the expander wave answers with that node's children (or "won").
constexpr int SK_JOB_START = -2, SK_JOB_END = -3;
struct SokoPipe {
  int32_t jseq, rseq;     // job posted by the heap wave / answered by the expander
  int32_t cur;            // node to expand, or SK_JOB_START (new search: slot, px, py, b2) / SK_JOB_END (report the best node)
  int32_t slot, px, py, b2;
  // answer (two 16-byte words, read by the heap wave with one LDS round trip): won, n, h, depth -- won + the node's
  // (h, depth); START: h = the root's heuristic; END: the best node's -- and the open-list entries of the children in push order
  alignas(16) sk_u32x4 res;
  sk_u32x4 item;
};
struct SokoShared {
  SokoLevel level;
  SokoMail mail;
  SokoPipe pipe[SK_STAGES];
};
__device__ inline SokoShared &sk_shared() {
  __shared__ SokoShared s;
  return s;
}
__device__ inline int sk_ld(const int32_t *x) { return __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void sk_st(int32_t *x, int v) { __hip_atomic_store(x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct SokoCtx {
  SokoLevel *lv;
  sk_u32x4 SK_GLOBAL *nodes;  // [max_nodes] SokoNode::pack()
  uint16_t SK_GLOBAL *crates;  // [max_nodes][cstride]  x | y << 8 per crate (the area is sized for SK_MAXC per node)
  sk_u32x4 SK_GLOBAL *vis;     // [SK_VCAP]  x = (epoch << 17) | (node + 1), yzw = the state's key (SkKey)
  uint32_t SK_GLOBAL *q;       // [max_nodes] BFS queue (node) / A* heap (key << 16 | node)
  int32_t n_nodes, max_nodes, ncr;
  int32_t cstride;  // crate-list stride of this level in uint16: ncr rounded up to 4 (compact records stay in the L2)
  int32_t npr;      // levels with more than SK_MAXC pairs: node crate lists per stage workspace's crate area (else unused)
  size_t region_stride;  // ... and the distance between those areas (one stage workspace)
  uint32_t epoch;
  int lane;
  int stage;  // which quarter of the slot this context is bound to
  uint32_t SK_LDS *hl;  // helper waves: the first `hcap` entries of the A* heap live in LDS instead of c.q (else hcap = 0)
  int hcap;
  bool pool_full;
  unsigned long long *dbg;  // PCGRL_SK_TIMING builds
};

__device__ inline bool sk_bit(const uint64_t *rows, int x, int y) { return (rows[y] >> x) & 1ull; }

// minimum over the 64 lanes (result uniform): DPP butterfly inside the 16-lane rows, the four rows through SGPRs
__device__ inline uint32_t sk_wave_min(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}
// xor over the 64 lanes (result uniform)
__device__ inline uint32_t sk_wave_xor(uint32_t v) {
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// where node n's crate list lives
template <int NH>
__device__ inline uint16_t SK_GLOBAL *sk_crate_ptr(const SokoCtx &c, int n) {
  if constexpr (NH > 2) {  // list n % npr of the crate area of stage workspace n / npr (4 * npr >= max_nodes: cstride <= 512)
    const int r = n / c.npr, i = n - r * c.npr;
    return (uint16_t SK_GLOBAL *)((uint8_t SK_GLOBAL *)c.crates + (size_t)r * c.region_stride) + (size_t)i * c.cstride;
  } else {
    return c.crates + (size_t)n * c.cstride;
  }
}

// The crate list of one node, spread over the wave: lane k holds crate k in c0, (NH >= 2: levels with more than 64 crates)
// crate k + 64 in c1 and (NH > 2: more than 128, the whole-slot search) crate k + 64 j in cx[j - 2].
template <int NH>
struct SkCrates {
  static constexpr int NX = NH > 2 ? NH - 2 : 1;
  uint32_t c0, c1;  // x | y << 8, SK_NOCRATE beyond the list
  uint32_t cx[NX];
  __device__ inline void clear() {
    c0 = c1 = SK_NOCRATE;
#pragma unroll
    for (int j = 0; j < NX; j++) cx[j] = SK_NOCRATE;
  }
  __device__ inline void load(const SokoCtx &c, int n) {
    const uint16_t SK_GLOBAL *src = sk_crate_ptr<NH>(c, n);
    c0 = c.lane < c.ncr ? src[c.lane] : SK_NOCRATE;
    c1 = ((NH >= 2) && c.lane + 64 < c.ncr) ? src[c.lane + 64] : SK_NOCRATE;
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++) cx[j] = c.lane + 64 * (j + 2) < c.ncr ? src[c.lane + 64 * (j + 2)] : SK_NOCRATE;
    }
  }
  __device__ inline void load_level(const SokoCtx &c, const uint16_t *list) {  // the level's root list (LDS)
    c0 = c.lane < c.ncr ? list[c.lane] : SK_NOCRATE;
    c1 = ((NH >= 2) && c.lane + 64 < c.ncr) ? list[c.lane + 64] : SK_NOCRATE;
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++) cx[j] = c.lane + 64 * (j + 2) < c.ncr ? list[c.lane + 64 * (j + 2)] : SK_NOCRATE;
    }
  }
  __device__ inline void store(const SokoCtx &c, int n) const {
    uint16_t SK_GLOBAL *dst = sk_crate_ptr<NH>(c, n);
    if (c.lane < c.ncr) dst[c.lane] = (uint16_t)c0;
    if ((NH >= 2) && c.lane + 64 < c.ncr) dst[c.lane + 64] = (uint16_t)c1;
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++)
        if (c.lane + 64 * (j + 2) < c.ncr) dst[c.lane + 64 * (j + 2)] = (uint16_t)cx[j];
    }
  }
  // crate `moved` of the list takes the cell np
  __device__ inline void move(const SokoCtx &c, int moved, uint32_t np) {
    if (NH < 2 || moved < 64) c0 = c.lane == moved ? np : c0;
    else if (NH == 2 || moved < 128) c1 = c.lane == moved - 64 ? np : c1;
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++) cx[j] = (c.lane == moved - 64 * (j + 2)) ? np : cx[j];
    }
  }
  // crate k of the list, broadcast
  __device__ inline uint32_t get(int k) const {
    if (NH < 2 || k < 64) return (uint32_t)__builtin_amdgcn_readlane((int)c0, k & 63);
    if (NH == 2 || k < 128) return (uint32_t)__builtin_amdgcn_readlane((int)c1, (k - 64) & 63);
    uint32_t v = SK_NOCRATE;
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++)
        if ((k >> 6) == j + 2) v = (uint32_t)__builtin_amdgcn_readlane((int)cx[j], k & 63);
    }
    return v;
  }
  // index of the crate standing at (x, y), -1 if none (engine.py:263-267)
  __device__ inline int at(int x, int y) const {
    const uint32_t key = (uint32_t)x | ((uint32_t)y << 8);
    const uint64_t b0 = __ballot(c0 == key);
    if (NH < 2) return b0 ? __builtin_ctzll(b0) : -1;
    const uint64_t b1 = __ballot(c1 == key);
    if constexpr (NH > 2) {
      if (b0 == 0 && b1 == 0) {
#pragma unroll
        for (int j = 0; j < NX; j++) {
          const uint64_t bj = __ballot(cx[j] == key);
          if (bj) return 64 * (j + 2) + __builtin_ctzll(bj);
        }
        return -1;
      }
    }
    return b0 ? __builtin_ctzll(b0) : (b1 ? 64 + __builtin_ctzll(b1) : -1);
  }
  // number of crates whose cell has its bit set in `rows`
  __device__ inline int count_on(const uint64_t *rows) const {
    const bool h0 = c0 != SK_NOCRATE && sk_bit(rows, c0 & 255, c0 >> 8);
    int n = __popcll(__ballot(h0));
    if (NH >= 2) {
      const bool h1 = c1 != SK_NOCRATE && sk_bit(rows, c1 & 255, c1 >> 8);
      n += __popcll(__ballot(h1));
    }
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++) {
        const bool hj = cx[j] != SK_NOCRATE && sk_bit(rows, cx[j] & 255, cx[j] >> 8);
        n += __popcll(__ballot(hj));
      }
    }
    return n;
  }
  __device__ inline bool same(const SkCrates &o) const {
    bool d = c0 != o.c0 || ((NH >= 2) && c1 != o.c1);
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < NX; j++) d = d || cx[j] != o.cx[j];
    }
    return __ballot(d) == 0;
  }
};

template <int NH>
__device__ inline bool sk_free_cell(const SokoCtx &c, const SkCrates<NH> &cr, int x, int y) {  // checkMovableLocation :269-270
  if (x < 0 || y < 0 || x > c.lv->w - 1 || y > c.lv->h - 1) return false;
  return !sk_bit(c.lv->solid, x, y) && cr.at(x, y) < 0;
}

// engine.py:282-296 getHeuristic: crates in list order greedily take the nearest remaining target (first minimum in
// list order).  Lane t holds targets t and t + 64; per crate: broadcast its cell, one distance per lane, wave arg-min.
template <int NH>
__device__ __attribute__((always_inline)) inline int sk_heuristic(const SokoCtx &c, const SkCrates<NH> &cr) {
  const int nt = c.lv->ntg;
  if constexpr (NH > 2) {  // the same with target indices beyond 255: key = distance << 12 | target index
    uint32_t tg[NH];
    bool used[NH];
#pragma unroll
    for (int j = 0; j < NH; j++) {
      tg[j] = c.lane + 64 * j < nt ? c.lv->target[c.lane + 64 * j] : SK_NOCRATE;
      used[j] = tg[j] == SK_NOCRATE;
    }
    int distance = 0;
    for (int k = 0; k < c.ncr; k++) {
      const uint32_t ck = cr.get(k);
      const int cx = ck & 255, cy = ck >> 8;
      uint32_t key = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < NH; j++) {
        const uint32_t dj = used[j] ? 0xFFFFu : (uint32_t)(abs(cx - (int)(tg[j] & 255)) + abs(cy - (int)(tg[j] >> 8)));
        key = min(key, (dj << 12) | (uint32_t)(c.lane + 64 * j));
      }
      const uint32_t best = sk_wave_min(key);
      distance += (int)(best >> 12);
      const int t = best & 0xFFF;
#pragma unroll
      for (int j = 0; j < NH; j++) used[j] = used[j] || t == c.lane + 64 * j;
    }
    return distance;
  }
  const uint32_t t0 = c.lane < nt ? c.lv->target[c.lane] : SK_NOCRATE;
  const uint32_t t1 = ((NH >= 2) && c.lane + 64 < nt) ? c.lv->target[c.lane + 64] : SK_NOCRATE;
  bool u0 = t0 == SK_NOCRATE, u1 = t1 == SK_NOCRATE;  // "used" (absent targets never match)
  int distance = 0;
  for (int k = 0; k < c.ncr; k++) {
    const uint32_t ck = (NH < 2 || k < 64) ? (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, k & 63)
                                         : (uint32_t)__builtin_amdgcn_readlane((int)cr.c1, k - 64);
    const int cx = ck & 255, cy = ck >> 8;
    const uint32_t d0 = u0 ? 0xFFFFu : (uint32_t)(abs(cx - (int)(t0 & 255)) + abs(cy - (int)(t0 >> 8)));
    uint32_t key = (d0 << 8) | (uint32_t)c.lane;
    if (NH >= 2) {
      const uint32_t d1 = u1 ? 0xFFFFu : (uint32_t)(abs(cx - (int)(t1 & 255)) + abs(cy - (int)(t1 >> 8)));
      key = min(key, (d1 << 8) | (uint32_t)(c.lane + 64));
    }
    const uint32_t best = sk_wave_min(key);  // smallest distance, then smallest index: the first minimum
    distance += (int)(best >> 8);  // (ncr == ntg: an unused target always exists; the distance is < w + h)
    const int t = best & 255;
    u0 = u0 || t == c.lane;
    u1 = u1 || t == c.lane + 64;
  }
  return distance;
}

// engine.py:203-246 intializeDeadlocks.  A free, non-target cell in a corner of walls is dead; so is every cell strictly
// between two such corners of one row (column) when all cells between them are free, non-target and walled on at
// least one side -- i.e. inside a maximal run of such cells (corners are such cells themselves) everything between the
// first and the last corner is dead.  Lane r owns row r, then column r: shifts and scans in registers.
__device__ __attribute__((always_inline)) inline void sk_init_deadlocks(SokoCtx &c) {
  SokoLevel *lv = c.lv;
  const int w = lv->w, h = lv->h, r = c.lane;
  const bool inner = r >= 1 && r < h - 1;
  // between(E, C): for every maximal run of set bits of E, the bits from the run's first to its last C bit
  auto between = [](uint64_t E, uint64_t C, int n) -> uint64_t {
    uint64_t out = 0, span = 0;
    bool open = false;  // a corner was seen in the current run
    for (int x = 0; x < n; x++) {
      const uint64_t bit = 1ull << x;
      if (!(E & bit)) {
        open = false;
        span = 0;
        continue;
      }
      if (C & bit) {
        if (open) out |= span | bit;
        open = true;
        span = 0;
      }
      span |= open ? bit : 0ull;
    }
    return out;
  };
  // rows
  const uint64_t sol = inner ? lv->solid[r] : ~0ull, up = inner ? lv->solid[r - 1] : ~0ull, dn = inner ? lv->solid[r + 1] : ~0ull;
  const uint64_t tg = inner ? lv->tgt[r] : 0ull;
  const uint64_t lf = sol << 1, rt = sol >> 1;  // wall to the left / right of each cell
  const uint64_t colmask = ((1ull << (w - 1)) - 1ull) & ~1ull;  // inner columns 1 .. w-2
  const uint64_t corner = inner ? (~sol & ~tg & ((up & lf) | (up & rt) | (dn & lf) | (dn & rt)) & colmask) : 0ull;
  const uint64_t Eh = inner ? (~sol & ~tg & (up | dn) & colmask) : 0ull;
  uint64_t dead = corner | between(Eh, corner, w);
  // columns: lane x gathers column x of the masks (bit y = row y)
  uint64_t csol = 0, ctg = 0, ccor = 0, csl = 0, csr = 0;
  for (int y = 0; y < h; y++) {
    const uint64_t s = lv->solid[y], t = lv->tgt[y];
    const uint32_t clo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)corner, y), chi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(corner >> 32), y);
    const uint64_t cr = (uint64_t)clo | ((uint64_t)chi << 32);
    const int x = r;
    csol |= ((s >> x) & 1ull) << y;
    ctg |= ((t >> x) & 1ull) << y;
    ccor |= ((cr >> x) & 1ull) << y;
    csl |= (x >= 1 ? (s >> (x - 1)) & 1ull : 1ull) << y;  // wall in column x-1 / x+1 at this row
    csr |= (x < 63 ? (s >> (x + 1)) & 1ull : 0ull) << y;
  }
  const bool innerc = r >= 1 && r < w - 1;
  const uint64_t rowmask = ((1ull << (h - 1)) - 1ull) & ~1ull;
  const uint64_t Ev = innerc ? (~csol & ~ctg & (csl | csr) & rowmask) : 0ull;
  const uint64_t vspan = innerc ? between(Ev, ccor, h) : 0ull;
  // transpose the column spans back into row masks
  for (int y = 0; y < h; y++) {
    const uint64_t rowbits = __ballot((vspan >> y) & 1ull);
    if (r == y) dead |= rowbits;
  }
  if (r < h) lv->dead[r] = dead;
}

// State.getKey (engine.py:330-336: player position + the ORDERED crate list; targets never change) as 96 bits: exact for
// levels with up to five crates (every designed level), a 80-bit hash plus the position beyond (then a matching table
// entry is confirmed against the node's stored crate list).
struct SkKey {
  uint32_t k0, k1, k2;
  bool exact;
  __device__ inline int group() const {  // which line of the visited table
    uint32_t h = k0 * 0x9E3779B1u ^ k1 * 0x85EBCA77u ^ k2 * 0xC2B2AE3Du;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    return (int)((h ^ (h >> 13)) & (uint32_t)(SK_VGROUPS - 1));
  }
};
template <int NH>
__device__ __attribute__((always_inline)) inline SkKey sk_key(const SokoCtx &c, int px, int py, const SkCrates<NH> &cr) {
  SkKey k;
  const uint32_t xy = (uint32_t)px | ((uint32_t)py << 8);
  k.exact = c.ncr <= 5;
  if (k.exact) {  // (lanes beyond the list hold SK_NOCRATE)
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, 0), a1 = (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, 1);
    const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, 2), a3 = (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, 3);
    const uint32_t a4 = (uint32_t)__builtin_amdgcn_readlane((int)cr.c0, 4);
    k.k0 = xy | (a0 << 16);
    k.k1 = (a1 & 0xFFFFu) | (a2 << 16);
    k.k2 = (a3 & 0xFFFFu) | (a4 << 16);
  } else {  // two independent position-dependent mixes per crate, xor-combined over the wave
    uint32_t a = (cr.c0 + 0x9E3779B9u * (uint32_t)(c.lane + 1)) * 0x85EBCA6Bu;
    a ^= a >> 15;
    uint32_t b = (cr.c0 ^ (0x7F4A7C15u + 0x632BE5ABu * (uint32_t)c.lane)) * 0xC2B2AE35u;
    b ^= b >> 13;
    if (NH >= 2) {
      uint32_t a1 = (cr.c1 + 0x9E3779B9u * (uint32_t)(c.lane + 65)) * 0xC2B2AE35u;
      a ^= a1 ^ (a1 >> 13);
      uint32_t b1 = (cr.c1 ^ (0x1B873593u + 0x632BE5ABu * (uint32_t)(c.lane + 64))) * 0x85EBCA6Bu;
      b ^= b1 ^ (b1 >> 15);
    }
    if constexpr (NH > 2) {
#pragma unroll
      for (int j = 0; j < SkCrates<NH>::NX; j++) {
        uint32_t aj = (cr.cx[j] + 0x9E3779B9u * (uint32_t)(c.lane + 64 * (j + 2) + 1)) * 0xC2B2AE35u;
        a ^= aj ^ (aj >> 13);
        uint32_t bj = (cr.cx[j] ^ (0x1B873593u + 0x632BE5ABu * (uint32_t)(c.lane + 64 * (j + 2)))) * 0x85EBCA6Bu;
        b ^= bj ^ (bj >> 15);
      }
    }
    k.k1 = sk_wave_xor(a);
    k.k2 = sk_wave_xor(b);
    k.k0 = xy | ((k.k1 * 0x9E3779B1u ^ k.k2) & 0xFFFF0000u);
  }
  return k;
}

// returns true if the state (key; crates cr) was already in the visited set; inserts node n otherwise.  A probe reads one
// group of 8 entries with one load; a full group overflows into the next.
// `pre` / `have_pre`: the first group of the probe, already requested by the caller (the A* stage asks for it before its
// heap pop, so the memory round trip runs under the pop instead of after it).
template <int NH>
__device__ __attribute__((always_inline)) inline bool sk_visited_test_and_set(SokoCtx &c, int n, const SkKey &key, const SkCrates<NH> &cr,
                                                                               sk_u32x4 pre = sk_u32x4{0u, 0u, 0u, 0u}, bool have_pre = false) {
  int grp = key.group();
  while (true) {
    sk_u32x4 e = {0u, 0u, 0u, 0u};
    if (have_pre) e = pre;
    else if (c.lane < 8) e = c.vis[grp * 8 + c.lane];
    have_pre = false;
    const bool valid = c.lane < 8 && (e.x >> 17) == c.epoch && (e.x & 0x1FFFFu) != 0;
    const uint32_t empties = (uint32_t)__ballot(c.lane < 8 && !valid);
    const uint32_t matches = (uint32_t)__ballot(valid && e.y == key.k0 && e.z == key.k1 && e.w == key.k2);
    const int lim = empties ? __builtin_ctz(empties) : 8;  // entries are filled in order: nothing valid lies beyond
    uint32_t cand = matches & ((1u << lim) - 1u);
    if (key.exact) {
      if (cand) return true;
    } else {
      while (cand) {
        const int j = __builtin_ctz(cand);
        cand &= cand - 1u;
        const int m = (int)((uint32_t)__builtin_amdgcn_readlane((int)e.x, j) & 0x1FFFFu) - 1;
        SkCrates<NH> o;
        o.load(c, m);
        if (cr.same(o)) return true;
      }
    }
    if (empties) {
      if (c.lane == 0) {
        sk_u32x4 w;
        w.x = (c.epoch << 17) | (uint32_t)(n + 1);
        w.y = key.k0;
        w.z = key.k1;
        w.w = key.k2;
        c.vis[grp * 8 + lim] = w;
      }
      return false;
    }
    grp = (grp + 1) & (SK_VGROUPS - 1);
  }
}

// stage: b2 < 0 -> BFSAgent (engine.py:56-74); else AStarAgent with balance b2 / 2 (engine.py:96-119).  Uniform over
// the wave.  Node 0 = root, already filled by the caller.
// ---- the A* open list: CPython's heapq on c.q[0 .. tail), entries (2h + b2*depth) << 16 | node, compared by the key only
// (Node.__lt__ engine.py:49-50).  heapq's sift loops are chains of dependent reads; here a wavefront reads five levels of
// the tree per memory round trip (lane j holds the j-th descendant of the current position in level order) and walks them
// with scalar lane reads, remembers the path in its lanes and writes the whole path back with one store.
__device__ inline bool sk_key_lt(uint32_t a, uint32_t b) { return (a >> 16) < (b >> 16); }
// heap entry i: the top of the tree sits in LDS when the wave has some (no memory round trips there)
constexpr int SK_LDS_HEAP = 8192;  // entries per helper wave (32 KiB): a 10 000-iteration stage rarely grows beyond
#ifndef SK_ASYNC_HEAP
#define SK_ASYNC_HEAP 2048  // entries of the resumable solver's heap kept in LDS per workgroup (8 KiB; one env per workgroup)
#endif
// LDSONLY (the whole heap is in LDS, the common case): no global access is compiled in, so the sift never waits for the
// vector-memory counter -- i.e. for the previous iteration's record stores and the record requested ahead of time.
template <bool LDSONLY>
__device__ __attribute__((always_inline)) inline uint32_t sk_hq_load(const SokoCtx &c, int i, bool pred) {
  uint32_t v = 0;
  if (LDSONLY) {
    if (pred) v = c.hl[i];
  } else if (pred) {
    if (i < c.hcap) v = c.hl[i];
    else v = c.q[i];
  }
  return v;
}
template <bool LDSONLY>
__device__ __attribute__((always_inline)) inline void sk_hq_store(const SokoCtx &c, int i, uint32_t v, bool pred) {
  if (LDSONLY) {
    if (pred) c.hl[i] = v;
  } else if (pred) {
    if (i < c.hcap) c.hl[i] = v;
    else c.q[i] = v;
  }
}
// heapq.heappop (heapq.py:129-141 + _siftup :258-277 + _siftdown :205-218); tail > 0 on entry
// *new_top: the entry at the root afterwards (the next pop unless a smaller key is pushed first)
template <bool LDSONLY>
__device__ __attribute__((always_inline)) inline uint32_t sk_heappop_impl(SokoCtx &c, int &tail, uint32_t *new_top) {
  const int lane = c.lane;
  tail = sk_u(tail) - 1;
  int pos = 0, depth = 0;
  uint32_t last = 0, top = 0, root_pick = 0;
  bool leaf = false;
  // lane j stands for node j (level order, j = 0: `pos` itself) of the 6-level subtree below `pos` and reads that node's
  // two children itself: one memory wait per round; lane 63 is spare
  const int j1 = lane + 1, lvl = 31 - __builtin_clz((unsigned)j1), off = j1 - (1 << lvl);
  const int kids = lane < 31 ? 2 * lane + 1 : 0;  // lanes of my children
  for (int round = 0; !leaf; round++) {
    const int idx = ((pos + 1) << lvl) - 1 + off, il = 2 * idx + 1;
    const bool hasl = lane < 31 && il < tail, hasr = lane < 31 && il + 1 < tail;
    const uint32_t vl = sk_hq_load<LDSONLY>(c, il, hasl), vr = sk_hq_load<LDSONLY>(c, il + 1, hasr);
    if (round == 0) {  // the item to return (lane 0) and heap.pop(), the last element, re-inserted from the root (lane 63)
      const uint32_t v = sk_hq_load<LDSONLY>(c, lane == 63 ? tail : 0, lane == 0 || lane == 63);
      top = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
      last = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
      if (tail == 0) return last;  // it was the only element
    }
    // every inner node picks the child _siftup would move up: the right one unless left < right
    const bool right = hasr && !sk_key_lt(vl, vr);
    const int sel = hasl ? kids + (right ? 1 : 0) : -1;  // lane of that child, -1: no children
    const uint32_t pickv = right ? vr : vl;
    if (round == 0) root_pick = (uint32_t)__builtin_amdgcn_readlane((int)pickv, 0);
    // the walk follows `sel` (one scalar lane read per level) and collects the lanes of the nodes it leaves
    int loc = 0;
    uint64_t path = 0;
    for (int step = 0; step < 5; step++) {  // (the children of level 5 are not in this fetch)
      const int sl = __builtin_amdgcn_readlane(sel, loc);
      if (sl < 0) {
        leaf = true;
        break;
      }
      path |= 1ull << loc;
      loc = sl;
      depth++;
    }
    sk_hq_store<LDSONLY>(c, idx, pickv, ((path >> lane) & 1ull) != 0);  // heap[pos] = heap[childpos] along the path
    pos = __builtin_amdgcn_readlane(idx, loc);
  }
  // heap[pos] = newitem; _siftdown(heap, 0, pos): `last` climbs while it is smaller than its parent -- rarely more than
  // a level, it was a leaf
  while (pos > 0) {
    const int pp = (pos - 1) >> 1;
    const uint32_t parent = (uint32_t)sk_u((int)sk_hq_load<LDSONLY>(c, pp, true));
    if (!sk_key_lt(last, parent)) break;
    sk_hq_store<LDSONLY>(c, pos, parent, lane == 0);
    pos = pp;
  }
  sk_hq_store<LDSONLY>(c, pos, last, lane == 0);
  *new_top = (pos == 0) ? last : root_pick;
  return top;
}

// heapq.heappush (heapq.py:129-132 + _siftdown): the ancestors of the new leaf are read together (their positions follow
// from the leaf's alone); those larger than the item move down one level each.
template <bool LDSONLY>
__device__ __attribute__((always_inline)) inline void sk_heappush_impl(SokoCtx &c, int &tail, uint32_t item) {
  const int lane = c.lane, pos = sk_u(tail);
  tail = pos + 1;
  const int levels = 31 - __builtin_clz((unsigned)(pos + 1));  // ancestors of pos
  const int sh = lane < 30 ? lane : 30;
  const int anc = ((pos + 1) >> (sh + 1)) - 1;                 // lane j: ancestor j + 1 (ancestor 0 = pos itself)
  const bool has = lane < levels;
  const uint32_t v = sk_hq_load<LDSONLY>(c, anc, has);
  const uint64_t up = __ballot(has && sk_key_lt(item, v));
  const uint64_t stop = ~up;
  const int m = __builtin_ctzll(stop);  // (bit `levels` is always clear in `up`)
  const int mine = ((pos + 1) >> sh) - 1;  // ancestor `lane`
  sk_hq_store<LDSONLY>(c, mine, lane < m ? v : item, lane <= m);
}

__device__ __attribute__((always_inline)) inline uint32_t sk_heappop(SokoCtx &c, int &tail, uint32_t *new_top) {
  if (sk_u(tail) <= c.hcap) return sk_heappop_impl<true>(c, tail, new_top);
  return sk_heappop_impl<false>(c, tail, new_top);
}
__device__ __attribute__((always_inline)) inline void sk_heappush(SokoCtx &c, int &tail, uint32_t item) {
  if (sk_u(tail) < c.hcap) sk_heappush_impl<true>(c, tail, item);
  else sk_heappush_impl<false>(c, tail, item);
}

// Resumable form of a stage (asynchronous stepping, sokoban_solve_async): the search runs to an iteration budget per launch;
// everything it needs to continue -- queue / heap, node records, visited table -- already lives in the env's own workspace,
// so parking a search is saving these few wave-uniform words and resuming is loading them.  The pops, pushes and the visited
// set are those of the uninterrupted stage: the loop body does not know about the budget.
struct SkRes {
  bool resume;   // in: continue the parked stage instead of starting it
  bool parked;   // out: the stage stopped on the budget (state below saved)
  int budget;    // in / out: iteration units left in this launch (a BFS iteration costs 1, an A* iteration 2)
  int head, tail, n_nodes, best, best_h, best_depth, iters;
  uint32_t epoch;  // the workspace's visited-table epoch counter (kept in the park record instead of SokoPool::epochs)
};

// `cancel` (helper-wave mode): the stage gives up as soon as *cancel < my_stage (its result is not needed).
// RES: resumable form, `rs` carries the budget and the parked state (the synchronous kernels compile none of it).
template <int NH, bool RES = false>
__device__ __attribute__((always_inline)) inline bool sk_stage(SokoCtx &c, const SokoPool &pool, int slot, int b2, int max_iter, int &res_h, int &res_depth,
                                bool *exhausted = nullptr, const int32_t *cancel = nullptr, int my_stage = 0, SkRes *rs = nullptr) {
  uint32_t ep = 0;
  if constexpr (!RES) {
    uint32_t *epoch_word = &pool.epochs[slot * SK_STAGES + c.stage];
    if (c.lane == 0) ep = (atomicAdd(epoch_word, 1u) + 1u) & 0x7FFFu;
    ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
    if (ep == 0) {  // wrapped: start over with a clean table
      for (int i = c.lane; i < SK_VCAP; i += 64) c.vis[i] = sk_u32x4{0u, 0u, 0u, 0u};
      if (c.lane == 0) ep = (atomicAdd(epoch_word, 1u) + 1u) & 0x7FFFu;
      ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
    }
  } else {
    (void)pool;
    (void)slot;
    if (rs->resume) {
      ep = rs->epoch;
    } else {
      ep = (rs->epoch + 1u) & 0x7FFFu;
      if (ep == 0) {  // wrapped: start over with a clean table
        for (int i = c.lane; i < SK_VCAP; i += 64) c.vis[i] = sk_u32x4{0u, 0u, 0u, 0u};
        ep = 1u;
      }
      rs->epoch = ep;
    }
    rs->parked = false;
  }
  c.epoch = ep;
  c.n_nodes = 1;
  const int DX[4] = {-1, 1, 0, 0}, DY[4] = {0, 0, -1, 1};  // engine.py:3
  int head = 0, tail = 0, best = -1, best_h = 0, best_depth = 0, iters = 0;
  bool fresh = true;
  if constexpr (RES) {
    if (rs->resume) {
      fresh = false;
      head = rs->head;
      tail = rs->tail;
      best = rs->best;
      best_h = rs->best_h;
      best_depth = rs->best_depth;
      iters = rs->iters;
      c.n_nodes = rs->n_nodes;
    }
  }
  if constexpr (RES) {
    // the top of a parked A* heap comes back into LDS (the whole heap is in the workspace, see the park below)
    if (!fresh && b2 >= 0 && c.hcap > 0) {
      const int n = sk_u(tail) < c.hcap ? sk_u(tail) : c.hcap;
      for (int i = c.lane; i < n; i += 64) c.hl[i] = c.q[i];
    }
  }
  if (fresh) {
    const int h_root = sk_u(SokoNode::unpack(c.nodes[0]).h);
    if (b2 < 0) {
      if (c.lane == 0) c.q[0] = 0u;
    } else {
      sk_hq_store<false>(c, 0, (uint32_t)(2 * h_root) << 16, c.lane == 0);
    }
    tail = 1;
  }
  // BFS: the queue is read 64 entries at a time (lane l holds q[qbase + l]) and the node after the current one is
  // already in it, so its record is requested one iteration ahead
  int qbase = 0, qvalid = 0;
  uint32_t qv = 0;
  auto bfs_entry = [&](int i) -> int {  // q[i], head <= i < tail
    if (i - qbase >= qvalid) {
      qbase = i;
      qvalid = tail - i < 64 ? tail - i : 64;
      qv = c.lane < qvalid ? c.q[i + c.lane] : 0u;
    }
    return __builtin_amdgcn_readlane((int)qv, i - qbase);
  };
  // The next node's record is REQUESTED one iteration ahead and first looked at when that iteration begins: it stays packed
  // until then (unpacking it here -- a handful of bit operations -- made the compiler wait for the load on the spot, so the
  // "prefetch" was a round trip of its own in every iteration, rounds 4-6).
  // (Only the two words the expansion reads: with the whole record the unused `parent` register was reused right after the
  // load was issued -- the same wait by another route.)
  bool pre_valid = false;
  int pre_cur = 0;
  uint32_t pre_y = 0, pre_z = 0;  // words 1 and 2 of the record: depth | h << 16, px | py << 8
  auto pre_unpack = [&]() -> SokoNode {
    sk_u32x4 r = {0u, pre_y, pre_z, 0u};
    return SokoNode::unpack(r);
  };
  SkCrates<NH> pre_cr;
  pre_cr.clear();
  SK_T_DECL();
  while (iters < max_iter && head < tail) {
    if (cancel != nullptr && sk_u(sk_ld(cancel)) < my_stage) break;
    if constexpr (RES) {
      if (rs->budget <= 0) break;
      rs->budget -= b2 < 0 ? 1 : 2;
    }
    iters++;
    head = sk_u(head);
    tail = sk_u(tail);
    c.n_nodes = sk_u(c.n_nodes);
    SK_T_MARK(4);
    int cur;
    bool had_pre = pre_valid;
    uint32_t new_top = 0;
    SkKey pkey;
    sk_u32x4 pgrp = {0u, 0u, 0u, 0u};
    bool have_pgrp = false;
    pkey.k0 = pkey.k1 = pkey.k2 = 0u;
    pkey.exact = true;
    if (b2 < 0) {
      cur = had_pre ? pre_cur : bfs_entry(head);  // queue.pop(0)
      head++;
    } else {  // heapq.heappop
      if (had_pre) {  // the next node is (almost always) the root the previous pop left behind: its visited-set line is
        const SokoNode pn = pre_unpack();
        pkey = sk_key(c, sk_u(pn.px), sk_u(pn.py), pre_cr);  // requested now and arrives while the heap is sifted
        if (c.lane < 8) pgrp = c.vis[pkey.group() * 8 + c.lane];
        have_pgrp = true;
      }
      cur = (int)(sk_heappop(c, tail, &new_top) & 0xFFFFu);
      had_pre = had_pre && cur == pre_cur;  // (the record requested ahead of time: the root the previous pop left behind)
      have_pgrp = have_pgrp && had_pre;
    }
    SK_T_MARK(0);  // pop
    SokoNode nd;
    SkCrates<NH> cr;
    if (had_pre) {
      nd = pre_unpack();
      cr = pre_cr;
    } else {
      nd = SokoNode::unpack(c.nodes[cur]);
      cr.load(c, cur);
    }
    nd.depth = sk_u(nd.depth);
    nd.h = sk_u(nd.h);
    nd.px = sk_u(nd.px);
    nd.py = sk_u(nd.py);
    pre_valid = head < tail;  // BFS: the next queue entry; A*: the heap's new root
    if (pre_valid) {
      pre_cur = b2 < 0 ? bfs_entry(head) : (int)(new_top & 0xFFFFu);
      const uint32_t SK_GLOBAL *rec = (const uint32_t SK_GLOBAL *)(c.nodes + pre_cur);
      pre_cr.load(c, pre_cur);
      pre_y = rec[1];
      pre_z = rec[2];
    }
    const int px = nd.px, py = nd.py;
    if (c.lv->ntg == c.ncr && c.ncr > 0 && cr.count_on(c.lv->tgt) == c.ncr) {  // checkWin engine.py:272-280
      res_h = nd.h;
      res_depth = nd.depth;
      SK_T_FLUSH(b2 < 0 ? 0 : 1, iters);
      return true;
    }
    SK_T_MARK(1);  // record loads + win test
    const SkKey key = have_pgrp ? pkey : sk_key(c, px, py, cr);
    const bool seen = sk_visited_test_and_set(c, cur, key, cr, pgrp, have_pgrp);
    SK_T_MARK(2);  // visited set
    if (!seen) {
      if (best < 0 || nd.h < best_h || (nd.h == best_h && nd.depth < best_depth)) {  // engine.py:66-69
        best = cur;
        best_h = nd.h;
        best_depth = nd.depth;
      }
      const int n_dead = cr.count_on(c.lv->dead);
      // Node.getChildren engine.py:14-25 + State.update :298-328
      for (int d = 0; d < 4; d++) {
        const int nx = px + DX[d], ny = py + DY[d];
        if (nx < 0 || ny < 0 || nx > c.lv->w - 1 || ny > c.lv->h - 1 || sk_bit(c.lv->solid, nx, ny)) continue;
        const int moved = cr.at(nx, ny);
        SkCrates<NH> ch = cr;
        int h = nd.h;  // the heuristic depends on the crates only
        if (moved >= 0) {
          const int bx = nx + DX[d], by = ny + DY[d];
          if (!sk_free_cell(c, cr, bx, by)) continue;
          // engine.py:22-23 checkDeadlock over all crates of the child
          const int ndead = n_dead - (sk_bit(c.lv->dead, nx, ny) ? 1 : 0) + (sk_bit(c.lv->dead, bx, by) ? 1 : 0);
          if (ndead > 0) continue;
          const uint32_t np = (uint32_t)bx | ((uint32_t)by << 8);
          ch.move(c, moved, np);
          h = sk_heuristic(c, ch);
        }
        if (c.n_nodes >= c.max_nodes) {  // cannot happen (<= 1 + 4 * iterations nodes per stage); reported if it does
          c.pool_full = true;
          continue;
        }
        const int k = c.n_nodes++;
        ch.store(c, k);
        if (c.lane == 0) {
          SokoNode nn;
          nn.parent = cur;
          nn.depth = nd.depth + 1;
          nn.h = h;
          nn.px = nx;
          nn.py = ny;
          c.nodes[k] = nn.pack();
        }
        const uint32_t item = b2 < 0 ? (uint32_t)k : (((uint32_t)(2 * h + b2 * (nd.depth + 1)) << 16) | (uint32_t)k);
        if (b2 < 0) {
          if (c.lane == 0) c.q[tail] = item;
          tail++;
        } else {
          sk_heappush(c, tail, item);
        }
      }
      SK_T_MARK(3);  // children
    }
  }
  SK_T_FLUSH(b2 < 0 ? 0 : 1, iters);
  if constexpr (RES) {
    if (iters < max_iter && head < tail) {  // stopped on the budget: park
      if (b2 >= 0 && c.hcap > 0) {  // the heap's top leaves LDS: the workspace holds the whole open list between launches
        const int n = sk_u(tail) < c.hcap ? sk_u(tail) : c.hcap;
        for (int i = c.lane; i < n; i += 64) c.q[i] = c.hl[i];
      }
      rs->parked = true;
      rs->head = sk_u(head);
      rs->tail = sk_u(tail);
      rs->n_nodes = sk_u(c.n_nodes);
      rs->best = sk_u(best);
      rs->best_h = sk_u(best_h);
      rs->best_depth = sk_u(best_depth);
      rs->iters = sk_u(iters);
      return false;
    }
  }
  res_h = best_h;
  res_depth = best_depth;
  if (exhausted) *exhausted = head >= tail;  // the open list ran dry: every reachable state was expanded
  return false;
}

// bind the context to the stage workspace at `b` (of pool.stage_bytes bytes)
__device__ inline void sk_bind_at(SokoCtx &c, const SokoPool &pool, uint8_t *b, int stage) {
  c.stage = stage;
  c.max_nodes = pool.max_nodes;
  c.nodes = (sk_u32x4 SK_GLOBAL *)b;
  b += SK_NODE_BYTES * (size_t)c.max_nodes;
  c.crates = (uint16_t SK_GLOBAL *)b;
  c.region_stride = pool.stage_bytes;
  c.npr = c.cstride > 0 ? (int)(((size_t)c.max_nodes * SK_MAXC) / (size_t)c.cstride) : c.max_nodes;
  b += (size_t)c.max_nodes * SK_MAXC * sizeof(uint16_t);
  c.vis = (sk_u32x4 SK_GLOBAL *)b;
  b += SK_VIS_BYTES;
  c.q = (uint32_t SK_GLOBAL *)b;
}
// bind the context to stage workspace `stage` of `slot`
__device__ inline void sk_bind(SokoCtx &c, const SokoPool &pool, int slot, int stage) {
  uint8_t *b = pool.base + ((size_t)slot * SK_STAGES + stage) * pool.stage_bytes;
  c.stage = stage;
  c.max_nodes = pool.max_nodes;
  c.nodes = (sk_u32x4 SK_GLOBAL *)b;
  b += SK_NODE_BYTES * (size_t)c.max_nodes;
  c.crates = (uint16_t SK_GLOBAL *)b;
  c.region_stride = pool.stage_bytes;
  c.npr = c.cstride > 0 ? (int)(((size_t)c.max_nodes * SK_MAXC) / (size_t)c.cstride) : c.max_nodes;
  b += (size_t)c.max_nodes * SK_MAXC * sizeof(uint16_t);
  c.vis = (sk_u32x4 SK_GLOBAL *)b;
  b += SK_VIS_BYTES;
  c.q = (uint32_t SK_GLOBAL *)b;
}

// node 0 of the bound workspace = the level's root state (crates in c.lv->root)
template <int NH>
__device__ inline int sk_root(SokoCtx &c, int px, int py) {
  SkCrates<NH> root;
  root.load_level(c, c.lv->root);
  root.store(c, 0);
  const int h0 = sk_heuristic(c, root);
  if (c.lane == 0) {
    SokoNode n0;
    n0.parent = -1;
    n0.depth = 0;
    n0.h = h0;
    n0.px = px;
    n0.py = py;
    c.nodes[0] = n0.pack();
  }
  return h0;
}

// ---- an A* stage on two wavefronts.  An iteration is pop -> expand -> push the children, and the pop is a chain of
// dependent heap reads that touches no node record, while the expansion touches no heap entry.  The node the pop will
// return is the heap's root before the sift, so the heap wave posts it first and sifts while the expander wave loads
// the record, tests win / visited, builds the children and their open-list entries; the heap wave then pushes those
// entries in the same order.  Same pops, same pushes, same visited set as the one-wave stage -- an iteration costs the
// heap operations alone (tools/solver_phase.py: 3 600 + 1 200 of 7 400 cycles).
__device__ __attribute__((always_inline)) inline void sk_pipe_post(SokoPipe &pp, int lane, int seq, int cur) {
  if (lane == 0) {
    pp.cur = cur;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    sk_st(&pp.jseq, seq);
  }
}
__device__ __attribute__((always_inline)) inline void sk_pipe_wait(SokoPipe &pp, int seq) {
  while (__builtin_amdgcn_readfirstlane(sk_ld(&pp.rseq)) != seq) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// heap wave of stage k: c.hl / c.q hold the open list; `seq` is the pipe's job counter (kept by this wave).
// The node an iteration pops is known one iteration ahead: after the pop the heap's root is `new_top`, and a child takes
// its place only with a strictly smaller key (heapq's sift-up compares with <), the first such child first -- so the next
// job is posted BEFORE the children are pushed and the expander works on it during the pushes and the next sift.  It is
// posted only if that iteration is certain to run (the iteration limit is known; a cancelled stage's result is unused).
__device__ __attribute__((always_inline)) inline bool sk_stage_heap(SokoCtx &c, SokoPipe &pp, int &seq, int max_iter, int h_root, int &res_h,
                                                          int &res_depth, const int32_t *cancel, int my_stage) {
  int tail = 1, iters = 0;
  bool won = false;
  sk_hq_store<false>(c, 0, (uint32_t)(2 * h_root) << 16, c.lane == 0);
  bool live = sk_u(sk_ld(cancel)) >= my_stage;
  bool more = max_iter > 0 && live;
  if (more) sk_pipe_post(pp, c.lane, ++seq, 0);  // the root
  while (more) {
    iters++;
    tail = sk_u(tail);
    uint32_t new_top = 0;
    (void)sk_heappop(c, tail, &new_top);  // (returns the entry of the node posted for this iteration)
    sk_pipe_wait(pp, seq);
    const sk_u32x4 r = pp.res, it = pp.item;  // (one round trip for both)
    if (sk_u((int)r.x) != 0) {
      won = true;
      res_h = sk_u((int)r.z);
      res_depth = sk_u((int)r.w);
      break;
    }
    const int n = sk_u((int)r.y);
    const uint32_t items[4] = {(uint32_t)sk_u((int)it.x), (uint32_t)sk_u((int)it.y), (uint32_t)sk_u((int)it.z), (uint32_t)sk_u((int)it.w)};
    if ((iters & 15) == 0) live = sk_u(sk_ld(cancel)) >= my_stage;  // (a cancelled stage's result is unused: no hurry)
    more = iters < max_iter && (tail > 0 || n > 0) && live;
    if (more) {  // the root after the pushes below
      uint32_t next = tail > 0 ? new_top : items[0];
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (i < n && sk_key_lt(items[i], next)) next = items[i];
      sk_pipe_post(pp, c.lane, ++seq, (int)(next & 0xFFFFu));
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < n) sk_heappush(c, tail, items[i]);
  }
  sk_pipe_post(pp, c.lane, ++seq, SK_JOB_END);
  sk_pipe_wait(pp, seq);
  if (!won) {
    const sk_u32x4 r = pp.res;
    res_h = sk_u((int)r.z);
    res_depth = sk_u((int)r.w);
  }
  return won;
}

// expander wave: one node of the search bound to c (AStarAgent.getSolution's loop body without the heap, engine.py:104-119)
template <int NH>
__device__ __attribute__((always_inline)) inline void sk_expand(SokoCtx &c, SokoPipe &pp, int cur, int b2, int &best, int &best_h, int &best_depth) {
  const int DX[4] = {-1, 1, 0, 0}, DY[4] = {0, 0, -1, 1};  // engine.py:3
  SokoNode nd = SokoNode::unpack(c.nodes[cur]);
  SkCrates<NH> cr;
  cr.load(c, cur);
  nd.depth = sk_u(nd.depth);
  nd.h = sk_u(nd.h);
  nd.px = sk_u(nd.px);
  nd.py = sk_u(nd.py);
  const int px = nd.px, py = nd.py;
  if (c.lv->ntg == c.ncr && c.ncr > 0 && cr.count_on(c.lv->tgt) == c.ncr) {  // checkWin engine.py:272-280
    if (c.lane == 0) pp.res = sk_u32x4{1u, 0u, (uint32_t)nd.h, (uint32_t)nd.depth};
    return;
  }
  const SkKey key = sk_key(c, px, py, cr);
  const bool seen = sk_visited_test_and_set(c, cur, key, cr);
  int n = 0;
  uint32_t items[4] = {0u, 0u, 0u, 0u};
  if (!seen) {
    if (best < 0 || nd.h < best_h || (nd.h == best_h && nd.depth < best_depth)) {  // engine.py:66-69
      best = cur;
      best_h = nd.h;
      best_depth = nd.depth;
    }
    const int n_dead = cr.count_on(c.lv->dead);
    // Node.getChildren engine.py:14-25 + State.update :298-328
    for (int d = 0; d < 4; d++) {
      const int nx = px + DX[d], ny = py + DY[d];
      if (nx < 0 || ny < 0 || nx > c.lv->w - 1 || ny > c.lv->h - 1 || sk_bit(c.lv->solid, nx, ny)) continue;
      const int moved = cr.at(nx, ny);
      SkCrates<NH> ch = cr;
      int h = nd.h;  // the heuristic depends on the crates only
      if (moved >= 0) {
        const int bx = nx + DX[d], by = ny + DY[d];
        if (!sk_free_cell(c, cr, bx, by)) continue;
        // engine.py:22-23 checkDeadlock over all crates of the child
        const int ndead = n_dead - (sk_bit(c.lv->dead, nx, ny) ? 1 : 0) + (sk_bit(c.lv->dead, bx, by) ? 1 : 0);
        if (ndead > 0) continue;
        const uint32_t np = (uint32_t)bx | ((uint32_t)by << 8);
        ch.move(c, moved, np);
        h = sk_heuristic(c, ch);
      }
      if (c.n_nodes >= c.max_nodes) {  // cannot happen (<= 1 + 4 * iterations nodes per stage); reported if it does
        c.pool_full = true;
        continue;
      }
      const int k = c.n_nodes++;
      ch.store(c, k);
      if (c.lane == 0) {
        SokoNode nn;
        nn.parent = cur;
        nn.depth = nd.depth + 1;
        nn.h = h;
        nn.px = nx;
        nn.py = ny;
        c.nodes[k] = nn.pack();
      }
      const uint32_t item = ((uint32_t)(2 * h + b2 * (nd.depth + 1)) << 16) | (uint32_t)k;
#pragma unroll
      for (int i = 0; i < 4; i++) items[i] = i == n ? item : items[i];
      n++;
    }
  }
  if (c.lane == 0) {
    pp.res = sk_u32x4{0u, (uint32_t)n, 0u, 0u};
    pp.item = sk_u32x4{items[0], items[1], items[2], items[3]};
  }
}

// Body of expander wave k (1..3): serves the heap wave of stage k until the simulate wave leaves.
__device__ __attribute__((always_inline)) inline void sokoban_expander(const Params &p, int k) {
  const SokoPool &pool = *(const SokoPool *)p.soko;
  SokoShared &sh = sk_shared();
  SokoMail &m = sh.mail;
  SokoPipe &pp = sh.pipe[k];
  SokoCtx c;
  c.lv = &sh.level;
  c.lane = (int)(threadIdx.x & 63);
  c.pool_full = false;
  c.hl = (uint32_t SK_LDS *)nullptr;
  c.hcap = 0;
  c.dbg = nullptr;
  c.ncr = 0;
  c.cstride = 0;
  int seen = 0, best = -1, best_h = 0, best_depth = 0, b2 = 0;
  while (true) {
    int s;
    while ((s = __builtin_amdgcn_readfirstlane(sk_ld(&pp.jseq))) == seen) {
      if (__builtin_amdgcn_readfirstlane(sk_ld(&m.exit)) != 0) return;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    seen = s;
    const int cur = sk_u(sk_ld(&pp.cur));
    if (cur == SK_JOB_START) {  // a new search: workspace k of the slot, a fresh visited epoch, node 0 = the root
      const int slot = sk_u(sk_ld(&pp.slot)), px = sk_u(sk_ld(&pp.px)), py = sk_u(sk_ld(&pp.py));
      b2 = sk_u(sk_ld(&pp.b2));
      c.ncr = sk_u(sh.level.ncr);
      c.cstride = (c.ncr + 3) & ~3;
      sk_bind(c, pool, slot, k);
      uint32_t *epoch_word = &pool.epochs[slot * SK_STAGES + k];
      uint32_t ep = 0;
      if (c.lane == 0) ep = (atomicAdd(epoch_word, 1u) + 1u) & 0x7FFFu;
      ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
      if (ep == 0) {  // wrapped: start over with a clean table
        for (int i = c.lane; i < SK_VCAP; i += 64) c.vis[i] = sk_u32x4{0u, 0u, 0u, 0u};
        if (c.lane == 0) ep = (atomicAdd(epoch_word, 1u) + 1u) & 0x7FFFu;
        ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)ep);
      }
      c.epoch = ep;
      c.n_nodes = 1;
      best = -1;
      best_h = best_depth = 0;
      const int h0 = c.ncr > 64 ? sk_root<2>(c, px, py) : sk_root<1>(c, px, py);
      if (c.lane == 0) pp.res = sk_u32x4{0u, 0u, (uint32_t)h0, 0u};
    } else if (cur == SK_JOB_END) {
      if (c.lane == 0) {
        pp.res = sk_u32x4{0u, 0u, (uint32_t)best_h, (uint32_t)best_depth};
        if (c.pool_full) atomicOr(p.err, 2);
      }
      c.pool_full = false;
    } else {
      c.n_nodes = sk_u(c.n_nodes);
      if (c.ncr > 64) sk_expand<2>(c, pp, cur, b2, best, best_h, best_depth);
      else sk_expand<1>(c, pp, cur, b2, best, best_h, best_depth);
      best = sk_u(best);
      best_h = sk_u(best_h);
      best_depth = sk_u(best_depth);
    }
    if (c.lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      sk_st(&pp.rseq, s);
    }
  }
}

// The reference's cascade (sokoban_prob.py:99-148), all four stages on the calling wave (stage workspace 0).
template <int NH>
__device__ __attribute__((always_inline)) inline bool sk_cascade(SokoCtx &c, const SokoPool &pool, int slot, int power, int px, int py, int &h, int &depth) {
  sk_bind(c, pool, slot, 0);
  sk_root<NH>(c, px, py);
  // If the BFS stage expands the whole reachable state space without finding a win, no stage can win, each A*
  // stage would expand exactly the same set of states (pushes = 1 + sum of children over unique states, whatever
  // the order) and end with bestNode.h = min h over that set -- which the BFS stage already holds.  Skipping the
  // three A* stages is therefore exact (pinned by tests/golden/stats_sokoban_solver.npz against the reference).
  bool exhausted = false, won = false;
  for (int st = 0; st < SK_STAGES && !won && !exhausted; st++)  // (one call site: the stage is inlined once)
  {
    bool ex = false;
    won = sk_stage<NH>(c, pool, slot, st == 0 ? -1 : 3 - st, power, h, depth, &ex);
    exhausted = st == 0 && ex;  // (only the BFS stage's flag ends the cascade, see above)
  }
  return won;
}

// The same with helper waves: this wave runs the BFS stage, helper k (1..3) the A* stage with balance (3 - k) / 2.
template <int NH>
__device__ __attribute__((always_inline)) inline bool sk_cascade_helped(SokoCtx &c, const SokoPool &pool, int slot, int power, int px, int py, int &h, int &depth) {
  SokoMail &m = sk_shared().mail;
  int seq = 0;
  if (c.lane == 0) {
    seq = sk_ld(&m.seq) + 1;
    m.slot = slot;
    m.px = px;
    m.py = py;
    m.power = power;
    sk_st(&m.cancel_after, SK_STAGES);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the level in LDS, the dead-cell table, the fields above)
    sk_st(&m.seq, seq);
  }
  seq = __builtin_amdgcn_readfirstlane(seq);
  sk_bind(c, pool, slot, 0);
  sk_root<NH>(c, px, py);
  bool exhausted = false;
  bool won = sk_stage<NH>(c, pool, slot, -1, power, h, depth, &exhausted);
  if (won || exhausted) {
    if (c.lane == 0) sk_st(&m.cancel_after, 0);
  } else {
    for (int k = 1; k < SK_STAGES && !won; k++) {
      while (__builtin_amdgcn_readfirstlane(sk_ld(&m.done[k])) != seq) __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      won = __builtin_amdgcn_readfirstlane(sk_ld(&m.won[k])) != 0;
      h = __builtin_amdgcn_readfirstlane(sk_ld(&m.h[k]));
      depth = __builtin_amdgcn_readfirstlane(sk_ld(&m.depth[k]));
      if (won && c.lane == 0) atomicMin(&m.cancel_after, k);
    }
  }
  // the helpers are done with the slot (cancelled stages stop within one iteration)
  for (int k = 1; k < SK_STAGES; k++)
    while (__builtin_amdgcn_readfirstlane(sk_ld(&m.done[k])) != seq) __builtin_amdgcn_s_sleep(2);
  return won;
}

// Body of helper wave k (1..3) of a workgroup launched with Params::sk_helpers: serve the simulate wave's jobs until it
// leaves.  The caller has passed the workgroup barrier that follows sokoban_helpers_init.
__device__ __attribute__((always_inline)) inline void sokoban_helper(const Params &p, int k, uint32_t *lds_heap) {
  const SokoPool &pool = *(const SokoPool *)p.soko;
  SokoShared &sh = sk_shared();
  SokoMail &m = sh.mail;
  SokoCtx c;
  c.lv = &sh.level;
  c.lane = (int)(threadIdx.x & 63);
  c.pool_full = false;
  c.hl = (uint32_t SK_LDS *)lds_heap;  // [SK_LDS_HEAP], or null
  c.hcap = lds_heap ? SK_LDS_HEAP : 0;
#ifdef PCGRL_SK_TIMING
  c.dbg = (unsigned long long *)(p.err + 64);
#else
  c.dbg = nullptr;
#endif
  int seen = 0, pipe_seq = 0;
  while (true) {
    int s;
    while ((s = __builtin_amdgcn_readfirstlane(sk_ld(&m.seq))) == seen) {
      if (__builtin_amdgcn_readfirstlane(sk_ld(&m.exit)) != 0) return;
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    seen = s;
    const int slot = m.slot, px = m.px, py = m.py, power = m.power;
    c.ncr = sk_u(sh.level.ncr);
    c.cstride = (c.ncr + 3) & ~3;
    sk_bind(c, pool, slot, k);
    int h = 0, depth = 0;
    bool won = false;
    if (__builtin_amdgcn_readfirstlane(sk_ld(&m.cancel_after)) >= k) {
      // this wave keeps the open list; its expander wave owns the node records and the visited set (sk_stage_heap)
      SokoPipe &pp = sh.pipe[k];
      if (c.lane == 0) {
        pp.slot = slot;
        pp.px = px;
        pp.py = py;
        pp.b2 = 3 - k;  // balance 1, 0.5, 0
      }
      sk_pipe_post(pp, c.lane, ++pipe_seq, SK_JOB_START);
      sk_pipe_wait(pp, pipe_seq);
      const int h_root = sk_u((int)pp.res.z);
      won = sk_stage_heap(c, pp, pipe_seq, power, h_root, h, depth, &m.cancel_after, k);
    }
    if (c.lane == 0) {
      if (won) atomicMin(&m.cancel_after, k);
      m.won[k] = won ? 1 : 0;
      m.h[k] = h;
      m.depth[k] = depth;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      sk_st(&m.done[k], s);
    }
  }
}
// first helper wave, before the workgroup barrier: an empty mailbox
__device__ inline void sokoban_helpers_init() {
  SokoMail &m = sk_shared().mail;
  if ((threadIdx.x & 63) == 0) {
    m.seq = 0;
    m.exit = 0;
    m.cancel_after = SK_STAGES;
    for (int k = 0; k < SK_STAGES; k++) {
      m.done[k] = 0;
      sk_shared().pipe[k].jseq = 0;
      sk_shared().pipe[k].rseq = 0;
    }
  }
}
// simulate wave, when it leaves the kernel
__device__ inline void sokoban_helpers_release() {
  if ((threadIdx.x & 63) == 0) sk_st(&sk_shared().mail.exit, 1);
}

// The level of lane group `gi` in LDS (sk_shared().level), built by the whole wave: the group's row masks are broadcast;
// level coords = map coords + 1 (sokoban_prob.py:107-124: one-tile solid border around the map); crates / targets are listed
// in row-major order (engine.py:170-188).  Returns the player's cell and the list lengths (wave-uniform).
template <int LPE, typename M, int MAXC>
__device__ __attribute__((always_inline)) inline void sk_build_level(const Grp<LPE> &g, int gi, int H, int W, M solid, M player, M crate, M target,
                                                                     int &px, int &py, int &ncr, int &ntg) {
  SokoLevel &s_level = sk_shared().level;
  {
    // the level: the group's rows are broadcast to the wave; level coords = map coords + 1 (sokoban_prob.py:107-124:
    // one-tile solid border around the map); crates / targets are listed in row-major order (engine.py:170-188)
    const uint64_t full = W + 2 >= 64 ? ~0ull : (1ull << (W + 2)) - 1ull;
    auto row_of = [&](M v, int src) -> uint64_t {  // lane src's row mask, broadcast
      if constexpr (sizeof(M) == 4) {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, src);
      } else {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
        return (uint64_t)lo | ((uint64_t)hi << 32);
      }
    };
    if (g.lane == 0) {
      s_level.w = W + 2;
      s_level.h = H + 2;
      s_level.solid[0] = full;
      s_level.solid[H + 1] = full;
      s_level.tgt[0] = 0;
      s_level.tgt[H + 1] = 0;
    }
    for (int r = 0; r < H; r++) {
      const int src = gi * LPE + r;
      const uint64_t sm = row_of(solid, src), pl = row_of(player, src), cr = row_of(crate, src), tg = row_of(target, src);
      if (g.lane == 0) {
        s_level.solid[r + 1] = (sm << 1) | 1ull | (1ull << (W + 1));
        s_level.tgt[r + 1] = tg << 1;
      }
      if (pl) {
        px = __builtin_ctzll(pl) + 1;
        py = r + 1;
      }
      // lane j owns bit j of the row (W <= 62): list entry = count so far + rank of the bit
      const uint64_t below = (1ull << g.lane) - 1ull;
      if ((cr >> g.lane) & 1ull) {
        const int k = ncr + __popcll(cr & below);
        if (k < MAXC) s_level.root[k] = (uint16_t)((g.lane + 1) | ((r + 1) << 8));
      }
      if ((tg >> g.lane) & 1ull) {
        const int k = ntg + __popcll(tg & below);
        if (k < MAXC) s_level.target[k] = (uint16_t)((g.lane + 1) | ((r + 1) << 8));
      }
      ncr += __popcll(cr);
      ntg += __popcll(tg);
    }
  }
}

// Called by every lane of the wave in uniform control flow; `need` is uniform per group.  Groups that need the
// solver are served one after the other by the WHOLE wave, so a wavefront holds at most one workspace slot at a time
// and never waits for a slot while holding one.
// HUGE: the kernel also carries the search for levels with more than SK_MAXC pairs (every kernel except the compile-time
// 16x16 ones, whose maps hold at most 127 pairs)
template <int LPE, typename M, bool HUGE>
__device__ __attribute__((always_inline)) inline void sokoban_solve(const Grp<LPE> &g, const Params &p, int env, bool need, M solid, M player,
                                                        M crate, M target, int &dist_win, int &sol_len) {
  (void)env;
  const SokoPool &pool = *(const SokoPool *)p.soko;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  constexpr int EPW = 64 / LPE;
  SokoLevel &s_level = sk_shared().level;
  for (int gi = 0; gi < EPW; gi++) {
    const bool mine = need && (g.lane / LPE) == gi;
    if (__ballot(mine) == 0) continue;
    SokoCtx c;
    c.lv = &s_level;
    c.lane = g.lane;
    c.pool_full = false;
    c.hl = (uint32_t SK_LDS *)nullptr;
    c.hcap = 0;
#ifdef PCGRL_SK_TIMING
    c.dbg = (unsigned long long *)(p.err + 64);
#else
    c.dbg = nullptr;
#endif
    // take a workspace slot (lane 0; the slot index is broadcast)
    int slot = 0;
    if (g.lane == 0) {
      int s = (int)((blockIdx.x * 7u + gi) % (unsigned)pool.n_slots);
      while (atomicCAS(&pool.locks[s], 0, 1) != 0) {
        s = (s + 1) % pool.n_slots;
        __builtin_amdgcn_s_sleep(8);
      }
      slot = s;
    }
    slot = __builtin_amdgcn_readfirstlane(slot);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // tell the host that searches are running: it then launches the step kernel with one env per wavefront
    if (g.lane == 0 && p.solver_seen != nullptr) __hip_atomic_fetch_add(p.solver_seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int px = 0, py = 0, ncr = 0, ntg = 0;
    sk_build_level<LPE, M, (HUGE ? SK_MAXC_HUGE : SK_MAXC)>(g, gi, H, W, solid, player, crate, target, px, py, ncr, ntg);
    int dw = dist_win, sl = sol_len;
    constexpr int MAXC = HUGE ? SK_MAXC_HUGE : SK_MAXC;
    if (ncr > MAXC || ntg > MAXC || W + 2 > SK_MAXDIM || H + 2 > SK_MAXDIM) {
      if (g.lane == 0) atomicOr(p.err, 2);  // beyond the device solver's limits: reported by pcgrl_poll_error
    } else {
      c.ncr = ncr;
      c.cstride = (ncr + 3) & ~3;
      if (g.lane == 0) {
        s_level.ncr = ncr;
        s_level.ntg = ntg;
      }
      sk_init_deadlocks(c);
      bool won;
      int h = 0, depth = 0;
      if (ncr > SK_MAXC) {
        // more pairs than two registers per lane hold: this wave alone, stage after stage, eight registers per lane (the
        // helper waves, if any, get no job and keep waiting)
        if constexpr (HUGE) won = sk_cascade<SK_NH_HUGE>(c, pool, slot, p.cfg.solver_power, px, py, h, depth);
        else won = false;
      } else if (p.sk_helpers != 0) {
        if (ncr > 64) won = sk_cascade_helped<2>(c, pool, slot, p.cfg.solver_power, px, py, h, depth);
        else won = sk_cascade_helped<1>(c, pool, slot, p.cfg.solver_power, px, py, h, depth);
      } else {
        if (ncr > 64) won = sk_cascade<2>(c, pool, slot, p.cfg.solver_power, px, py, h, depth);
        else won = sk_cascade<1>(c, pool, slot, p.cfg.solver_power, px, py, h, depth);
      }
      if (won) {
        dw = 0;
        sl = depth;
      } else {
        dw = h;  // heuristic of the last stage's best node (sokoban_prob.py:147)
        sl = 0;
      }
      if (c.pool_full && g.lane == 0) atomicOr(p.err, 2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (g.lane == 0) atomicExch(&pool.locks[slot], 0);
    if (mine) {
      dist_win = dw;
      sol_len = sl;
    }
  }
}

// ---------------------------------------------------------------------------------------------- resumable solver
// Asynchronous stepping (pcgrl_step_ready).  In the reference a slow SokobanProblem._run_game (sokoban_prob.py:99-148: up to
// 4 x solver_power iterations) stalls ONE env -- a Ray worker's handful -- not the fleet (rl/utils.py:412-415).  Here a step
// launch used to wait for the slowest search of the whole batch (30-40 ms for a level that runs every stage to its cap).
// In asynchronous mode every env owns one stage workspace (slot = env: no locks) and a park record; a launch gives every
// search `Params::sk_budget` iteration units, an unfinished search is PARKED (its queue / heap / node records / visited table
// are in the workspace already; the dozen wave-uniform words of sk_stage go to the record) and the env reports itself busy;
// the next launch resumes it.  The cascade runs stage after stage on the env's simulate wave (sk_cascade's form): with most
// envs searching, the machine is full of independent searches and what counts is the work per search, not its latency.
//
// The record also carries the level the search belongs to (the map's three tile planes, exact): a search is resumed -- or
// its finished result reused -- only for exactly that level; for any other level the workspace starts over.  The result is a
// pure function of the level (and solver_power), so this is exact whatever happened to the env in between (resets,
// checkpoints restored into another engine, a step abandoned by pcgrl_reset).
struct SkPark {
  uint32_t state;   // 0 = empty, 1 = search parked, 2 = finished (res_*)
  int32_t stage;    // parked: the stage the search is in (0 = BFS, 1..3 = A* with balance 1, 0.5, 0)
  int32_t iters, head, tail, n_nodes, best, best_h, best_depth;
  uint32_t epoch;   // visited-table epoch counter of the workspace
  int32_t res_won, res_h, res_depth;
  int32_t pad_[3];
  uint64_t sig[3][SK_MAXDIM];  // tile planes of the level's map rows
};
static_assert(sizeof(SkPark) == 64 + 3 * SK_MAXDIM * 8, "SkPark: a 16-word header + the level");
constexpr int SK_PARK_EMPTY = 0, SK_PARK_RUNNING = 1, SK_PARK_DONE = 2;

// The cascade of sk_cascade on the env's own workspace, to a budget.  Returns true when the search ended in this launch
// (won / h / depth valid), false when it was parked.
template <int NH>
__device__ __attribute__((always_inline)) inline bool sk_cascade_budget(SokoCtx &c, const SokoPool &pool, SkPark *pk, bool resume, int stage0,
                                                                        SkRes &rs, int power, int px, int py, bool &won, int &h, int &depth) {
  if (!resume) sk_root<NH>(c, px, py);
  won = false;
  bool exhausted = false;
  for (int st = resume ? stage0 : 0; st < SK_STAGES && !won && !exhausted; st++) {
    bool ex = false;
    won = sk_stage<NH, true>(c, pool, 0, st == 0 ? -1 : 3 - st, power, h, depth, &ex, nullptr, 0, &rs);
    if (rs.parked) {
      if (c.lane == 0) {
        pk->stage = st;
        pk->iters = rs.iters;
        pk->head = rs.head;
        pk->tail = rs.tail;
        pk->n_nodes = rs.n_nodes;
        pk->best = rs.best;
        pk->best_h = rs.best_h;
        pk->best_depth = rs.best_depth;
        pk->epoch = rs.epoch;
        pk->state = SK_PARK_RUNNING;
      }
      return false;
    }
    exhausted = st == 0 && ex;  // (only the BFS stage's flag ends the cascade, see sk_cascade)
    rs.resume = false;          // the next stage starts from the root
  }
  return true;
}

// sokoban_solve for asynchronous stepping.  Same calling convention; returns (per lane, uniform over a group) whether the
// group's search is still unfinished -- then dist_win / sol_len are untouched and the caller must not use the statistics.
// `planes`: the map's tile planes of this lane's row (masked to the map): the level's identity.
template <int LPE, typename M>
__device__ __attribute__((always_inline)) inline bool sokoban_solve_async(const Grp<LPE> &g, const Params &p, int env, bool need, M solid, M player,
                                                                          M crate, M target, const M *planes, int &dist_win, int &sol_len) {
  const SokoPool &pool = *(const SokoPool *)p.soko;
  const int H = p.cfg.dims[0], W = p.cfg.dims[1];
  constexpr int EPW = 64 / LPE;
  SokoLevel &s_level = sk_shared().level;
  bool unfinished = false;
  for (int gi = 0; gi < EPW; gi++) {
    const bool mine = need && (g.lane / LPE) == gi;
    if (__ballot(mine) == 0) continue;
    const int ge = __builtin_amdgcn_readlane(env, gi * LPE);  // the group's env (uniform)
    SkPark *pk = (SkPark *)pool.async_park + ge;
    // the record's header, one word per lane, and whether it belongs to this very level
    const uint32_t hw = g.lane < 16 ? ((const uint32_t *)pk)[g.lane] : 0u;
    const bool my_row = (g.lane / LPE) == gi && g.row < H;
    bool diff = false;
    if (my_row) {
#pragma unroll
      for (int k = 0; k < 3; k++) diff = diff || pk->sig[k][g.row] != (uint64_t)planes[k];
    }
    const uint32_t state = (uint32_t)__builtin_amdgcn_readlane((int)hw, 0);
    const bool match = state != SK_PARK_EMPTY && __ballot(diff) == 0;
    int dw = dist_win, sl = sol_len;
    if (match && state == SK_PARK_DONE) {  // the same level again (a re-injected map, a replayed step): the result stands
      const int won = __builtin_amdgcn_readlane((int)hw, 10), rh = __builtin_amdgcn_readlane((int)hw, 11);
      const int rd = __builtin_amdgcn_readlane((int)hw, 12);
      dw = won ? 0 : rh;
      sl = won ? rd : 0;
      if (mine) {
        dist_win = dw;
        sol_len = sl;
      }
      continue;
    }
    SkRes rs;
    rs.resume = match;  // (state == SK_PARK_RUNNING)
    rs.parked = false;
    rs.budget = p.sk_budget;
    rs.iters = __builtin_amdgcn_readlane((int)hw, 2);
    rs.head = __builtin_amdgcn_readlane((int)hw, 3);
    rs.tail = __builtin_amdgcn_readlane((int)hw, 4);
    rs.n_nodes = __builtin_amdgcn_readlane((int)hw, 5);
    rs.best = __builtin_amdgcn_readlane((int)hw, 6);
    rs.best_h = __builtin_amdgcn_readlane((int)hw, 7);
    rs.best_depth = __builtin_amdgcn_readlane((int)hw, 8);
    rs.epoch = (uint32_t)__builtin_amdgcn_readlane((int)hw, 9);
    const int stage0 = __builtin_amdgcn_readlane((int)hw, 1);
    if (!match) {  // another level: the workspace starts over (the epoch counter lives on: older table entries are empty)
      if (my_row) {
#pragma unroll
        for (int k = 0; k < 3; k++) pk->sig[k][g.row] = (uint64_t)planes[k];
      }
      if (g.lane == 0) pk->state = SK_PARK_EMPTY;
    }
    SokoCtx c;
    c.lv = &s_level;
    c.lane = g.lane;
    c.pool_full = false;
    // The top SK_ASYNC_HEAP entries of the A* open list live in LDS while a launch works on the search (as the helper waves of
    // the synchronous kernels keep theirs): a pop or a push of a heap that fits walks no global memory at all, where each of
    // the sift's rounds was a dependent round trip -- most of an A* iteration's time in this mode.  Loaded at resume, written
    // back at park (sk_stage); static, so every kernel that runs the resumable solver has it (one search at a time per wave).
    __shared__ uint32_t s_async_heap[SK_ASYNC_HEAP];
    c.hl = (uint32_t SK_LDS *)s_async_heap;
    c.hcap = SK_ASYNC_HEAP;
    c.dbg = nullptr;
    int px = 0, py = 0, ncr = 0, ntg = 0;
    sk_build_level<LPE, M, SK_MAXC>(g, gi, H, W, solid, player, crate, target, px, py, ncr, ntg);
    bool finished = true;
    if (ncr > SK_MAXC || ntg > SK_MAXC || W + 2 > SK_MAXDIM || H + 2 > SK_MAXDIM) {
      // beyond what one stage workspace holds (more than 128 pairs: maps of >= 258 cells only): reported by pcgrl_poll_error,
      // the level keeps the solver-less statistics
      if (g.lane == 0) atomicOr(p.err, 2);
    } else {
      c.ncr = ncr;
      c.cstride = (ncr + 3) & ~3;
      if (g.lane == 0) {
        s_level.ncr = ncr;
        s_level.ntg = ntg;
      }
      sk_init_deadlocks(c);
      sk_bind_at(c, pool, pool.async_ws + (size_t)ge * pool.stage_bytes, 0);
      bool won = false;
      int h = 0, depth = 0;
      if (ncr > 64) finished = sk_cascade_budget<2>(c, pool, pk, rs.resume, stage0, rs, p.cfg.solver_power, px, py, won, h, depth);
      else finished = sk_cascade_budget<1>(c, pool, pk, rs.resume, stage0, rs, p.cfg.solver_power, px, py, won, h, depth);
      if (finished) {
        dw = won ? 0 : h;  // heuristic of the last stage's best node (sokoban_prob.py:147)
        sl = won ? depth : 0;
        if (g.lane == 0) {
          pk->res_won = won ? 1 : 0;
          pk->res_h = h;
          pk->res_depth = depth;
          pk->epoch = rs.epoch;
          pk->state = SK_PARK_DONE;
        }
      }
      if (c.pool_full && g.lane == 0) atomicOr(p.err, 2);
    }
    if (mine) {
      if (finished) {
        dist_win = dw;
        sol_len = sl;
      } else {
        unfinished = true;
      }
    }
  }
  return unfinished;
}

// ---------------------------------------------------------------------------------------------- host side
static inline int sokoban_slots_for(int n_levels) {  // one slot per four levels, between 4 and 512
  const int want = (n_levels + 3) / 4;
  return want < 4 ? 4 : (want > 512 ? 512 : want);
}
constexpr int SK_SLOTS_AT_CREATE = 64;  // pool of an engine whose solver has not been seen running yet (2.9 GB)
// A pool of n_slots workspace slots (the engine grows it: pcgrl_engine.hip soko_pool_for).
static inline hipError_t sokoban_alloc(Params &p, std::vector<void *> &allocs, int n_slots, int *n_slots_out = nullptr,
                                       void *async_ws = nullptr, void *async_park = nullptr) {
  SokoPool pool;
  pool.async_ws = (uint8_t *)async_ws;  // (a grown pool keeps the per-env workspaces of asynchronous stepping)
  pool.async_park = async_park;
  // A slot = SK_STAGES stage workspaces (~11 MB each at the default solver_power: 46 MB per slot).  Full size: one slot
  // per four envs of the batch, between 4 (a single-env adapter or an RLlib worker's small engine pins 0.18 GB, not 2.9)
  // and 512 (23 GB in all at 2048 envs); searches beyond the pool wait for a slot.  pcgrl_create allocates at most
  // SK_SLOTS_AT_CREATE of them: most workloads (random rollouts) hardly ever meet the solver's precondition.
  pool.n_slots = n_slots < 1 ? 1 : n_slots;
  if (n_slots_out) *n_slots_out = pool.n_slots;
  pool.max_nodes = 4 * (p.cfg.solver_power > 0 ? p.cfg.solver_power : 1) + 8;
  const size_t vis_off = SK_NODE_BYTES * (size_t)pool.max_nodes + (size_t)pool.max_nodes * SK_MAXC * sizeof(uint16_t);
  size_t sz = vis_off + SK_VIS_BYTES + sizeof(uint32_t) * (size_t)pool.max_nodes;
  pool.stage_bytes = (sz + 255) & ~(size_t)255;
  const size_t n_ws = (size_t)pool.n_slots * SK_STAGES;
  hipError_t e;
  void *base = nullptr, *locks = nullptr, *epochs = nullptr, *dpool = nullptr;
  // (all or nothing: a failure part-way hands back what this call allocated instead of leaving tens of GB pinned in `allocs`)
  auto undo = [&](hipError_t err) {
    (void)hipDeviceSynchronize();
    for (void *q : {dpool, epochs, locks, base})
      if (q) (void)hipFree(q);
    return err;
  };
  if ((e = hipMalloc(&base, pool.stage_bytes * n_ws)) != hipSuccess) return undo(e);
  // only the visited tables need a defined start (entries carry the epoch of the stage that wrote them; 0 = empty)
  for (size_t s = 0; s < n_ws; s++)
    if ((e = hipMemsetAsync((uint8_t *)base + s * pool.stage_bytes + vis_off, 0, SK_VIS_BYTES, 0)) != hipSuccess) return undo(e);
  if ((e = hipMalloc(&locks, sizeof(int32_t) * pool.n_slots)) != hipSuccess) return undo(e);
  if ((e = hipMemset(locks, 0, sizeof(int32_t) * pool.n_slots)) != hipSuccess) return undo(e);
  if ((e = hipMalloc(&epochs, sizeof(uint32_t) * n_ws)) != hipSuccess) return undo(e);
  if ((e = hipMemset(epochs, 0, sizeof(uint32_t) * n_ws)) != hipSuccess) return undo(e);
  pool.base = (uint8_t *)base;
  pool.locks = (int32_t *)locks;
  pool.epochs = (uint32_t *)epochs;
  if ((e = hipMalloc(&dpool, sizeof(SokoPool))) != hipSuccess) return undo(e);
  if ((e = hipMemcpy(dpool, &pool, sizeof(pool), hipMemcpyHostToDevice)) != hipSuccess) return undo(e);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return undo(e);
  for (void *q : {base, locks, epochs, dpool}) allocs.push_back(q);
  p.soko = dpool;
  return hipSuccess;
}

// Asynchronous stepping: one stage workspace and one park record per env (pcgrl_set_solver_budget); the device pool record
// learns where they are.
static inline hipError_t sokoban_alloc_async(Params &p, std::vector<void *> &allocs, int n_envs, void **ws_out, void **park_out) {
  SokoPool pool;
  hipError_t e;
  if ((e = hipMemcpy(&pool, p.soko, sizeof(pool), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  void *ws = nullptr, *park = nullptr;
  auto undo = [&](hipError_t err) {
    (void)hipDeviceSynchronize();
    for (void *q : {park, ws})
      if (q) (void)hipFree(q);
    return err;
  };
  if ((e = hipMalloc(&ws, pool.stage_bytes * (size_t)n_envs)) != hipSuccess) return undo(e);
  const size_t vis_off = SK_NODE_BYTES * (size_t)pool.max_nodes + (size_t)pool.max_nodes * SK_MAXC * sizeof(uint16_t);
  for (int i = 0; i < n_envs; i++)  // (entries carry the epoch of the stage that wrote them; 0 = empty)
    if ((e = hipMemsetAsync((uint8_t *)ws + (size_t)i * pool.stage_bytes + vis_off, 0, SK_VIS_BYTES, 0)) != hipSuccess) return undo(e);
  if ((e = hipMalloc(&park, sizeof(SkPark) * (size_t)n_envs)) != hipSuccess) return undo(e);
  if ((e = hipMemset(park, 0, sizeof(SkPark) * (size_t)n_envs)) != hipSuccess) return undo(e);
  pool.async_ws = (uint8_t *)ws;
  pool.async_park = park;
  if ((e = hipMemcpy(p.soko, &pool, sizeof(pool), hipMemcpyHostToDevice)) != hipSuccess) return undo(e);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return undo(e);
  allocs.push_back(ws);
  allocs.push_back(park);
  *ws_out = ws;
  *park_out = park;
  return hipSuccess;
}

}  // namespace pcgrl
