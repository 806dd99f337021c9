// pcgrl_engine.hip -- host side of libpcgrl_amd.so: the C ABI declared in include/pcgrl_amd.h.
//
// Owns the per-env state in HBM, seeds the numpy-compatible RNG streams, validates configs and launches the
// gfx950 kernels of pcgrl_kernels2d.h / pcgrl_kernels3d.h on the caller's HIP stream.  No torch types, no
// hidden synchronisation in reset/step/observe.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <map>
#include <mutex>
#include <tuple>

#include "pcgrl_common.h"
#include "pcgrl_dispatch.h"
#include "pcgrl_kernels2d.h"  // U128 / Pcg (host side of the RNG), the small engine-level kernels
#include "pcgrl_sokoban.h"    // sokoban_alloc
#include "pcgrl_kernels3d.h"  // M3_* limits

using namespace pcgrl;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t _e = (x);                                                                            \
    if (_e != hipSuccess) return fail(PCGRL_EHIP, std::string(#x) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace pcgrl {
struct RedScratch;
struct SampleState;
}
struct pcgrl_engine {
  Params p;
  int device = 0;
  int lpe = 16;
  size_t lds_bytes = 0;
  int cpl = 0;  // 3-D: cells per lane of the reset RNG split
  int32_t *seen_host = nullptr;  // sokoban: host-mapped counter of device solver runs
  int soko_slots = 0;            // sokoban: workspace slots of the current pool (p.soko)
  int32_t seen_last = 0, spread_left = 0;
  bool soko_lazy = true;         // grow the solver pool by itself the first time the solver has been seen running
  int sk_budget = 0;             // sokoban, asynchronous stepping: solver iteration units per env and launch (0 = synchronous)
  void *sk_ws = nullptr, *sk_park = nullptr;  // ... and its per-env stage workspaces / park records (also in the device SokoPool)
  bool soko_grow_failed = false; // the last growth attempt failed (message in soko_grow_msg): not retried by itself
  std::string soko_grow_msg;
  uint8_t *hdr_host = nullptr;   // pinned: the 256-byte header of pcgrl_export_state images (rebuilt by pcgrl_set_static)
  // development switches, read ONCE at pcgrl_create (never getenv() in a stepping call: not thread-safe against a host's
  // putenv, and a stray variable must not steer a running engine): PCGRL_ROLLOUT_KERNEL 1 = always the one-launch rollout
  // kernel, 0 = always step launches, unset = by shape; PCGRL_OBS_NT_MB = observation bytes per launch (MB) from which the
  // stores are non-temporal (default 384, 0 = never)
  int rollout_form = -1;
  int rollout_epw = -1;  // PCGRL_ROLLOUT_EPW (development): envs per wavefront of the rollout's simulate role, -1 = chosen by batch size
  // pcgrl_rollout as two kernels (16x16 compile-time kernels, plain mode): the observe role runs on this stream from a
  // snapshot of the pre-call state (tile planes, the scalars of the hot state line, both RNG streams)
  bool split_ok = false;
  int64_t two_role_resident = 0;  // workgroups of the two-role rollout kernel the device holds at once (occupancy x CUs)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void *snap_planes = nullptr;
  EnvState *snap_st = nullptr;
  RngState *snap_rng = nullptr;
  long obs_nt_mb = 384;
  bool maybe_stale = false;  // pcgrl_update ran since the last refresh / full reset: some env may carry ENV_STATS_DIRTY
  int64_t obs_bytes = 0;
  int obs_ndim = 0;
  int32_t obs_shape[4] = {0, 0, 0, 0};
  RedScratch *red = nullptr;  // pcgrl_reduce_episodes block partials
  SampleState *sample = nullptr;  // pcgrl_sample_actions: draw counter + ticket
  std::vector<void *> allocs;
  // the persistent per-env arrays (pointer, bytes per env): what pcgrl_export_state / pcgrl_import_state carry
  std::vector<std::pair<void *, size_t>> state_arrays;
};

extern "C" {
static void state_header_build(pcgrl_engine *h);
}

// ---------------------------------------------------------------------------------------------- RNG seeding (host)
// numpy.random.SeedSequence(seed).generate_state(4, uint64) -> PCG64 state/inc (numpy/random/bit_generator.pyx,
// _pcg64.pyx, src/pcg64/pcg64.h); reached in the reference through gymnasium.utils.seeding.np_random
// (envs/reps/representation.py:50-53, envs/probs/problem.py:79-81).
static void seedseq_words(uint64_t seed, uint64_t out[4]) {
  const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
  const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
  uint32_t ent[4] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0, 0};
  int n_ent = ent[1] ? 2 : 1;
  uint32_t pool[4], hc = INIT_A;
  auto hashmix = [&](uint32_t v) {
    v ^= hc;
    hc *= MULT_A;
    v *= hc;
    v ^= v >> 16;
    return v;
  };
  for (int i = 0; i < 4; i++) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) {
        uint32_t r = MIX_L * pool[d] - MIX_R * hashmix(pool[s]);
        r ^= r >> 16;
        pool[d] = r;
      }
  uint32_t w[8], hb = INIT_B;
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3] ^ hb;
    hb *= MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  for (int i = 0; i < 4; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

static void pcg64_seed_state(uint64_t seed, uint64_t st[4]) {
  uint64_t w[4];
  seedseq_words(seed, w);
  const U128 mult{PCG_MULT_HI, PCG_MULT_LO};
  U128 initstate{w[0], w[1]}, initseq{w[2], w[3]};
  U128 inc{(initseq.hi << 1) | (initseq.lo >> 63), (initseq.lo << 1) | 1ull};
  U128 s{0, 0};
  s = add128(mul128(s, mult), inc);
  s = add128(s, initstate);
  s = add128(mul128(s, mult), inc);
  st[0] = s.hi;
  st[1] = s.lo;
  st[2] = inc.hi;
  st[3] = inc.lo;
}

// ---------------------------------------------------------------------------------------------- config
static int n_tiles_of(int prob) {
  switch (prob) {
    case PCGRL_PROB_BINARY: return 2;
    case PCGRL_PROB_ZELDA: return 8;
    case PCGRL_PROB_SOKOBAN: return 5;
    case PCGRL_PROB_MC3DMAZE: return 2;
  }
  return 0;
}
static int n_stats_of(int prob) {
  switch (prob) {
    case PCGRL_PROB_BINARY: return 2;
    case PCGRL_PROB_ZELDA: return 7;
    case PCGRL_PROB_SOKOBAN: return 7;
    case PCGRL_PROB_MC3DMAZE: return 3;
  }
  return 0;
}

static int validate(const pcgrl_config &c, int &lpe, int64_t &obs_bytes, int &obs_chunks, int32_t shape[4], int &ndim) {
  if (c.problem < 0 || c.problem > PCGRL_PROB_MC3DMAZE) return fail(PCGRL_EINVAL, "unknown problem");
  if (c.representation < 0 || c.representation > PCGRL_REP_WIDE) return fail(PCGRL_EINVAL, "unknown representation");
  if (c.n_stats != n_stats_of(c.problem)) return fail(PCGRL_EINVAL, "n_stats does not match the problem");
  if (c.n_ctrl < 0 || c.n_ctrl > c.n_stats) return fail(PCGRL_EINVAL, "n_ctrl out of range");
  for (int j = 0; j < c.n_ctrl; j++)
    if (c.ctrl_idx[j] < 0 || c.ctrl_idx[j] >= c.n_stats || !(c.ctrl_range[j] > 0.0) || !c.has_trg[c.ctrl_idx[j]])
      return fail(PCGRL_EINVAL, "control metric: bad stat index, non-positive range, or a stat without a target");
  const int nt = n_tiles_of(c.problem);
  if (c.problem == PCGRL_PROB_MC3DMAZE) {
    if (c.ndim != 3) return fail(PCGRL_EINVAL, "minecraft_3D_maze needs ndim == 3");
    if (c.static_tiles || c.act_window[0] != 0)
      return fail(PCGRL_EUNSUPPORTED, "static tiles / act_window are only on the accelerated path for 2-D problems");
    if (c.representation != PCGRL_REP_NARROW)
      return fail(PCGRL_EUNSUPPORTED, "minecraft_3D_maze: only the narrow representation is on the accelerated path");
    const int Z = c.dims[0], Y = c.dims[1], X = c.dims[2];
    if (!m3_supported(Z, Y, X))
      return fail(PCGRL_EUNSUPPORTED, "minecraft_3D_maze: need 1 <= Z, Y, X <= 16 (the reference's stock map is 15 x 15 x 15)");
    const int64_t cells = (int64_t)c.obs_window[0] * c.obs_window[1] * c.obs_window[2];
    if (cells < 1 || cells % 4) return fail(PCGRL_EUNSUPPORTED, "3-D obs_window volume must be a positive multiple of 4");
    lpe = 64;
    obs_chunks = 0;
    obs_bytes = cells * 4;
    shape[0] = c.obs_window[0];
    shape[1] = c.obs_window[1];
    shape[2] = c.obs_window[2];
    shape[3] = 4;  // out-of-bounds, AIR, DIRT, path overlay
    ndim = 4;
    return PCGRL_OK;
  }
  if (c.ndim != 2) return fail(PCGRL_EINVAL, "2-D problem needs ndim == 2");
  const int H = c.dims[0], W = c.dims[1];
  if (c.act_window[0] != 0 || c.act_window[1] != 0) {  // reps/wrappers.py:397-545; turtle / wide raise in the reference
    if (c.representation != PCGRL_REP_NARROW)
      return fail(PCGRL_EUNSUPPORTED, "act_window: the reference's MultiActionRepresentation only runs on narrow");
    if (c.act_window[0] < 1 || c.act_window[1] < 1 || c.act_window[0] > H || c.act_window[1] > W)
      return fail(PCGRL_EINVAL, "act_window must fit inside map_shape");
  }
  if (c.static_tiles) {  // reps/wrappers.py:234-376; wide raises in the reference (update() got an unexpected 'pos')
    if (c.representation == PCGRL_REP_WIDE)
      return fail(PCGRL_EUNSUPPORTED, "static tiles: the reference's StaticTileRepresentation does not run on wide");
    if (!(c.static_prob >= 0.0 && c.static_prob <= 1.0) || c.n_static_walls < 0)
      return fail(PCGRL_EINVAL, "static_prob must be in [0, 1] and n_static_walls >= 0");
    if (c.n_static_walls > 0 && (H < 3 || W < 3)) return fail(PCGRL_EINVAL, "static walls need a map of at least 3x3");
  } else if (c.static_prob != 0.0 || c.n_static_walls != 0) {
    return fail(PCGRL_EINVAL, "static_prob / n_static_walls need static_tiles = 1");
  }
  if (H < 1 || W < 1 || H > 64 || W > 64) return fail(PCGRL_EUNSUPPORTED, "map_shape: need 1 <= H <= 64, 1 <= W <= 64");
  if (c.problem == PCGRL_PROB_SOKOBAN && (W + 2 > SK_MAXDIM || H + 2 > SK_MAXDIM))
    return fail(PCGRL_EUNSUPPORTED, "sokoban: the device solver's level (map + border) is at most 64 x 64: need H, W <= 62");
  if (c.problem == PCGRL_PROB_SOKOBAN && (c.solver_power < 1 || c.solver_power > SK_MAX_POWER))
    return fail(PCGRL_EUNSUPPORTED, "sokoban: solver_power must be in [1, " + std::to_string(SK_MAX_POWER) +
                                        "] (the device solver's visited table and node ids are sized for that)");
  lpe = H <= 8 ? 8 : (H <= 16 ? 16 : (H <= 32 ? 32 : 64));
  if (W > 32 && lpe < 32) lpe = 32;  // the 64-bit row-mask kernels are built for 32 and 64 lanes per env: short maps leave rows idle
  if (c.representation == PCGRL_REP_WIDE) {
    if (c.obs_window[0] != H || c.obs_window[1] != W)
      return fail(PCGRL_EINVAL, "wide representation needs obs_window == map_shape (reference wrappers.py:140-150 reshape)");
    if (H != W) return fail(PCGRL_EUNSUPPORTED, "wide representation: the reference's transposed write needs a square map");
    obs_chunks = (W * nt + 15) / 16;  // rows that are not a multiple of 16 bytes take the byte-string store path
    obs_bytes = (int64_t)H * W * nt;
    shape[0] = H;
    shape[1] = W;
    shape[2] = nt;
    ndim = 3;
  } else {
    const int OH = c.obs_window[0], OW = c.obs_window[1], C = nt + 1 + (c.static_tiles ? 1 : 0);
    if (OH < 1 || OW < 1) return fail(PCGRL_EINVAL, "obs_window must be positive");
    obs_chunks = (OW * C + 15) / 16;
    obs_bytes = (int64_t)OH * OW * C;
    shape[0] = OH;
    shape[1] = OW;
    shape[2] = C;
    ndim = 3;
  }
  return PCGRL_OK;
}

static std::vector<JumpEntry> make_jump_table(int H, int W, int last = -1) {
  // A_k = a^k, G_k = 1 + a + ... + a^(k-1)  (mod 2^128); entry r = skip r*W draws, entry H = H*W draws
  // (or `last` draws when given: 3-D uses H = 64 lanes, W = cells per lane, last = n_cells)
  std::vector<JumpEntry> t(H + 1);
  U128 A{0, 1}, G{0, 0};
  const U128 a{PCG_MULT_HI, PCG_MULT_LO};
  int k = 0;
  for (int r = 0; r <= H; r++) {
    const int want = (r == H && last >= 0) ? last : r * W;
    if (want < k) {  // only for the final entry when last < H*W: restart
      A = U128{0, 1};
      G = U128{0, 0};
      k = 0;
    }
    while (k < want) {
      G = add128(mul128(G, a), U128{0, 1});
      A = mul128(A, a);
      k++;
    }
    t[r] = JumpEntry{A.hi, A.lo, G.hi, G.lo};
  }
  return t;
}

// observation bytes of one launch from which the 16x16 / 3-D kernels store non-temporally (profiles/r05_dev_traces.md)
static bool obs_nt_for(const pcgrl_engine *e, int64_t bytes_per_launch) {
  return e->obs_nt_mb > 0 && bytes_per_launch >= (int64_t)e->obs_nt_mb * 1000000;
}

// ---------------------------------------------------------------------------------------------- dispatch
// The kernels live in their own translation units (pcgrl_k_*.hip, see pcgrl_dispatch.h).
static hipError_t launch(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s, int cpl = 0) {
  const bool wide64 = p.cfg.dims[1] > 32;  // 64-bit row masks (W <= 64); validate() guarantees lpe >= 32 there
  switch (p.cfg.problem) {
    case PCGRL_PROB_MC3DMAZE: return launch_3d(id, p, cpl, s);
    case PCGRL_PROB_BINARY: return wide64 ? launch_binary64(id, lpe, p, lds, s) : launch_binary32(id, lpe, p, lds, s);
    case PCGRL_PROB_ZELDA: return wide64 ? launch_zelda64(id, lpe, p, lds, s) : launch_zelda32(id, lpe, p, lds, s);
    default: return wide64 ? launch_sokoban64(id, lpe, p, lds, s) : launch_sokoban32(id, lpe, p, lds, s);
  }
}

// Every entry point runs on the engine's device whatever the caller's current device is, and leaves the caller's
// current device as it found it.
struct DeviceGuard {
  int prev = -1;
  hipError_t err;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    else if (err == hipSuccess) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define ON_DEVICE(dev)                                                                            \
  DeviceGuard _guard(dev);                                                                        \
  if (_guard.err != hipSuccess) return fail(PCGRL_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(_guard.err))

// ---------------------------------------------------------------------------------------------- engine-level kernels
namespace pcgrl {

__global__ __launch_bounds__(64) void queue_targets_kernel(Params p, const double *lo, const double *hi) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= p.n_envs || (p.mask != nullptr && p.mask[env] == 0)) return;
  double *q = p.trg_pending + (size_t)env * PCGRL_MAX_STATS * 2;
  for (int k = 0; k < p.cfg.n_stats; k++) {
    q[2 * k] = lo[(size_t)env * p.cfg.n_stats + k];
    q[2 * k + 1] = hi[(size_t)env * p.cfg.n_stats + k];
  }
  p.trg_flag[env] |= 1;  // (bits 1..: the env's resampling draw counter)
}

// pcgrl_reduce_episodes: sum of the per-env episode totals in a FIXED order (block b owns envs [b*chunk, (b+1)*chunk),
// the last block to finish adds the block partials in block order), so the result does not depend on timing.
constexpr int RED_BLOCKS = 64, RED_THREADS = 256, RED_W = 3 + PCGRL_MAX_STATS;
struct RedScratch {
  double part[RED_BLOCKS][RED_W];
  unsigned int ticket;
};
__global__ __launch_bounds__(RED_THREADS) void reduce_episodes_kernel(Params p, RedScratch *scr, double *out, int n_blocks, int clear) {
  __shared__ double sh[RED_THREADS / 64][RED_W];
  __shared__ bool last;
  const int chunk = (p.n_envs + n_blocks - 1) / n_blocks;
  const int lo = blockIdx.x * chunk, hi = min(p.n_envs, lo + chunk);
  double ret = 0.0;
  int64_t v[RED_W - 1];
  for (int k = 0; k < RED_W - 1; k++) v[k] = 0;
  for (int e = lo + (int)threadIdx.x; e < hi; e += RED_THREADS) {  // fixed env -> thread assignment
    EpAcc *A = &p.st[e].acc;
    ret += A->sum_return;
    v[0] += A->sum_len;
    v[1] += A->n;
    for (int k = 0; k < PCGRL_MAX_STATS; k++) v[2 + k] += A->sum_stats[k];
    if (clear) {
      A->sum_return = 0.0;
      A->sum_len = 0;
      A->n = 0;
      for (int k = 0; k < PCGRL_MAX_STATS; k++) A->sum_stats[k] = 0;
    }
  }
  // lanes -> wave (xor butterfly: a fixed tree), waves -> block (in wave order)
  double w[RED_W];
  w[0] = ret;
  for (int k = 1; k < RED_W; k++) w[k] = (double)v[k - 1];  // integer totals < 2^53: exact in double from here on
  for (int k = 0; k < RED_W; k++)
    for (int o = 32; o >= 1; o >>= 1) w[k] += __shfl_xor(w[k], o, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < RED_W; k++) sh[wave][k] = w[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < RED_W; k++) {
      double s = 0.0;
      for (int i = 0; i < RED_THREADS / 64; i++) s += sh[i][k];
      scr->part[blockIdx.x][k] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(&scr->ticket, 1u) == (unsigned)n_blocks - 1u;
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  if (threadIdx.x < RED_W) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; b++) s += __hip_atomic_load(&scr->part[b][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // out = sum_return, sum_len, n_episodes, sum_final_stats[n_stats]
    if ((int)threadIdx.x < 3 + p.cfg.n_stats) out[threadIdx.x] = s;
  }
  if (threadIdx.x == 0) scr->ticket = 0;  // ready for the next call on the same stream
}

// small batches: one block, no cross-block hand-off (same fixed summation order: thread t owns envs t, t + 1024, ...;
// lanes -> wave by an xor butterfly, waves in wave order)
__global__ __launch_bounds__(1024) void reduce_episodes_small_kernel(Params p, double *out, int clear) {
  __shared__ double sh[16][RED_W];
  const int cols = 3 + p.cfg.n_stats;
  // The totals sit in the envs' second state line, which only episode ends touch: cold in HBM.  All of a thread's
  // counters are requested before the first is looked at, so their latencies overlap (n_envs <= 16384: 16 per thread).
  int64_t cnt[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int e = (int)threadIdx.x + 1024 * i;
    cnt[i] = e < p.n_envs ? p.st[e].acc.n : 0;
  }
  double ret = 0.0;
  int64_t v[RED_W - 1];
  for (int k = 0; k < RED_W - 1; k++) v[k] = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (cnt[i] == 0) continue;  // nothing finished in this env since the last clearing call
    EpAcc *A = &p.st[(int)threadIdx.x + 1024 * i].acc;
    ret += A->sum_return;
    v[0] += A->sum_len;
    v[1] += cnt[i];
    for (int k = 0; k < PCGRL_MAX_STATS; k++) v[2 + k] += A->sum_stats[k];
    if (clear) {
      A->sum_return = 0.0;
      A->sum_len = 0;
      A->n = 0;
      for (int k = 0; k < PCGRL_MAX_STATS; k++) A->sum_stats[k] = 0;
    }
  }
  double w[RED_W];
  w[0] = ret;
  for (int k = 1; k < RED_W; k++) w[k] = (double)v[k - 1];
#pragma unroll
  for (int k = 0; k < RED_W; k++) {
    if (k >= cols) break;  // (uniform: only the problem's own statistics are summed)
    for (int o = 32; o >= 1; o >>= 1) w[k] += __shfl_xor(w[k], o, 64);
  }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < RED_W; k++) sh[threadIdx.x >> 6][k] = w[k];
  __syncthreads();
  if ((int)threadIdx.x < cols) {
    double s = 0.0;
    for (int i = 0; i < 16; i++) s += sh[i][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// pcgrl_sample_actions: action_space.sample() for every env.  Counter-based (entry i of draw c under seed s is a pure
// function of (s, c, i)); the draw counter lives in device memory and is advanced by the last block to finish, after every
// block has read it, so a launch captured in a HIP graph draws new actions at every replay.
struct SampleState {
  unsigned long long counter;
  unsigned int ticket;
};
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void sample_actions_kernel(int32_t *out, int32_t n, uint32_t n_actions, uint64_t seed, SampleState *st) {
  __shared__ unsigned long long c_sh;
  if (threadIdx.x == 0) c_sh = __hip_atomic_load(&st->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint64_t c = c_sh;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const uint64_t r = mix64(mix64(seed + c * 0x9e3779b97f4a7c15ull) ^ ((uint64_t)i * 0xd1b54a32d192ed03ull + 0x8cb92ba72f3d8dd7ull));
    out[i] = (int32_t)__umul64hi(r, (uint64_t)n_actions);  // floor(r * n / 2^64): bias < n / 2^64
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1u) {  // every block has read the counter before its ticket
      st->ticket = 0;
      __hip_atomic_store(&st->counter, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// pcgrl_get_rng_state / pcgrl_set_rng_state: [N][10] = rep state hi, lo, inc hi, lo; prob state hi, lo, inc hi, lo;
// representation-wrapper flags | has32 << 32, val32
__global__ __launch_bounds__(64) void rng_state_kernel(Params p, uint64_t *out, const uint64_t *in) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= p.n_envs || (p.mask != nullptr && p.mask[env] == 0)) return;
  RngState *r = &p.rng[env];
  uint32_t *xs = p.xstate ? p.xstate + (size_t)env * 4 : nullptr;
  if (out) {
    uint64_t *o = out + (size_t)env * 10;
    for (int k = 0; k < 4; k++) {
      o[k] = r->rep[k];
      o[4 + k] = r->prob[k];
    }
    o[8] = xs ? ((uint64_t)xs[0] | ((uint64_t)xs[1] << 32)) : 0ull;
    o[9] = xs ? (uint64_t)xs[2] : 0ull;
  }
  if (in) {
    const uint64_t *i = in + (size_t)env * 10;
    for (int k = 0; k < 4; k++) {
      r->rep[k] = i[k];
      r->prob[k] = i[4 + k];
    }
    if (xs) {
      xs[0] = (uint32_t)i[8];
      xs[1] = (uint32_t)(i[8] >> 32);
      xs[2] = (uint32_t)i[9];
    }
  }
}

// pcgrl_rollout in its two-kernel form: what the observe kernel needs of the state as it is BEFORE the call -- the tile planes,
// the first 32 bytes of the hot state line (position, counters, flags) and both RNG streams -- copied in 16-byte pieces
__global__ __launch_bounds__(256) void rollout_snapshot_kernel(const uint4 *planes, uint4 *s_planes, int64_t plane_q, const uint4 *st, uint4 *s_st,
                                                               const uint4 *rng, uint4 *s_rng, int32_t n_envs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_st = (int64_t)n_envs * 2, n_rng = (int64_t)n_envs * (int64_t)(sizeof(RngState) / 16);
  if (i < plane_q) {
    s_planes[i] = planes[i];
  } else if (i < plane_q + n_st) {
    const int64_t j = i - plane_q, e = j >> 1, q = j & 1;
    s_st[e * (int64_t)(sizeof(EnvState) / 16) + q] = st[e * (int64_t)(sizeof(EnvState) / 16) + q];
  } else if (i < plane_q + n_st + n_rng) {
    const int64_t j = i - plane_q - n_st;
    s_rng[j] = rng[j];
  }
}

// pcgrl_env_busy: 1 = the env waits for a parked search (asynchronous stepping)
__global__ __launch_bounds__(256) void env_busy_kernel(Params p, uint8_t *out) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env < p.n_envs) out[env] = (p.st[env].flags & (ENV_PENDING_STEP | ENV_PENDING_STATS)) != 0 ? 1 : 0;
}

// pcgrl_import_state with a mask: rows (envs) of one state array, 4 bytes per thread
__global__ __launch_bounds__(256) void masked_rows_copy_kernel(uint32_t *dst, const uint32_t *src, int64_t words_per_env, int32_t n_envs,
                                                              const uint8_t *mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= words_per_env * n_envs) return;
  if (mask[i / words_per_env]) dst[i] = src[i];
}

}  // namespace pcgrl

// sokoban: while the device solver has been running in recent launches, step with one env per wavefront (every search gets
// a wave of its own); otherwise 64 / LPE envs share a wave.  The counter lives in host-mapped memory: reading it costs
// nothing and may lag a launch or two, which only delays the switch.
// sokoban: the solver's workspace pool is allocated at full size (one slot per four envs, sokoban_slots_for) the first time
// the solver has been SEEN running, not at pcgrl_create: until then at most SK_SLOTS_AT_CREATE slots exist (searches
// beyond the pool wait for a slot: speed, not results).  Synchronous (hipMalloc + a device synchronise), once per engine,
// never while `stream` is being captured; launches issued earlier (or frozen in a HIP graph) keep using the pool they were
// issued with, which stays allocated until pcgrl_destroy.
static int soko_pool_for(pcgrl_engine *h, Params &p, int want, hipStream_t stream, bool lazy = true) {
  if (!h->p.soko || h->soko_slots >= want) return PCGRL_OK;
  if (lazy && (!h->soko_lazy || h->soko_grow_failed)) return PCGRL_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream != nullptr && (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
    (void)hipGetLastError();
    return lazy ? PCGRL_OK : fail(PCGRL_EINVAL, "solver pool: `stream` is being captured (hipMalloc is not capturable)");
  }
  Params grown = h->p;
  int got = 0;
  const hipError_t e = sokoban_alloc(grown, h->allocs, want, &got, h->sk_ws, h->sk_park);
  if (e == hipSuccess) {
    h->p.soko = grown.soko;
    p.soko = grown.soko;
    h->soko_slots = got;
    h->soko_grow_failed = false;
    h->soko_grow_msg.clear();
    return PCGRL_OK;
  }
  (void)hipGetLastError();
  // (out of memory: sokoban_alloc handed back what it had taken; the engine keeps the pool it has -- searches beyond it wait
  // for a slot: speed, not results -- and does not try again by itself; pcgrl_solver_pool_slots reports it,
  // pcgrl_reserve_solver_pool retries on request)
  h->soko_grow_failed = true;
  h->soko_grow_msg = std::string("solver pool: growing from ") + std::to_string(h->soko_slots) + " to " + std::to_string(want) +
                     " slots failed: " + hipGetErrorString(e);
  return lazy ? PCGRL_OK : fail(PCGRL_EHIP, h->soko_grow_msg);
}

static void soko_pool_lazy(pcgrl_engine *h, Params &p, hipStream_t stream) {
  if (!h->seen_host || h->soko_slots >= sokoban_slots_for(h->p.n_envs)) return;
  if (*(volatile int32_t *)h->seen_host != 0) soko_pool_for(h, p, sokoban_slots_for(h->p.n_envs), stream);
}

static void choose_spread(pcgrl_engine *h, Params &p) {
  if (!h->seen_host || h->p.ext) return;
  static const int force = getenv("PCGRL_FORCE_SPREAD") ? atoi(getenv("PCGRL_FORCE_SPREAD")) : 0;  // development: 1 = spread, 2 = + helpers
  if (force) {
    p.spread = 1;
    p.sk_helpers = force == 2 ? 3 : 0;
    return;
  }
  // (a workgroup per env only pays while the launch is short of wavefronts: at 32 768 envs it is 32 768 workgroups of eight
  // waves and the step launch goes from 10 to 74 us)
  if (h->p.n_envs > 8192) return;
  const int32_t seen = *(volatile int32_t *)h->seen_host;
  if (seen != h->seen_last) {
    h->seen_last = seen;
    h->spread_left = 16;
  }
  if (h->spread_left > 0) {
    h->spread_left--;
    p.spread = 1;
    // ... and three more wavefronts per env run the A* stages next to the BFS stage, when their heaps fit the CU's 160 KiB
    // of LDS next to the observation staging rows (they do for every window up to about 2 x 62 cells per row)
    p.sk_helpers = h->lds_bytes + 3 * (size_t)SK_HELPER_LDS + 4096 <= 160 * 1024 ? 3 : 0;
  }
}

// ---------------------------------------------------------------------------------------------- C ABI
extern "C" {

const char *pcgrl_last_error(void) { return g_err.c_str(); }
const char *pcgrl_version(void) { return "pcgrl_amd 0.6.0 (gfx950)"; }

int pcgrl_create(const pcgrl_config *cfg, int32_t n_envs, int32_t device, pcgrl_handle *out) {
  if (!cfg || !out || n_envs < 1) return fail(PCGRL_EINVAL, "pcgrl_create: bad arguments");
  int lpe = 16, obs_chunks = 0, ndim = 0;
  int64_t obs_bytes = 0;
  int32_t shape[4] = {0, 0, 0, 0};
  int rc = validate(*cfg, lpe, obs_bytes, obs_chunks, shape, ndim);
  if (rc) return rc;
  ON_DEVICE(device);
  pcgrl_engine *e = new pcgrl_engine();
  {
    const char *f = getenv("PCGRL_ROLLOUT_KERNEL");
    if (f != nullptr && (f[0] == '0' || f[0] == '1')) e->rollout_form = f[0] - '0';
    if (getenv("PCGRL_OBS_NT_MB")) e->obs_nt_mb = atol(getenv("PCGRL_OBS_NT_MB"));
    if (getenv("PCGRL_ROLLOUT_EPW")) e->rollout_epw = atoi(getenv("PCGRL_ROLLOUT_EPW"));
  }
  e->device = device;
  e->lpe = lpe;
  e->obs_bytes = obs_bytes;
  e->obs_ndim = ndim;
  memcpy(e->obs_shape, shape, sizeof(shape));
  Params &p = e->p;
  memset(&p, 0, sizeof(p));
  p.cfg = *cfg;
  {  // integer form of the static targets (get_loss): statistics are far below 2^29 in magnitude, so infinite bounds clamp there
    bool integral = true;
    for (int k = 0; k < PCGRL_MAX_STATS; k++) {
      const double lo = cfg->trg_lo[k], hi = cfg->trg_hi[k];
      const int32_t big = 1 << 29;
      int32_t lo_i = 0, hi_i = 0;
      if (k < cfg->n_stats && cfg->has_trg[k]) {
        if (lo != lo || hi != hi || lo > hi) integral = false;  // (NaN, or an empty interval: the float64 form decides)
        else {
          if (lo <= -(double)big) lo_i = -big;
          else if (lo >= (double)big) lo_i = big;
          else if (lo == (double)(int32_t)lo) lo_i = (int32_t)lo;
          else integral = false;
          if (hi >= (double)big) hi_i = big;
          else if (hi <= -(double)big) hi_i = -big;
          else if (hi == (double)(int32_t)hi) hi_i = (int32_t)hi;
          else integral = false;
        }
      }
      p.trg_lo_i[k] = lo_i;
      p.trg_hi_i[k] = hi_i;
    }
    p.int_targets = integral ? 1 : 0;
  }
  p.n_envs = n_envs;
  p.n_tiles = n_tiles_of(cfg->problem);
  const bool is3d = cfg->problem == PCGRL_PROB_MC3DMAZE;
  p.n_cells = cfg->dims[0] * cfg->dims[1] * (is3d ? cfg->dims[2] : 1);
  p.obs_chunks = obs_chunks;
  if (is3d) m3_edge_masks_host(cfg->dims[1], cfg->dims[2], p.m3_notx0, p.m3_notxl);
  p.ext = (!is3d && (cfg->static_tiles || cfg->act_window[0] > 0)) ? 1 : 0;
  p.n_act = (!is3d && cfg->act_window[0] > 0) ? cfg->act_window[0] * cfg->act_window[1] : 1;
  // one padded observation row per lane + the OOB row (+ 2 rows per env of the wave with the static_builds plane)
  // (+ 64 bytes per env of the wave: the observe wave's copy of the RNG streams, see step_kernel)
  e->lds_bytes = (size_t)(obs_chunks * 16 + 32) * (65 + (cfg->static_tiles ? 16 : 0)) + 64 * 8;  // (+32: lds_row_stride may add a chunk)
  {
    // Cropped windows whose rows are whole 16-byte chunks, off the compile-time 16x16 / 32x32 point, without the static
    // tiles' extra channel: the general kernels compute the observation chunks from one code byte per cell
    // (encode_obs_codes) -- per env H rows of W + 20 bytes (rounded to 16) instead of H one-hot rows
    const int Hc = cfg->dims[0], Wc = cfg->dims[1], OWc = cfg->obs_window[1], Cc = p.n_tiles + 1;
    const bool fast_cfg = Hc == 16 && Wc == 16 && cfg->obs_window[0] == 32 && OWc == 32;
    p.obs16 = (!is3d && fast_cfg) ? 1 : 0;
    // non-temporal observation stores where a launch writes far more than the 256 MB last-level cache holds (store_obs16_nt)
    if ((p.obs16 || is3d) && obs_nt_for(e, (int64_t)n_envs * obs_bytes)) p.obs16 |= 2;  // (3-D: bit 1 alone)
    // (only where the one-hot rows are what limits occupancy: at 13 KB per workgroup -- binary 32 x 32 -- the rows are
    // cheaper: 14.0 vs 15.6 us per launch; at 26 KB -- binary 64 x 64 -- 67 vs 47.5; at 38 KB -- zelda 32 x 32 -- 44 vs 27.6)
    if (!is3d && cfg->representation != PCGRL_REP_WIDE && !cfg->static_tiles && !fast_cfg && (OWc * Cc) % 16 == 0 &&
        e->lds_bytes > 20000) {
      const int rs = (Wc + 8 + 12 + 15) & ~15;
      const int epw = 64 / lpe;
      p.obs_codes = rs;
      e->lds_bytes = (size_t)epw * Hc * rs + rs + (size_t)Cc * 48 + 64 * 8;
      e->lds_bytes = (e->lds_bytes + 15) & ~(size_t)15;
    }
  }
  p.lds_pair_bytes = (int32_t)e->lds_bytes;
  e->cpl = is3d ? (p.n_cells + 63) / 64 : 0;
  const int H = cfg->dims[0], W = cfg->dims[1];
  auto dalloc = [&](void **ptr, size_t bytes) -> hipError_t {
    hipError_t err = hipMalloc(ptr, bytes);
    if (err == hipSuccess) {
      e->allocs.push_back(*ptr);
      err = hipMemset(*ptr, 0, bytes);
    }
    return err;
  };
#define CREATE_CHK(x)                                                                                \
  do {                                                                                               \
    hipError_t _e = (x);                                                                             \
    if (_e != hipSuccess) {                                                                          \
      pcgrl_destroy(e);                                                                              \
      return fail(PCGRL_EHIP, std::string(#x) + ": " + hipGetErrorString(_e));                       \
    }                                                                                                \
  } while (0)
  if (is3d) {
    // one record per env: tile bits, overlay bits, column masks, cached start-plane results, move table (M3Lay)
    CREATE_CHK(dalloc(&p.planes, (size_t)n_envs * m3_layout(cfg->dims[0], cfg->dims[1], cfg->dims[2]).rec_words * sizeof(uint32_t)));
  } else
    CREATE_CHK(dalloc(&p.planes, (size_t)n_envs * ROW_WORDS * H * (W > 32 ? sizeof(uint64_t) : sizeof(uint32_t))));
  CREATE_CHK(dalloc((void **)&p.st, (size_t)n_envs * sizeof(EnvState)));
  CREATE_CHK(dalloc((void **)&p.rng, (size_t)n_envs * sizeof(RngState)));
  CREATE_CHK(dalloc((void **)&e->red, sizeof(RedScratch)));
  CREATE_CHK(dalloc((void **)&e->sample, sizeof(SampleState)));
  // [0..3] error flags; from int 64 on: per-workgroup phase-timing accumulators (PCGRL_PHASE_TIMING builds)
  CREATE_CHK(dalloc((void **)&p.err, sizeof(int32_t) * 64 + sizeof(uint64_t) * 8 * (size_t)(n_envs + 64)));
  std::vector<JumpEntry> jt = is3d ? make_jump_table(64, e->cpl, p.n_cells) : make_jump_table(H, W);
  JumpEntry *djt = nullptr;
  CREATE_CHK(dalloc((void **)&djt, jt.size() * sizeof(JumpEntry)));
  CREATE_CHK(hipMemcpy(djt, jt.data(), jt.size() * sizeof(JumpEntry), hipMemcpyHostToDevice));
  p.jump = djt;
  if (p.ext) {
    const size_t mbytes = W > 32 ? sizeof(uint64_t) : sizeof(uint32_t);
    const int nb = p.n_tiles <= 2 ? 1 : 3;
    CREATE_CHK(dalloc(&p.xplanes, (size_t)n_envs * (1 + nb) * H * mbytes));
    CREATE_CHK(dalloc((void **)&p.xstate, (size_t)n_envs * 4 * sizeof(uint32_t)));
    std::vector<JumpEntry> jb = make_jump_table(H + 2, W + 2);
    JumpEntry *djb = nullptr;
    CREATE_CHK(dalloc((void **)&djb, jb.size() * sizeof(JumpEntry)));
    CREATE_CHK(hipMemcpy(djb, jb.data(), jb.size() * sizeof(JumpEntry), hipMemcpyHostToDevice));
    p.jump_b = djb;
  }
  if (cfg->problem == PCGRL_PROB_SOKOBAN) {
    CREATE_CHK(sokoban_alloc(p, e->allocs, std::min(SK_SLOTS_AT_CREATE, sokoban_slots_for(n_envs)), &e->soko_slots));
    CREATE_CHK(hipHostMalloc((void **)&e->seen_host, sizeof(int32_t), hipHostMallocMapped));
    *e->seen_host = 0;
    CREATE_CHK(hipHostGetDevicePointer((void **)&p.solver_seen, e->seen_host, 0));
  }
  if (cfg->n_ctrl > 0) {  // controllable mode: per-env targets, initialised with the static ones
    std::vector<double> init((size_t)n_envs * PCGRL_MAX_STATS * 2, 0.0);
    for (int i = 0; i < n_envs; i++)
      for (int k = 0; k < PCGRL_MAX_STATS; k++) {
        init[((size_t)i * PCGRL_MAX_STATS + k) * 2] = cfg->trg_lo[k];
        init[((size_t)i * PCGRL_MAX_STATS + k) * 2 + 1] = cfg->trg_hi[k];
      }
    // (the engine-wide resampling record sits right in front of the targets: TrgResample, pcgrl_common.h)
    uint8_t *trg_block = nullptr;
    CREATE_CHK(dalloc((void **)&trg_block, sizeof(TrgResample) + init.size() * sizeof(double)));
    p.trg = (double *)(trg_block + sizeof(TrgResample));
    CREATE_CHK(dalloc((void **)&p.trg_pending, init.size() * sizeof(double)));
    CREATE_CHK(dalloc((void **)&p.trg_flag, (size_t)n_envs * sizeof(int32_t)));
    CREATE_CHK(hipMemcpy(p.trg, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
  }
#undef CREATE_CHK
  {
    const size_t plane_bytes = is3d ? (size_t)m3_layout(cfg->dims[0], cfg->dims[1], cfg->dims[2]).rec_words * sizeof(uint32_t)
                                    : (size_t)ROW_WORDS * H * (W > 32 ? sizeof(uint64_t) : sizeof(uint32_t));
    e->state_arrays.push_back({p.planes, plane_bytes});
    e->state_arrays.push_back({p.st, sizeof(EnvState)});
    e->state_arrays.push_back({p.rng, sizeof(RngState)});
    if (p.xplanes) {
      const size_t mbytes = W > 32 ? sizeof(uint64_t) : sizeof(uint32_t);
      e->state_arrays.push_back({p.xplanes, (size_t)(1 + (p.n_tiles <= 2 ? 1 : 3)) * H * mbytes});
      e->state_arrays.push_back({p.xstate, 4 * sizeof(uint32_t)});
    }
    if (p.trg) {
      e->state_arrays.push_back({p.trg, PCGRL_MAX_STATS * 2 * sizeof(double)});
      e->state_arrays.push_back({p.trg_pending, PCGRL_MAX_STATS * 2 * sizeof(double)});
      e->state_arrays.push_back({p.trg_flag, sizeof(int32_t)});
    }
  }
  // the two-kernel rollout exists for the compile-time 16x16 kernels in plain mode (no wrappers, no control metrics)
  if (!is3d && H == 16 && W == 16 && !p.ext && cfg->n_ctrl == 0 &&
      (cfg->representation == PCGRL_REP_WIDE || (cfg->obs_window[0] == 32 && cfg->obs_window[1] == 32))) {
    hipError_t he = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming);
    if (he == hipSuccess) he = dalloc(&e->snap_planes, (size_t)n_envs * ROW_WORDS * H * sizeof(uint32_t));
    if (he == hipSuccess) he = dalloc((void **)&e->snap_st, (size_t)n_envs * sizeof(EnvState));
    if (he == hipSuccess) he = dalloc((void **)&e->snap_rng, (size_t)n_envs * sizeof(RngState));
    if (he != hipSuccess) {
      pcgrl_destroy(e);
      return fail(PCGRL_EHIP, std::string("pcgrl_create (rollout side stream / snapshot): ") + hipGetErrorString(he));
    }
    e->split_ok = true;
    int cus = 0;
    rollout_resident_per_cu = 0;
    if (launch(K_ROLLOUT_RESIDENT, e->lpe, e->p, e->lds_bytes, nullptr, e->cpl) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess)
      e->two_role_resident = (int64_t)rollout_resident_per_cu * cus;
    // (development / tests: PCGRL_ROLLOUT_RESIDENT=<workgroups> stands in for the device's figure, so that small batches take
    // the forms of large ones)
    if (getenv("PCGRL_ROLLOUT_RESIDENT")) e->two_role_resident = atol(getenv("PCGRL_ROLLOUT_RESIDENT"));
    (void)hipGetLastError();
  }
  {
    const hipError_t he = hipHostMalloc((void **)&e->hdr_host, 256, hipHostMallocDefault);
    if (he != hipSuccess) {
      pcgrl_destroy(e);
      return fail(PCGRL_EHIP, std::string("hipHostMalloc (state header): ") + hipGetErrorString(he));
    }
    memset(e->hdr_host, 0, 256);
    state_header_build(e);
  }
  *out = e;
  // default seeding: env i gets seed i (callers normally call pcgrl_seed)
  std::vector<uint64_t> seeds(n_envs);
  for (int i = 0; i < n_envs; i++) seeds[i] = (uint64_t)i;
  return pcgrl_seed(e, seeds.data());
}

void pcgrl_destroy(pcgrl_handle h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  if (h->seen_host) (void)hipHostFree(h->seen_host);
  if (h->hdr_host) (void)hipHostFree(h->hdr_host);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->side) (void)hipStreamDestroy(h->side);
  for (void *a : h->allocs) (void)hipFree(a);
  delete h;
}

int pcgrl_seed(pcgrl_handle h, const uint64_t *seeds) {
  if (!h || !seeds) return fail(PCGRL_EINVAL, "pcgrl_seed: bad arguments");
  ON_DEVICE(h->device);
  std::vector<RngState> r(h->p.n_envs);
  for (int i = 0; i < h->p.n_envs; i++) {
    pcg64_seed_state(seeds[i], r[i].rep);
    memcpy(r[i].prob, r[i].rep, sizeof(r[i].rep));  // envs/pcgrl_env.py:142-146: same seed for both streams
  }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h->p.rng, r.data(), r.size() * sizeof(RngState), hipMemcpyHostToDevice));
  // seeding a numpy bit generator drops the spare 32 bits of its last draw
  if (h->p.xstate) HIPCHK(hipMemset(h->p.xstate, 0, (size_t)h->p.n_envs * 4 * sizeof(uint32_t)));
  return PCGRL_OK;
}

int pcgrl_reset(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_init_grids, const int32_t *d_init_pos, void *stream) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_reset: null handle");
  ON_DEVICE(h->device);
  Params p = h->p;
  soko_pool_lazy(h, p, (hipStream_t)stream);
  p.mask = d_mask;
  p.init_grids = d_init_grids;
  p.init_pos = d_init_pos;
  p.sk_budget = h->sk_budget;  // (asynchronous stepping: a playable level's search may stay parked, the env busy)
  HIPCHK(launch(K_RESET, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  if (d_mask == nullptr) h->maybe_stale = false;
  return PCGRL_OK;
}

#define NOT_WITH_BUDGET(h, what)                                                                                         \
  if ((h)->sk_budget > 0)                                                                                                \
  return fail(PCGRL_EINVAL, std::string(what) + ": a solver budget is set (pcgrl_set_solver_budget): envs may be busy, which only " \
                                                "pcgrl_step_ready can report")

int pcgrl_step(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward, uint8_t *d_done,
               int32_t *d_stats, void *stream) {
  if (!h || !d_actions) return fail(PCGRL_EINVAL, "pcgrl_step: bad arguments");
  NOT_WITH_BUDGET(h, "pcgrl_step");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.no_fast = h->maybe_stale ? 1 : 0;
  soko_pool_lazy(h, p, (hipStream_t)stream);
  choose_spread(h, p);
  p.actions = d_actions;
  p.auto_reset = auto_reset;
  p.obs = d_obs;
  p.reward = d_reward;
  p.done = d_done;
  p.stats_out = d_stats;
  HIPCHK(launch(K_STEP, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_step_seq(pcgrl_handle h, const int32_t *d_action_rows, int64_t row_stride, int32_t n_rows, int32_t first_row,
                   int32_t n_steps, int32_t auto_reset, uint8_t *d_obs, float *d_reward, uint8_t *d_done, int32_t *d_stats,
                   void *stream) {
  if (!h || !d_action_rows || n_rows < 1 || n_steps < 0 || first_row < 0 || row_stride < 0)
    return fail(PCGRL_EINVAL, "pcgrl_step_seq: bad arguments");
  for (int32_t k = 0; k < n_steps; k++) {
    const int rc = pcgrl_step(h, d_action_rows + (size_t)((first_row + k) % n_rows) * (size_t)row_stride, auto_reset, d_obs, d_reward,
                              d_done, d_stats, stream);
    if (rc != PCGRL_OK) return rc;
  }
  return PCGRL_OK;
}

int pcgrl_step_ex(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward,
                  double *d_reward64, uint8_t *d_done, int32_t *d_stats, float *d_ctrl_obs, void *stream) {
  if (!h || !d_actions) return fail(PCGRL_EINVAL, "pcgrl_step_ex: bad arguments");
  NOT_WITH_BUDGET(h, "pcgrl_step_ex");
  ON_DEVICE(h->device);
  if (d_ctrl_obs && h->p.cfg.n_ctrl == 0) return fail(PCGRL_EINVAL, "pcgrl_step_ex: d_ctrl_obs needs cfg.n_ctrl > 0");
  Params p = h->p;
  p.no_fast = h->maybe_stale ? 1 : 0;
  soko_pool_lazy(h, p, (hipStream_t)stream);
  choose_spread(h, p);
  p.actions = d_actions;
  p.auto_reset = auto_reset;
  p.obs = d_obs;
  p.reward = d_reward;
  p.reward64 = d_reward64;
  p.done = d_done;
  p.stats_out = d_stats;
  p.ctrl_obs = d_ctrl_obs;
  HIPCHK(launch(K_STEP, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

// pcgrl_rollout where the rollout kernel loses to stepping.  That kernel keeps the env in registers and does both roles on
// one wave: it pays where a step is short or its launches wait for their slowest env (16-row maps: 3.5 vs 5.9 us per step
// for binary; the 3-D mazes, whose waves then advance independently: 10 vs 21 us at 7^3, 112 vs 126 at 15^3 over the same
// steps) and costs on the 2-D maps of more than 16 rows, where the observation is large (or computed chunk by chunk from tile
// codes, Params::obs_codes) and wants a wave of its own next to the statistics -- zelda_big 49.7 vs 34 us per step,
// zelda_bigger 217 vs 126, binary_bigger 60 vs 51, binary_big 16.7 vs 15.5 (bench lines of rounds 4 / 5).  There the call
// issues its n_steps as step launches: same results by the entry point's own definition.
static bool rollout_as_steps(const pcgrl_engine *h) {
  if (h->rollout_form >= 0) return h->rollout_form == 0;  // (pcgrl_set_rollout_form; PCGRL_ROLLOUT_KERNEL at pcgrl_create)
  // (maps of more than 16 rows without tile codes, e.g. binary_big 32 x 32: 16.7 us per step in one launch -- its kernel sits at
  // 256 VGPRs -- against 15.0-15.5 as step launches)
  // lpe > 16: a 2-D map of more than 16 rows OR more than 32 columns (64-bit row masks run on 32 / 64 lanes per env)
  return h->p.cfg.problem != PCGRL_PROB_MC3DMAZE && (h->p.obs_codes > 0 || h->lpe > 16);
}

int32_t pcgrl_rollout_is_one_launch(pcgrl_handle h) { return h ? (rollout_as_steps(h) ? 0 : 1) : -1; }  // (1 also for the two-kernel form)

int pcgrl_set_rollout_form(pcgrl_handle h, int32_t form) {
  if (!h || form < -1 || form > 2)
    return fail(PCGRL_EINVAL, "pcgrl_set_rollout_form: form must be -1 (by shape), 0 (step launches), 1 (one kernel) or 2 (two kernels)");
  if (form == 2 && !h->split_ok)
    return fail(PCGRL_EUNSUPPORTED, "pcgrl_set_rollout_form(2): the two-kernel rollout exists for 16 x 16 maps with the default window, plain mode");
  h->rollout_form = form;
  return PCGRL_OK;
}

int pcgrl_rollout_ex(pcgrl_handle h, const int32_t *d_actions, int32_t n_steps, int32_t auto_reset, uint8_t *d_obs,
                     int32_t obs_last_only, float *d_reward, double *d_reward64, uint8_t *d_done, int32_t *d_stats, float *d_ctrl_obs,
                     void *stream) {
  if (!h || !d_actions || n_steps < 1) return fail(PCGRL_EINVAL, "pcgrl_rollout: bad arguments");
  NOT_WITH_BUDGET(h, "pcgrl_rollout");
  if (d_ctrl_obs && h->p.cfg.n_ctrl == 0) return fail(PCGRL_EINVAL, "pcgrl_rollout_ex: d_ctrl_obs needs cfg.n_ctrl > 0");
  if (rollout_as_steps(h)) {
    const size_t N = (size_t)h->p.n_envs;
    for (int32_t k = 0; k < n_steps; k++) {
      const bool last = k == n_steps - 1;
      uint8_t *obs_k = d_obs == nullptr ? nullptr : (obs_last_only ? (last ? d_obs : nullptr) : d_obs + (size_t)k * N * (size_t)h->obs_bytes);
      const int rc = pcgrl_step_ex(h, d_actions + (size_t)k * N * (size_t)h->p.n_act, auto_reset, obs_k, d_reward ? d_reward + (size_t)k * N : nullptr,
                                   d_reward64 ? d_reward64 + (size_t)k * N : nullptr, d_done ? d_done + (size_t)k * N : nullptr,
                                   d_stats ? d_stats + (size_t)k * N * (size_t)h->p.cfg.n_stats : nullptr, last ? d_ctrl_obs : nullptr, stream);
      if (rc != PCGRL_OK) return rc;
    }
    return PCGRL_OK;
  }
  ON_DEVICE(h->device);
  Params p = h->p;
  p.no_fast = h->maybe_stale ? 1 : 0;
  soko_pool_lazy(h, p, (hipStream_t)stream);
  p.actions = d_actions;
  p.n_steps = n_steps;
  p.spread = 0;
  // The roles as kernels of their own (16 x 16 compile-time kernels, plain mode; not while statistics may be stale and not in the
  // controllable-output form), see rollout_kernel's ROLE.  The simulate kernel takes 103 / 114 / 157 VGPRs (binary / zelda /
  // sokoban) against the two-role kernel's 224 / 323 / 335, whose workgroups are all resident at once only up to
  // two_role_resident of them (binary 1024 = 4096 envs, zelda and sokoban 512 = 2048 envs); beyond that it runs in rounds and the
  // roles are better off apart.  By shape (form -1), measured in profiles/r06_dev_traces.md section 3:
  //   no observation asked for    the simulate kernel alone, always (never slower; binary-narrow 16 384 envs 9.3 -> 6.1 us per
  //                               step, zelda-turtle 4096 envs 4.4 -> 2.1)
  //   only the last observation   when the two-role kernel would run in rounds: the simulate kernel, then pcgrl_observe's kernel
  //                               on the state it left, same stream (zelda-turtle 4096 envs 4.2 -> 2.2; one launch more per call,
  //                               which costs binary-narrow at 4096 envs 4.04 -> 4.32 at 8 steps per call: not below the bound)
  //   every observation           when the two-role kernel would run in rounds and the call has >= 16 steps: both kernels, the
  //                               observe one on the engine's side stream from a snapshot of the pre-call state (form 2).
  //                               Snapshot + fork + join cost 15-25 us per CALL (binary-narrow 4096 envs, 8 steps: 5.97 vs 4.10 us
  //                               per step; 64 steps: 3.42 vs 3.21), the gain is the simulate role's occupancy (binary-narrow
  //                               8192 envs 7.86 -> 5.87, zelda-turtle 4096 envs 8.64 -> 7.51, 16 384 envs 33.4 -> 28.4).
  // Fewer envs per simulate wave (PCGRL_ROLLOUT_EPW) do NOT pay: a wave64 instruction occupies its SIMD for four cycles whatever
  // the number of active lanes, and two waves per SIMD already issue back to back (binary-narrow, 1 / 2 envs per wave: 4.59 / 3.54).
  const bool roles_ok = h->split_ok && !h->maybe_stale && d_reward64 == nullptr && d_ctrl_obs == nullptr && h->rollout_form != 1;
  const int64_t two_role_wgs = ((int64_t)h->p.n_envs + 64 / h->lpe - 1) / (64 / h->lpe);
  const bool in_rounds = h->two_role_resident > 0 && two_role_wgs > h->two_role_resident;
  const bool split = roles_ok && d_obs != nullptr && (h->rollout_form == 2 || (in_rounds && !obs_last_only && n_steps >= 16));
  const bool sim_then_observe = roles_ok && !split && d_obs != nullptr && obs_last_only && in_rounds;
  const bool sim_only = (roles_ok && d_obs == nullptr) || sim_then_observe;
  if (h->p.cfg.problem != PCGRL_PROB_MC3DMAZE) {
    const int full = 64 / h->lpe;
    int epw = h->rollout_epw;
    if (epw > full) epw = full;
    p.spread = epw > 0 ? epw : 0;
  }
  p.auto_reset = auto_reset;
  p.obs = d_obs;
  p.obs_last_only = obs_last_only;
  p.obs_env_bytes = h->obs_bytes;
  // (a rollout that keeps every step's observation writes n_steps times a step launch's bytes: 805 MB for 64 steps of 4096 binary envs)
  if ((p.obs16 || h->p.cfg.problem == PCGRL_PROB_MC3DMAZE) && d_obs && !obs_last_only && obs_nt_for(h, (int64_t)n_steps * h->p.n_envs * h->obs_bytes)) p.obs16 |= 2;
  p.reward = d_reward;
  p.reward64 = d_reward64;
  p.done = d_done;
  p.stats_out = d_stats;
  p.ctrl_obs = d_ctrl_obs;
  if (split || sim_only) {
    hipStream_t s0 = (hipStream_t)stream;
    if (split) {  // the observe kernel: from a snapshot of the pre-call state, on the side stream
      const int64_t plane_q = (int64_t)h->p.n_envs * ROW_WORDS * h->p.cfg.dims[0] * (int64_t)sizeof(uint32_t) / 16;
      const int64_t total = plane_q + (int64_t)h->p.n_envs * 2 + (int64_t)h->p.n_envs * (int64_t)(sizeof(RngState) / 16);
      hipLaunchKernelGGL(rollout_snapshot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s0, (const uint4 *)h->p.planes,
                         (uint4 *)h->snap_planes, plane_q, (const uint4 *)h->p.st, (uint4 *)h->snap_st, (const uint4 *)h->p.rng, (uint4 *)h->snap_rng,
                         h->p.n_envs);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(h->ev_fork, s0));
      HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      Params po = p;
      po.planes = h->snap_planes;
      po.st = h->snap_st;
      po.rng = h->snap_rng;
      po.spread = 0;
      po.reward = nullptr;
      po.done = nullptr;
      po.stats_out = nullptr;
      HIPCHK(launch(K_ROLLOUT_OBS, h->lpe, po, h->lds_bytes, h->side, h->cpl));
    }
    Params ps = p;
    ps.obs = nullptr;
    HIPCHK(launch(K_ROLLOUT_SIM, h->lpe, ps, 0, s0, h->cpl));
    if (split) {
      HIPCHK(hipEventRecord(h->ev_join, h->side));
      HIPCHK(hipStreamWaitEvent(s0, h->ev_join, 0));
    }
    if (sim_then_observe) {
      Params po = h->p;
      po.obs = d_obs;
      HIPCHK(launch(K_OBSERVE, h->lpe, po, h->lds_bytes, s0, h->cpl));
    }
    return PCGRL_OK;
  }
  HIPCHK(launch(K_ROLLOUT, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_rollout(pcgrl_handle h, const int32_t *d_actions, int32_t n_steps, int32_t auto_reset, uint8_t *d_obs,
                  int32_t obs_last_only, float *d_reward, uint8_t *d_done, int32_t *d_stats, void *stream) {
  return pcgrl_rollout_ex(h, d_actions, n_steps, auto_reset, d_obs, obs_last_only, d_reward, nullptr, d_done, d_stats, nullptr, stream);
}

int pcgrl_update(pcgrl_handle h, const int32_t *d_actions, uint8_t *d_obs, void *stream) {
  if (!h || !d_actions) return fail(PCGRL_EINVAL, "pcgrl_update: bad arguments");
  NOT_WITH_BUDGET(h, "pcgrl_update");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.actions = d_actions;
  p.obs = d_obs;
  p.update_only = 1;
  h->maybe_stale = true;
  HIPCHK(launch(K_STEP, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_refresh_stats(pcgrl_handle h, int32_t *d_stats, void *stream) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_refresh_stats: null handle");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.refresh_only = 1;
  p.stats_out = d_stats;
  p.sk_budget = h->sk_budget;
  HIPCHK(launch(K_RESET, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  h->maybe_stale = false;
  return PCGRL_OK;
}

int pcgrl_queue_targets(pcgrl_handle h, const uint8_t *d_mask, const double *d_trg_lo, const double *d_trg_hi, void *stream) {
  if (!h || !d_trg_lo || !d_trg_hi) return fail(PCGRL_EINVAL, "pcgrl_queue_targets: bad arguments");
  ON_DEVICE(h->device);
  if (h->p.cfg.n_ctrl == 0) return fail(PCGRL_EINVAL, "pcgrl_queue_targets: the engine was created without control metrics");
  Params p = h->p;
  p.mask = d_mask;
  hipLaunchKernelGGL(queue_targets_kernel, dim3((p.n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, p, d_trg_lo, d_trg_hi);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_set_target_resampling(pcgrl_handle h, int32_t enable, uint64_t seed, const double *lo, const double *hi) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_set_target_resampling: null handle");
  if (h->p.cfg.n_ctrl == 0) return fail(PCGRL_EINVAL, "pcgrl_set_target_resampling: the engine was created without control metrics");
  if (enable && (!lo || !hi)) return fail(PCGRL_EINVAL, "pcgrl_set_target_resampling: bounds missing");
  ON_DEVICE(h->device);
  TrgResample r;
  memset(&r, 0, sizeof(r));
  r.enable = enable ? 1 : 0;
  r.seed = seed;
  for (int j = 0; j < h->p.cfg.n_ctrl && enable; j++) {
    if (!(lo[j] <= hi[j])) return fail(PCGRL_EINVAL, "pcgrl_set_target_resampling: need lo <= hi for every control metric");
    r.lo[j] = lo[j];
    r.hi[j] = hi[j];
  }
  HIPCHK(hipDeviceSynchronize());  // (launches in flight keep the parameters they were issued under)
  HIPCHK(hipMemcpy((TrgResample *)h->p.trg - 1, &r, sizeof(r), hipMemcpyHostToDevice));
  return PCGRL_OK;
}

int pcgrl_ctrl_observe(pcgrl_handle h, float *d_ctrl_obs, void *stream) {
  if (!h || !d_ctrl_obs) return fail(PCGRL_EINVAL, "pcgrl_ctrl_observe: bad arguments");
  ON_DEVICE(h->device);
  if (h->p.cfg.n_ctrl == 0) return fail(PCGRL_EINVAL, "pcgrl_ctrl_observe: the engine was created without control metrics");
  Params p = h->p;
  p.ctrl_obs = d_ctrl_obs;
  dim3 grid((p.n_envs + 63) / 64), block(64);
  switch (p.cfg.n_stats) {
    case 2: hipLaunchKernelGGL((ctrl_observe_kernel<2>), grid, block, 0, (hipStream_t)stream, p); break;
    case 3: hipLaunchKernelGGL((ctrl_observe_kernel<3>), grid, block, 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL((ctrl_observe_kernel<7>), grid, block, 0, (hipStream_t)stream, p); break;
  }
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_observe(pcgrl_handle h, uint8_t *d_obs, void *stream) {
  if (!h || !d_obs) return fail(PCGRL_EINVAL, "pcgrl_observe: bad arguments");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.obs = d_obs;
  HIPCHK(launch(K_OBSERVE, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_get_static(pcgrl_handle h, uint8_t *d_static, void *stream) {
  if (!h || !d_static) return fail(PCGRL_EINVAL, "pcgrl_get_static: bad arguments");
  ON_DEVICE(h->device);
  if (!h->p.cfg.static_tiles) return fail(PCGRL_EINVAL, "pcgrl_get_static: the engine was created without static tiles");
  Params p = h->p;
  p.out_static = d_static;
  const int nb = p.n_tiles <= 2 ? 1 : 3;
  if (p.cfg.dims[1] > 32)
    hipLaunchKernelGGL((get_static_kernel<uint64_t>), dim3(p.n_envs), dim3(64), 0, (hipStream_t)stream, p, nb);
  else
    hipLaunchKernelGGL((get_static_kernel<uint32_t>), dim3(p.n_envs), dim3(64), 0, (hipStream_t)stream, p, nb);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_set_static(pcgrl_handle h, double static_prob, int32_t n_static_walls, int32_t eval_mode) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_set_static: null handle");
  if (!h->p.cfg.static_tiles) return fail(PCGRL_EINVAL, "pcgrl_set_static: the engine was created without static tiles");
  if (static_prob > 1.0) return fail(PCGRL_EINVAL, "pcgrl_set_static: static_prob must be <= 1");
  if (n_static_walls > 0 && (h->p.cfg.dims[0] < 3 || h->p.cfg.dims[1] < 3))
    return fail(PCGRL_EINVAL, "pcgrl_set_static: static walls need a map of at least 3x3");
  if (static_prob >= 0.0) h->p.cfg.static_prob = static_prob;
  if (n_static_walls >= 0) h->p.cfg.n_static_walls = n_static_walls;
  if (eval_mode >= 0) h->p.cfg.static_eval = eval_mode ? 1 : 0;
  state_header_build(h);
  return PCGRL_OK;
}

int64_t pcgrl_obs_bytes(pcgrl_handle h) { return h ? h->obs_bytes : -1; }

int pcgrl_obs_shape(pcgrl_handle h, int32_t shape_out[4], int32_t *ndim_out) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_obs_shape: null handle");
  memcpy(shape_out, h->obs_shape, sizeof(h->obs_shape));
  *ndim_out = h->obs_ndim;
  return PCGRL_OK;
}

int pcgrl_get_state(pcgrl_handle h, uint8_t *d_grids, int32_t *d_pos, int32_t *d_counters, int32_t *d_stats, double *d_last_loss,
                    double *d_ep_return, void *stream) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_get_state: null handle");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.out_grids = d_grids;
  p.out_pos = d_pos;
  p.out_counters = d_counters;
  p.stats_out = d_stats;
  p.out_last_loss = d_last_loss;
  p.out_ep_return = d_ep_return;
  HIPCHK(launch(K_GET_STATE, h->lpe, p, 0, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_get_last_episode(pcgrl_handle h, double *d_ep_return, int32_t *d_ep_len, int32_t *d_final_stats, int64_t *d_n_episodes,
                           void *stream) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_get_last_episode: null handle");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.out_ep_return = d_ep_return;
  p.out_ep_len = d_ep_len;
  p.stats_out = d_final_stats;
  p.out_n_episodes = d_n_episodes;
  HIPCHK(launch(K_LAST_EPISODE, h->lpe, p, 0, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

// Scratch engines of the handle-less pcgrl_stats_for_grids: one per (problem, map shape, solver_power, device), created
// on first use and kept until pcgrl_stats_cache_clear() / process exit.  The statistics kernel needs no per-env state,
// only the engine's error word and (sokoban) solver workspace pool, so one engine serves any batch size.
static std::mutex g_stats_mu;
static std::map<std::tuple<int, int, int, int, int, int>, pcgrl_handle> g_stats_engines;

static int stats_engine_for(const pcgrl_config &cfg, int device, pcgrl_handle *out) {
  pcgrl_config c = cfg;
  c.representation = PCGRL_REP_NARROW;
  c.obs_window[0] = 2 * c.dims[0];
  c.obs_window[1] = c.ndim == 3 ? 2 * c.dims[1] : 32;  // unused by the statistics; any legal value
  c.obs_window[2] = c.ndim == 3 ? 2 * c.dims[2] : 1;
  c.n_ctrl = 0;
  c.static_tiles = 0;
  c.static_prob = 0.0;
  c.n_static_walls = 0;
  c.act_window[0] = c.act_window[1] = c.act_window[2] = 0;
  const auto key = std::make_tuple((int)c.problem, (int)c.dims[0], (int)c.dims[1], (int)(c.ndim == 3 ? c.dims[2] : 1),
                                   (int)c.solver_power, device);
  std::lock_guard<std::mutex> lock(g_stats_mu);
  auto it = g_stats_engines.find(key);
  if (it == g_stats_engines.end()) {
    pcgrl_handle h = nullptr;
    int rc = pcgrl_create(&c, 256, device, &h);  // 256 "envs": sizes the sokoban solver's workspace pool (64 slots)
    if (rc) return rc;
    it = g_stats_engines.emplace(key, h).first;
  }
  *out = it->second;
  return PCGRL_OK;
}

int pcgrl_stats_for_grids_h(pcgrl_handle h, int32_t n, const uint8_t *d_grids, int32_t *d_stats, void *stream) {
  if (!h || n < 1 || !d_grids || !d_stats) return fail(PCGRL_EINVAL, "pcgrl_stats_for_grids_h: bad arguments");
  ON_DEVICE(h->device);
  Params p;
  {
    // One level per workgroup, at most one such workgroup per CU: up to 256 searches run at once, each in a workspace
    // slot.  A batch that is much larger than the pool gets a larger pool the first time (synchronous, up to 256 slots =
    // 11.8 GB at the default solver_power; the old pool stays allocated until destroy: launches in flight on other streams
    // keep using it).  The handle-less entry point shares its hidden engines between threads: the pool is grown and the
    // launch parameters are read under one lock.
    static std::mutex grow_mu;
    std::lock_guard<std::mutex> lock(grow_mu);
    p = h->p;
    if (h->p.soko) soko_pool_for(h, p, std::min(256, sokoban_slots_for(n)), (hipStream_t)stream);
  }
  p.n_envs = n;  // the kernel touches no per-env engine state
  p.init_grids = d_grids;
  p.stats_out = d_stats;
  p.sk_helpers = h->p.soko ? 3 : 0;  // sokoban: one map per workgroup, the solver's stages side by side
  HIPCHK(launch(K_STATS_FOR_GRIDS, h->lpe, p, 0, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_stats_for_grids(const pcgrl_config *cfg, int32_t n, const uint8_t *d_grids, int32_t *d_stats, int32_t device, void *stream) {
  if (!cfg || n < 1 || !d_grids || !d_stats) return fail(PCGRL_EINVAL, "pcgrl_stats_for_grids: bad arguments");
  pcgrl_handle h = nullptr;
  int rc = stats_engine_for(*cfg, device, &h);
  if (rc) return rc;
  return pcgrl_stats_for_grids_h(h, n, d_grids, d_stats, stream);
}

int pcgrl_stats_poll_error(const pcgrl_config *cfg, int32_t device) {
  if (!cfg) return fail(PCGRL_EINVAL, "pcgrl_stats_poll_error: bad arguments");
  pcgrl_handle h = nullptr;
  int rc = stats_engine_for(*cfg, device, &h);
  if (rc) return rc;
  return pcgrl_poll_error(h);
}

void pcgrl_stats_cache_clear(void) {
  std::lock_guard<std::mutex> lock(g_stats_mu);
  for (auto &kv : g_stats_engines) pcgrl_destroy(kv.second);
  g_stats_engines.clear();
}

int pcgrl_reduce_episodes(pcgrl_handle h, double *d_out, int32_t clear, void *stream) {
  if (!h || !d_out) return fail(PCGRL_EINVAL, "pcgrl_reduce_episodes: bad arguments");
  ON_DEVICE(h->device);
  if (h->p.n_envs <= 16384) {
    hipLaunchKernelGGL(reduce_episodes_small_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, h->p, d_out, clear ? 1 : 0);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
  }
  const int n_blocks = std::min(RED_BLOCKS, (h->p.n_envs + RED_THREADS - 1) / RED_THREADS);
  hipLaunchKernelGGL(reduce_episodes_kernel, dim3(n_blocks), dim3(RED_THREADS), 0, (hipStream_t)stream, h->p, h->red, d_out, n_blocks,
                     clear ? 1 : 0);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_set_state(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_grids, const int32_t *d_pos, const int32_t *d_counters,
                    const double *d_ep_return, void *stream) {
  if (!h || !d_grids) return fail(PCGRL_EINVAL, "pcgrl_set_state: bad arguments");
  if (h->p.ext) return fail(PCGRL_EUNSUPPORTED, "pcgrl_set_state: not with static tiles / action patches (their masks are drawn at reset)");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.mask = d_mask;
  p.init_grids = d_grids;
  p.init_pos = d_pos;
  p.in_counters = d_counters;
  p.in_ep_return = d_ep_return;
  p.set_state = 1;
  p.sk_budget = h->sk_budget;
  HIPCHK(launch(K_RESET, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  if (d_mask == nullptr) h->maybe_stale = false;
  return PCGRL_OK;
}

static size_t state_section(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// The image starts with one 256-byte header section: what it was exported from.  pcgrl_import_state refuses an image whose
// header does not match the importing engine (another config -- even one with the same byte size, e.g. other weights or
// targets --, another batch size, another per-env layout / library version) instead of importing it silently.
struct StateHeader {
  uint64_t magic;        // "PCGRLST2"
  uint64_t fingerprint;  // FNV-1a over the create-time config fields, n_envs, the library version string and every array's bytes per env
  int32_t n_envs, n_arrays;
  uint64_t total_bytes;
  // the three static-tile parameters pcgrl_set_static changes at run time (the reference's set_static_prob /
  // set_n_static_walls / set_eval_mode, reps/wrappers.py:256-263): state, not identity -- a full import applies them
  double static_prob;
  int32_t n_static_walls, static_eval;
};
static constexpr uint64_t STATE_MAGIC = 0x3254534c52474350ull;  // "PCGRLST2", little endian
static constexpr size_t STATE_HDR_BYTES = 256;
static_assert(sizeof(StateHeader) <= STATE_HDR_BYTES, "state header section");

static uint64_t state_fingerprint(pcgrl_handle h) {
  uint64_t x = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    for (size_t i = 0; i < n; i++) x = (x ^ ((const uint8_t *)p)[i]) * 1099511628211ull;
  };
  const pcgrl_config &c = h->p.cfg;  // field by field: the struct's padding bytes are not part of the config
#define MIX(f) mix(&c.f, sizeof(c.f))
  MIX(problem); MIX(representation); MIX(ndim); MIX(dims); MIX(obs_window); MIX(max_iterations); MIX(max_changes); MIX(n_stats);
  MIX(has_trg); MIX(weights); MIX(trg_lo); MIX(trg_hi); MIX(solver_power); MIX(n_ctrl); MIX(ctrl_idx); MIX(ctrl_range);
  MIX(act_window); MIX(static_tiles);  // (not static_prob / n_static_walls / static_eval: pcgrl_set_static moves them)
#undef MIX
  const int32_t n = h->p.n_envs;
  mix(&n, sizeof(n));
  const char *v = pcgrl_version();
  mix(v, strlen(v));
  for (auto &a : h->state_arrays) mix(&a.second, sizeof(a.second));
  return x;
}

// The header lives in pinned host memory owned by the engine: pcgrl_export_state copies it from there, so a captured export
// keeps a pointer that stays valid and every replay writes the header of the engine's CURRENT static-tile parameters.
static void state_header_build(pcgrl_handle h) {
  if (!h->hdr_host) return;
  StateHeader hdr{};
  hdr.magic = STATE_MAGIC;
  hdr.fingerprint = state_fingerprint(h);
  hdr.n_envs = h->p.n_envs;
  hdr.n_arrays = (int32_t)h->state_arrays.size();
  hdr.total_bytes = (uint64_t)pcgrl_state_bytes(h);
  hdr.static_prob = h->p.cfg.static_prob;
  hdr.n_static_walls = h->p.cfg.n_static_walls;
  hdr.static_eval = h->p.cfg.static_eval;
  // (no memset: the bytes behind the struct were zeroed once at create, and an export issued earlier may still be copying
  // from this pinned block -- magic / fingerprint / sizes are rewritten with the values they already hold, only the three
  // static-tile fields really change.  pcgrl_set_static must still be ordered after exports that are to carry the OLD
  // parameters: include/pcgrl_amd.h)
  memcpy(h->hdr_host, &hdr, sizeof(hdr));
}

int64_t pcgrl_state_bytes(pcgrl_handle h) {
  if (!h) return -1;
  size_t total = STATE_HDR_BYTES;
  for (auto &a : h->state_arrays) total += state_section(a.second * (size_t)h->p.n_envs);
  return (int64_t)total;
}

int pcgrl_export_state(pcgrl_handle h, uint8_t *d_buf, int32_t *maybe_stale_out, void *stream) {
  if (!h || !d_buf) return fail(PCGRL_EINVAL, "pcgrl_export_state: bad arguments");
  ON_DEVICE(h->device);
  // (pinned, engine-owned source: capturable, and no stack frame for the runtime to read later)
  HIPCHK(hipMemcpyAsync(d_buf, h->hdr_host, STATE_HDR_BYTES, hipMemcpyHostToDevice, (hipStream_t)stream));
  size_t off = STATE_HDR_BYTES;
  for (auto &a : h->state_arrays) {
    const size_t bytes = a.second * (size_t)h->p.n_envs;
    HIPCHK(hipMemcpyAsync(d_buf + off, a.first, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    off += state_section(bytes);
  }
  if (maybe_stale_out) *maybe_stale_out = h->maybe_stale ? 1 : 0;
  return PCGRL_OK;
}

int pcgrl_import_state(pcgrl_handle h, const uint8_t *d_mask, const uint8_t *d_buf, int32_t maybe_stale, void *stream) {
  if (!h || !d_buf) return fail(PCGRL_EINVAL, "pcgrl_import_state: bad arguments");
  ON_DEVICE(h->device);
  // the header is checked on the host before anything is overwritten: one small copy and a wait for `stream`
  StateHeader hdr{};
  HIPCHK(hipMemcpyAsync(&hdr, d_buf, sizeof(hdr), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  if (hdr.magic != STATE_MAGIC) return fail(PCGRL_EINVAL, "pcgrl_import_state: not a pcgrl_export_state image (bad magic)");
  if (hdr.n_envs != h->p.n_envs || hdr.n_arrays != (int32_t)h->state_arrays.size() || hdr.total_bytes != (uint64_t)pcgrl_state_bytes(h) ||
      hdr.fingerprint != state_fingerprint(h))
    return fail(PCGRL_EINVAL, "pcgrl_import_state: the image was exported by an engine with another config, batch size or library version");
  if (d_mask == nullptr && h->p.cfg.static_tiles) {
    // the exporter's run-time static-tile parameters (a curriculum's set_static_prob, evaluation mode) come with the image;
    // a masked import leaves the engine-wide parameters as they are
    if (!(hdr.static_prob >= 0.0 && hdr.static_prob <= 1.0) || hdr.n_static_walls < 0)
      return fail(PCGRL_EINVAL, "pcgrl_import_state: corrupt static-tile parameters in the image header");
    h->p.cfg.static_prob = hdr.static_prob;
    h->p.cfg.n_static_walls = hdr.n_static_walls;
    h->p.cfg.static_eval = hdr.static_eval ? 1 : 0;
    state_header_build(h);
  }
  size_t off = STATE_HDR_BYTES;
  for (auto &a : h->state_arrays) {
    const size_t bytes = a.second * (size_t)h->p.n_envs;
    if (d_mask == nullptr) {
      HIPCHK(hipMemcpyAsync(a.first, d_buf + off, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    } else {
      const int64_t wpe = (int64_t)(a.second / 4), total = wpe * h->p.n_envs;  // (every array's row is a multiple of 4 bytes)
      hipLaunchKernelGGL(masked_rows_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (uint32_t *)a.first, (const uint32_t *)(d_buf + off), wpe, h->p.n_envs, d_mask);
      HIPCHK(hipGetLastError());
    }
    off += state_section(bytes);
  }
  // the imported envs may carry statistics left stale by pcgrl_update: the host-side kernel choice follows the exporter's
  h->maybe_stale = d_mask == nullptr ? (maybe_stale != 0) : (h->maybe_stale || maybe_stale != 0);
  return PCGRL_OK;
}

int pcgrl_get_rng_state(pcgrl_handle h, uint64_t *d_out, void *stream) {
  if (!h || !d_out) return fail(PCGRL_EINVAL, "pcgrl_get_rng_state: bad arguments");
  ON_DEVICE(h->device);
  hipLaunchKernelGGL(rng_state_kernel, dim3((h->p.n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->p, d_out, (const uint64_t *)nullptr);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_set_rng_state(pcgrl_handle h, const uint8_t *d_mask, const uint64_t *d_in, void *stream) {
  if (!h || !d_in) return fail(PCGRL_EINVAL, "pcgrl_set_rng_state: bad arguments");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.mask = d_mask;
  hipLaunchKernelGGL(rng_state_kernel, dim3((p.n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, p, (uint64_t *)nullptr, d_in);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

static int32_t num_actions_of(const pcgrl_engine *h) {
  const pcgrl_config &c = h->p.cfg;
  if (c.act_window[0] > 0) return h->p.n_tiles;  // MultiDiscrete([n_tiles] * prod(act_window)), reps/wrappers.py:475-478
  switch (c.representation) {
    case PCGRL_REP_NARROW: return h->p.n_tiles;      // narrow_rep.py: Discrete(n_tiles)
    case PCGRL_REP_TURTLE: return h->p.n_tiles + 4;  // turtle_rep.py: 4 moves + n_tiles (A6)
    default: return h->p.n_cells * h->p.n_tiles;     // wide: ActionMap's flat (cell, tile) index (A7)
  }
}

int32_t pcgrl_num_actions(pcgrl_handle h) { return h ? num_actions_of(h) : -1; }

int pcgrl_sample_actions(pcgrl_handle h, int32_t *d_actions, uint64_t seed, void *stream) {
  if (!h || !d_actions) return fail(PCGRL_EINVAL, "pcgrl_sample_actions: bad arguments");
  ON_DEVICE(h->device);
  const int32_t n = h->p.n_envs * h->p.n_act;
  hipLaunchKernelGGL(sample_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_actions, n,
                     (uint32_t)num_actions_of(h), seed, h->sample);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_set_solver_budget(pcgrl_handle h, int32_t budget) {
  if (!h || budget < 0) return fail(PCGRL_EINVAL, "pcgrl_set_solver_budget: bad arguments");
  if (h->p.cfg.problem != PCGRL_PROB_SOKOBAN || !h->p.soko)
    return fail(PCGRL_EUNSUPPORTED, "pcgrl_set_solver_budget: only sokoban has a device solver");
  if (h->p.ext || h->p.cfg.n_ctrl > 0)
    return fail(PCGRL_EUNSUPPORTED, "pcgrl_set_solver_budget: asynchronous stepping is built for sokoban without static tiles, action "
                                    "patches or control metrics");
  if (budget > 0 && h->maybe_stale)
    return fail(PCGRL_EINVAL, "pcgrl_set_solver_budget: statistics left stale by pcgrl_update may exist (the asynchronous kernels carry no "
                              "code for them): call pcgrl_refresh_stats or reset the envs first");
  ON_DEVICE(h->device);
  if (budget > 0 && h->sk_ws == nullptr) {  // one stage workspace + one park record per env, once
    HIPCHK(hipDeviceSynchronize());
    const hipError_t e = sokoban_alloc_async(h->p, h->allocs, h->p.n_envs, &h->sk_ws, &h->sk_park);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(PCGRL_EHIP, std::string("pcgrl_set_solver_budget: allocating one solver workspace per env: ") + hipGetErrorString(e));
    }
  }
  if (budget == 0 && h->sk_budget > 0) {  // back to synchronous stepping: nobody may be waiting for a parked search
    HIPCHK(hipDeviceSynchronize());
    std::vector<EnvState> st(h->p.n_envs);
    HIPCHK(hipMemcpy(st.data(), h->p.st, st.size() * sizeof(EnvState), hipMemcpyDeviceToHost));
    for (auto &e : st)
      if (e.flags & (ENV_PENDING_STEP | ENV_PENDING_STATS))
        return fail(PCGRL_EINVAL, "pcgrl_set_solver_budget(0): an env is still busy (a parked search): keep stepping with pcgrl_step_ready "
                                  "or reset the envs first");
  }
  h->sk_budget = budget;
  return PCGRL_OK;
}

int32_t pcgrl_get_solver_budget(pcgrl_handle h) { return h ? h->sk_budget : -1; }

int pcgrl_step_ready(pcgrl_handle h, const int32_t *d_actions, int32_t auto_reset, uint8_t *d_obs, float *d_reward, uint8_t *d_done,
                     int32_t *d_stats, uint8_t *d_status, void *stream) {
  if (!h || !d_actions || !d_status) return fail(PCGRL_EINVAL, "pcgrl_step_ready: bad arguments");
  if (h->sk_budget <= 0) return fail(PCGRL_EINVAL, "pcgrl_step_ready: no solver budget set (pcgrl_set_solver_budget)");
  ON_DEVICE(h->device);
  Params p = h->p;
  p.no_fast = h->maybe_stale ? 1 : 0;
  p.spread = 1;  // one env per workgroup: every search has a wavefront of its own
  p.sk_helpers = 0;
  p.sk_budget = h->sk_budget;
  p.actions = d_actions;
  p.auto_reset = auto_reset;
  p.obs = d_obs;
  p.reward = d_reward;
  p.done = d_done;
  p.stats_out = d_stats;
  p.ready = d_status;
  HIPCHK(launch(K_STEP, h->lpe, p, h->lds_bytes, (hipStream_t)stream, h->cpl));
  return PCGRL_OK;
}

int pcgrl_env_busy(pcgrl_handle h, uint8_t *d_busy, void *stream) {
  if (!h || !d_busy) return fail(PCGRL_EINVAL, "pcgrl_env_busy: bad arguments");
  ON_DEVICE(h->device);
  hipLaunchKernelGGL(env_busy_kernel, dim3((h->p.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->p, d_busy);
  HIPCHK(hipGetLastError());
  return PCGRL_OK;
}

int pcgrl_reserve_solver_pool(pcgrl_handle h, int32_t n_slots, int32_t allow_lazy_growth) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_reserve_solver_pool: null handle");
  if (!h->p.soko) return fail(PCGRL_EINVAL, "pcgrl_reserve_solver_pool: the engine has no device solver (sokoban only)");
  ON_DEVICE(h->device);
  h->soko_lazy = allow_lazy_growth != 0;
  const int full = sokoban_slots_for(h->p.n_envs);
  const int want = n_slots < 0 ? h->soko_slots : (n_slots == 0 ? full : std::min((int)n_slots, 512));
  Params scratch = h->p;
  return soko_pool_for(h, scratch, want, nullptr, /*lazy=*/false);
}

int32_t pcgrl_solver_pool_slots(pcgrl_handle h, int32_t *full_size_out, int32_t *grow_failed_out) {
  if (!h || !h->p.soko) return 0;
  if (full_size_out) *full_size_out = sokoban_slots_for(h->p.n_envs);
  if (grow_failed_out) *grow_failed_out = h->soko_grow_failed ? 1 : 0;
  if (h->soko_grow_failed) g_err = h->soko_grow_msg;
  return h->soko_slots;
}

// Host utilities for a foreign-language host (ctypes): the HIP runtime THIS library is linked against -- the one that owns
// the handles the caller passes -- instead of a second dlopen of libamdhip64 by name.
int pcgrl_copy_to_host(void *dst_host, const void *d_src, int64_t bytes, void *stream) {
  if (!dst_host || !d_src || bytes < 0) return fail(PCGRL_EINVAL, "pcgrl_copy_to_host: bad arguments");
  HIPCHK(hipMemcpyAsync(dst_host, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return PCGRL_OK;
}

int pcgrl_stream_synchronize(void *stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return PCGRL_OK;
}

int pcgrl_graph_upload(void *graph_exec, void *stream) {
  if (!graph_exec) return fail(PCGRL_EINVAL, "pcgrl_graph_upload: null graph");
  HIPCHK(hipGraphUpload((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return PCGRL_OK;
}

int pcgrl_debug_counters(pcgrl_handle h, uint64_t *out, int32_t n) {
  if (!h || !out || n < 1 || n > 8 * (h->p.n_envs + 64)) return fail(PCGRL_EINVAL, "pcgrl_debug_counters: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->p.err + 64, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(h->p.err + 64, 0, sizeof(uint64_t) * n));
  return PCGRL_OK;
}

int pcgrl_poll_error(pcgrl_handle h) {
  if (!h) return fail(PCGRL_EINVAL, "pcgrl_poll_error: null handle");
  ON_DEVICE(h->device);
  HIPCHK(hipDeviceSynchronize());
  int32_t flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpy(flags, h->p.err, sizeof(flags), hipMemcpyDeviceToHost));
  if (flags[0]) {
    HIPCHK(hipMemset(h->p.err, 0, sizeof(flags)));
    if (flags[0] & 1) return fail(PCGRL_EACTION, "an action was outside the action space (the reference raises IndexError)");
    if (flags[0] & 2) return fail(PCGRL_EUNSUPPORTED, "sokoban solver: level exceeds the device solver's limits");
    if (flags[0] & 4) return fail(PCGRL_EUNSUPPORTED, "3-D maze path search: more live queue entries than the search ring holds (512 for planes of <= 64 cells, else 4096); the statistics of that step were kept at their previous values and are recomputed from scratch at the env's next changing step");
    if (flags[0] & 8)
      return fail(PCGRL_ESTALE, "a step launch issued before pcgrl_update (a replayed HIP graph?) met statistics that pcgrl_update "
                                "left stale: re-capture after pcgrl_update or call pcgrl_refresh_stats first (include/pcgrl_amd.h, HIP graphs)");
    return fail(PCGRL_EINVAL, "device error flag set");
  }
  return PCGRL_OK;
}

}  // extern "C"
