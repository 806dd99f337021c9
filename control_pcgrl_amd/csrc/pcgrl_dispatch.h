// pcgrl_dispatch.h -- kernel dispatch of the 2-D problems, shared by the per-problem translation units.
//
// The library is built from several translation units so that the ~170 kernel instantiations compile in parallel
// (pcgrl_k_<problem><maskbits>.hip + pcgrl_k_3d.hip + pcgrl_engine.hip, see _lib.build()).  Each 2-D unit defines ONE
// launch function with PCGRL_DEFINE_LAUNCH; pcgrl_engine.hip only sees the declarations below.
#pragma once
#include <hip/hip_runtime.h>

#include "pcgrl_common.h"

namespace pcgrl {

// K_ROLLOUT_SIM / K_ROLLOUT_OBS: the two roles of a rollout as kernels of their own (16x16 compile-time kernels, plain mode)
// K_ROLLOUT_RESIDENT launches nothing: it asks the runtime how many workgroups of the two-role rollout kernel one CU holds
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) and leaves the answer in rollout_resident_per_cu
enum KernelId { K_STEP, K_RESET, K_OBSERVE, K_GET_STATE, K_LAST_EPISODE, K_STATS_FOR_GRIDS, K_ROLLOUT, K_ROLLOUT_SIM, K_ROLLOUT_OBS, K_ROLLOUT_RESIDENT };
inline thread_local int rollout_resident_per_cu = 0;

// one per translation unit (M = row-mask type: 32-bit for W <= 32, 64-bit for W <= 64)
hipError_t launch_binary32(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_binary64(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_zelda32(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_zelda64(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s);
// sokoban carries the device solver in every statistics-computing kernel: one unit per lanes-per-env value
hipError_t launch_sokoban32_8(KernelId id, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_sokoban32_16(KernelId id, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_sokoban32_32(KernelId id, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_sokoban32_64(KernelId id, const Params &p, size_t lds, hipStream_t s);
inline hipError_t launch_sokoban32(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s) {
  switch (lpe) {
    case 8: return launch_sokoban32_8(id, p, lds, s);
    case 16: return launch_sokoban32_16(id, p, lds, s);
    case 32: return launch_sokoban32_32(id, p, lds, s);
    default: return launch_sokoban32_64(id, p, lds, s);
  }
}
// maps wider than 32: 64-bit row masks, 32 or 64 lanes per env
hipError_t launch_sokoban64_32(KernelId id, const Params &p, size_t lds, hipStream_t s);
hipError_t launch_sokoban64_64(KernelId id, const Params &p, size_t lds, hipStream_t s);
inline hipError_t launch_sokoban64(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s) {
  return lpe == 32 ? launch_sokoban64_32(id, p, lds, s) : launch_sokoban64_64(id, p, lds, s);
}
hipError_t launch_3d(KernelId id, const Params &p, int cpl, hipStream_t s);

}  // namespace pcgrl

#ifdef PCGRL_KERNEL_TU
#include "pcgrl_kernels2d.h"

namespace pcgrl {

// LDS above the 64 KiB default needs an explicit opt-in per kernel (64x64 zelda rows are 1152 B x 65 = 74 KiB)
template <typename K>
static hipError_t allow_lds(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int PROB, int LPE, typename M>
static hipError_t launch_pl(KernelId id, const Params &p, size_t lds, hipStream_t s) {
  const int epw = (id == K_STEP && p.spread) ? 1 : 64 / LPE;
  dim3 grid((p.n_envs + epw - 1) / epw), block(64);
  // compile-time specialised 16x16 kernels: cropped 32x32 window (reference default obs_window = 2 * map), or the
  // wide representation's whole-map observation
  const bool m16 = p.cfg.dims[0] == 16 && p.cfg.dims[1] == 16 && !p.ext;
  const bool fast_cfg = m16 && (p.cfg.representation == PCGRL_REP_WIDE
                                    ? true
                                    : (p.cfg.obs_window[0] == 32 && p.cfg.obs_window[1] == 32));
  // stale statistics (after pcgrl_update) are only handled by the general step / rollout kernels
  const bool fast = fast_cfg && !(p.no_fast && (id == K_STEP || id == K_ROLLOUT) && !p.update_only);
  // sokoban with p.sk_helpers: the solver's helper wavefronts ride behind the simulate / observe waves
  const unsigned helper_threads = PROB == PCGRL_PROB_SOKOBAN ? 2u * 64u * (unsigned)p.sk_helpers : 0u;  // (heap + expander wave per stage)
  const size_t helper_lds = PROB == PCGRL_PROB_SOKOBAN ? (size_t)p.sk_helpers * SK_HELPER_LDS : 0;  // (their A* heaps)
  hipError_t e = hipSuccess;
  if constexpr (PROB == PCGRL_PROB_SOKOBAN) {
    // asynchronous stepping (pcgrl_step_ready / pcgrl_reset with a solver budget): the resumable-solver variants, one env per
    // workgroup of two waves (simulate, observe), no helper waves
    if (p.sk_budget > 0 && id == K_STEP) {
      const dim3 g1(p.n_envs);
      if constexpr (LPE == 16 && sizeof(M) == 4) {
        if (fast) {
          hipLaunchKernelGGL((step_kernel<PROB, LPE, M, true, false, 1, true>), g1, dim3(128), lds, s, p);
          return hipGetLastError();
        }
      }
      if ((e = allow_lds(step_kernel<PROB, LPE, M, false, false, 1, true>, lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((step_kernel<PROB, LPE, M, false, false, 1, true>), g1, dim3(128), lds, s, p);
      return hipGetLastError();
    }
    if (p.sk_budget > 0 && id == K_RESET) {
      hipLaunchKernelGGL((reset_kernel<PROB, LPE, M, true>), dim3(p.n_envs), block, 0, s, p);
      return hipGetLastError();
    }
  }
  switch (id) {
    case K_STEP:
      if constexpr (LPE == 16 && sizeof(M) == 4) {
        if (fast) {
          const bool ctrl = p.trg || p.reward64;
#if !defined(PCGRL_PHASE_TIMING) && !defined(PCGRL_WAVE_TRACE)  // (the development counters are indexed by workgroup)
          if constexpr (PROB == PCGRL_PROB_BINARY) {  // two wave pairs per workgroup
            const dim3 g2((grid.x + 1) / 2);
            if (ctrl)
              hipLaunchKernelGGL((step_kernel<PROB, LPE, M, true, true, 2>), g2, dim3(256), 2 * lds, s, p);
            else
              hipLaunchKernelGGL((step_kernel<PROB, LPE, M, true, false, 2>), g2, dim3(256), 2 * lds, s, p);
            break;
          }
#endif
          if (ctrl) {
            if ((e = allow_lds(step_kernel<PROB, LPE, M, true, true>, lds + helper_lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((step_kernel<PROB, LPE, M, true, true>), grid, dim3(128 + helper_threads), lds + helper_lds, s, p);
          } else {
            if ((e = allow_lds(step_kernel<PROB, LPE, M, true, false>, lds + helper_lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((step_kernel<PROB, LPE, M, true, false>), grid, dim3(128 + helper_threads), lds + helper_lds, s, p);
          }
          break;
        }
      }
      if (p.trg || p.reward64) {
        if ((e = allow_lds(step_kernel<PROB, LPE, M, false, true>, lds + helper_lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((step_kernel<PROB, LPE, M, false, true>), grid, dim3(128 + helper_threads), lds + helper_lds, s, p);
      } else {
        if ((e = allow_lds(step_kernel<PROB, LPE, M, false, false>, lds + helper_lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((step_kernel<PROB, LPE, M, false, false>), grid, dim3(128 + helper_threads), lds + helper_lds, s, p);
      }
      break;
    case K_ROLLOUT: {
      const bool ctrl = p.trg || p.reward64;
      if (p.spread > 0) grid = dim3((p.n_envs + p.spread - 1) / p.spread);  // (envs per wavefront chosen by the host)
      if constexpr (LPE == 16 && sizeof(M) == 4) {
        if (fast) {
          if (ctrl)
            hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, true, true>), grid, dim3(128), lds, s, p);
          else
            hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, true, false>), grid, dim3(128), lds, s, p);
          break;
        }
      }
      if (ctrl) {
        if ((e = allow_lds(rollout_kernel<PROB, LPE, M, false, true>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, false, true>), grid, dim3(128), lds, s, p);
      } else {
        if ((e = allow_lds(rollout_kernel<PROB, LPE, M, false, false>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, false, false>), grid, dim3(128), lds, s, p);
      }
      break;
    }
    case K_ROLLOUT_SIM:
    case K_ROLLOUT_OBS:
    case K_ROLLOUT_RESIDENT:
      if constexpr (LPE == 16 && sizeof(M) == 4) {
        if (fast && !(p.trg || p.reward64)) {
          if (id == K_ROLLOUT_RESIDENT) {
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rollout_kernel<PROB, LPE, M, true, false, 0>, 128, lds);
            rollout_resident_per_cu = nb;
            return e;
          } else if (id == K_ROLLOUT_SIM) {
            const int epw = p.spread > 0 ? p.spread : 64 / LPE;
            hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, true, false, 1>), dim3((p.n_envs + epw - 1) / epw), dim3(64), 0, s, p);
          } else {
            hipLaunchKernelGGL((rollout_kernel<PROB, LPE, M, true, false, 2>), grid, dim3(64), lds, s, p);
          }
          break;
        }
      }
      return hipErrorInvalidValue;  // (the engine only asks for the split form where these kernels exist)
    case K_RESET: hipLaunchKernelGGL((reset_kernel<PROB, LPE, M>), grid, block, 0, s, p); break;
    case K_OBSERVE:
      if constexpr (LPE == 16 && sizeof(M) == 4) {
        if (fast) {
          hipLaunchKernelGGL((observe_kernel<PROB, LPE, M, true>), grid, block, lds, s, p);
          break;
        }
      }
      if ((e = allow_lds(observe_kernel<PROB, LPE, M, false>, lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((observe_kernel<PROB, LPE, M, false>), grid, block, lds, s, p);
      break;
    case K_GET_STATE: hipLaunchKernelGGL((get_state_kernel<PROB, LPE, M>), grid, block, 0, s, p); break;
    case K_LAST_EPISODE:
      hipLaunchKernelGGL((last_episode_kernel<PROB, LPE>), dim3((p.n_envs + 63) / 64), block, 0, s, p);
      break;
    case K_STATS_FOR_GRIDS:  // (sokoban: one map per wavefront, see the kernel)
      if ((e = allow_lds(stats_for_grids_kernel<PROB, LPE, M>, helper_lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((stats_for_grids_kernel<PROB, LPE, M>), PROB == PCGRL_PROB_SOKOBAN ? dim3(p.n_envs) : grid,
                         dim3(64 + helper_threads), helper_lds, s, p);
      break;
  }
  return hipGetLastError();
}

}  // namespace pcgrl

// W <= 32: one lane per row for H <= 8 / 16 / 32 / 64;  W <= 64: validate() guarantees H > 16
#define PCGRL_DEFINE_LAUNCH32(name, PROB)                                                            \
  hipError_t pcgrl::name(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s) {         \
    switch (lpe) {                                                                                   \
      case 8: return launch_pl<PROB, 8, uint32_t>(id, p, lds, s);                                    \
      case 16: return launch_pl<PROB, 16, uint32_t>(id, p, lds, s);                                  \
      case 32: return launch_pl<PROB, 32, uint32_t>(id, p, lds, s);                                  \
      default: return launch_pl<PROB, 64, uint32_t>(id, p, lds, s);                                  \
    }                                                                                                \
  }
#define PCGRL_DEFINE_LAUNCH_ONE(name, PROB, LPE, M)                                                  \
  hipError_t pcgrl::name(KernelId id, const Params &p, size_t lds, hipStream_t s) {                  \
    return launch_pl<PROB, LPE, M>(id, p, lds, s);                                                   \
  }
#define PCGRL_DEFINE_LAUNCH64(name, PROB)                                                            \
  hipError_t pcgrl::name(KernelId id, int lpe, const Params &p, size_t lds, hipStream_t s) {         \
    if (lpe == 32) return launch_pl<PROB, 32, uint64_t>(id, p, lds, s);                              \
    return launch_pl<PROB, 64, uint64_t>(id, p, lds, s);                                             \
  }
#endif  // PCGRL_KERNEL_TU
