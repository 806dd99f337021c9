// pcgrl_k_3d.hip -- translation unit: the minecraft_3D_maze kernels (see pcgrl_dispatch.h)
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_kernels3d.h"

namespace pcgrl {
template <int SC>
static hipError_t launch_3d_sc(KernelId id, const Params &p, int cpl, hipStream_t s) {
  dim3 grid(p.n_envs), block(64);
  switch (id) {
    case K_STEP: hipLaunchKernelGGL((m3_kernel<M3_STEP, SC>), grid, dim3(64 * (2 + m3_observers<SC>())), 0, s, p, cpl); break;  // simulate + observe + helper waves
    case K_RESET: hipLaunchKernelGGL((m3_kernel<M3_RESET, SC>), grid, block, 0, s, p, cpl); break;
    case K_OBSERVE: hipLaunchKernelGGL((m3_kernel<M3_OBSERVE, SC>), grid, block, 0, s, p, cpl); break;
    case K_GET_STATE: hipLaunchKernelGGL((m3_kernel<M3_GET_STATE, SC>), grid, block, 0, s, p, cpl); break;
    case K_STATS_FOR_GRIDS: hipLaunchKernelGGL((m3_kernel<M3_STATS_FOR_GRIDS, SC>), grid, block, 0, s, p, cpl); break;
    case K_LAST_EPISODE:
      hipLaunchKernelGGL((last_episode_kernel<PCGRL_PROB_MC3DMAZE, 64>), dim3((p.n_envs + 63) / 64), block, 0, s, p);
      break;
    case K_ROLLOUT: hipLaunchKernelGGL((m3_kernel<M3_ROLLOUT, SC>), grid, block, 0, s, p, cpl); break;
  }
  return hipGetLastError();
}
}  // namespace pcgrl

hipError_t pcgrl::launch_3d(KernelId id, const Params &p, int cpl, hipStream_t s) {
  dim3 grid(p.n_envs), block(64);
  const bool d7 = p.cfg.dims[0] == 7 && p.cfg.dims[1] == 7 && p.cfg.dims[2] == 7 && p.cfg.obs_window[0] == 14 &&
                  p.cfg.obs_window[1] == 14 && p.cfg.obs_window[2] == 14;
  if (d7 && id == K_STEP) {
    hipLaunchKernelGGL((m3_kernel<M3_STEP, 0, 7>), grid, dim3(192), 0, s, p, cpl);
    return hipGetLastError();
  }
  if (d7 && id == K_ROLLOUT) {
    hipLaunchKernelGGL((m3_kernel<M3_ROLLOUT, 0, 7>), grid, block, 0, s, p, cpl);
    return hipGetLastError();
  }
  // the reference's stock map (configs/config.py:153-157) with its 30^3 window: the same with compile-time dimensions (the
  // size class 1 kernels keep ~100 wave-uniform values alive and spill scalar registers into vector lanes: constant strides
  // and bounds take a third of the step kernel's instructions away)
  const bool d15 = p.cfg.dims[0] == 15 && p.cfg.dims[1] == 15 && p.cfg.dims[2] == 15 && p.cfg.obs_window[0] == 30 &&
                   p.cfg.obs_window[1] == 30 && p.cfg.obs_window[2] == 30;
  if (d15 && id == K_STEP) {
    hipLaunchKernelGGL((m3_kernel<M3_STEP, 1, 15>), grid, dim3(64 * (2 + m3_observers<1>())), 0, s, p, cpl);
    return hipGetLastError();
  }
  if (m3_size_class(p.cfg.dims[0], p.cfg.dims[1], p.cfg.dims[2]) == 0) return launch_3d_sc<0>(id, p, cpl, s);
  return launch_3d_sc<1>(id, p, cpl, s);
}
