// pcgrl_k_sokoban64_64.hip -- translation unit: the SOKOBAN kernels with 64-bit row masks (maps wider than 32), 64 lanes per env
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_sokoban.h"

PCGRL_DEFINE_LAUNCH_ONE(launch_sokoban64_64, PCGRL_PROB_SOKOBAN, 64, uint64_t)
