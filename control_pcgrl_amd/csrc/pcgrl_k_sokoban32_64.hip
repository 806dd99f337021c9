// pcgrl_k_sokoban32_64.hip -- translation unit: the SOKOBAN kernels with 32-bit row masks, 64 lanes per env
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_sokoban.h"

PCGRL_DEFINE_LAUNCH_ONE(launch_sokoban32_64, PCGRL_PROB_SOKOBAN, 64, uint32_t)
