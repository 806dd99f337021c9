// pcgrl_k_sokoban32_8.hip -- translation unit: the SOKOBAN kernels with 32-bit row masks, 8 lanes per env
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_sokoban.h"

PCGRL_DEFINE_LAUNCH_ONE(launch_sokoban32_8, PCGRL_PROB_SOKOBAN, 8, uint32_t)
