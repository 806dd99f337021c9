// pcgrl_k_sokoban32_16.hip -- translation unit: the SOKOBAN kernels with 32-bit row masks, 16 lanes per env
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_sokoban.h"

PCGRL_DEFINE_LAUNCH_ONE(launch_sokoban32_16, PCGRL_PROB_SOKOBAN, 16, uint32_t)
