// pcgrl_k_sokoban32_32.hip -- translation unit: the SOKOBAN kernels with 32-bit row masks, 32 lanes per env
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"
#include "pcgrl_sokoban.h"

PCGRL_DEFINE_LAUNCH_ONE(launch_sokoban32_32, PCGRL_PROB_SOKOBAN, 32, uint32_t)
